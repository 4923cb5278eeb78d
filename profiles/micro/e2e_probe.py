"""Where the end-to-end pipeline's time goes: H2D bandwidth of pinned uint8 frames, the preprocessing kernels alone,
and FramePipeline with / without the downloads.  usage: python profiles/micro/e2e_probe.py"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import torch
from sfh_amd import synth, engine as E
from sfh_amd.pipeline import FramePipeline
from sfh_amd.reconstructor import Reconstructor
B, W, H = 16, 640, 360
dev = torch.device("cuda", 0)
court = synth.load_court_template("ncaa_nc4_640x360", 4, B).to(dev); poi = synth.load_court_poi("pitch", B).to(dev)
net = Reconstructor(court, poi, target_size=(W, H), unet_size=(W, H), warp_size=(W, H), warp_with_nearest=True)
net.load_state_dict(synth.synth_state_dict(net.state_dict(), 0)); net.to(dev).eval()
for k in (1, 3):
    host = [torch.from_numpy(synth.synth_frames_u8(B, H * k, W * k, seed=s)).pin_memory() for s in range(2)]
    devb = torch.empty_like(host[0], device=dev)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for i in range(20): devb.copy_(host[i % 2], non_blocking=True)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 20
    print(f"k={k}: H2D {host[0].numel()/1e6:.1f} MB in {dt*1e3:.3f} ms = {host[0].numel()/dt/1e9:.1f} GB/s")
    t = time.perf_counter()
    for i in range(20): x = E.frames_u8_to_input(devb, (W, H) if k != 1 else None)
    torch.cuda.synchronize(); print(f"k={k}: preprocessing kernel {(time.perf_counter()-t)/20*1e3:.3f} ms")
    for req, cons in ((("theta", "warp_mask", "segm_mask"), True), (("theta",), False)):
        pipe = FramePipeline(net, B, (H * k, W * k), req_outputs=req, consistency=cons)
        with torch.no_grad():
            n = sum(1 for _ in pipe.run(host[i % 2] for i in range(4)))
            torch.cuda.synchronize(); t = time.perf_counter()
            n = sum(1 for _ in pipe.run(host[i % 2] for i in range(12)))
            torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 12
        print(f"k={k}: FramePipeline req={req} consistency={cons}: {dt*1e3:.3f} ms per batch = {B/dt:.1f} frames/s")
    with torch.no_grad():
        x = E.frames_u8_to_input(devb, (W, H) if k != 1 else None)
        for cons in (True, False):
            prev = None
            torch.cuda.synchronize(); t = time.perf_counter()
            for i in range(12):
                h = net.predict_async(x, consistency=cons)
                if prev is not None: prev.result()
                prev = h
            prev.result(); torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 12
            print(f"k={k}: device-to-device predict_async consistency={cons}: {dt*1e3:.3f} ms per batch")
