"""Does a HIGH-priority caller stream (UNet launches) next to the normal-priority side stream (ResNet-STN / warp of the
previous batch) let the small launches merely fill the gaps?  Pipelined predict_async loop, caller stream priority 0 / -1."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import torch
from sfh_amd import synth
from sfh_amd.reconstructor import Reconstructor
B, W, H = 16, 640, 360
dev = torch.device("cuda", 0)
court = synth.load_court_template("ncaa_nc4_640x360", 4, B).to(dev); poi = synth.load_court_poi("pitch", B).to(dev)
x = synth.frames_to_float(synth.synth_frames_u8(B, H, W, seed=0)).to(dev)
for rnd in range(2):
    for prio in (None, 0, -1):
        net = Reconstructor(court, poi, target_size=(W, H), unet_size=(W, H), warp_size=(W, H), warp_with_nearest=True)
        net.load_state_dict(synth.synth_state_dict(net.state_dict(), 0)); net.to(dev).eval()
        st = torch.cuda.current_stream(dev) if prio is None else torch.cuda.Stream(dev, priority=prio)   # None: torch's default stream
        with torch.no_grad(), torch.cuda.stream(st):
            def run(n):
                prev = None
                for _ in range(n):
                    h = net.predict_async(x, consistency=False)
                    if prev is not None: prev.result()
                    prev = h
                prev.result()
            run(4); torch.cuda.synchronize(); t = time.perf_counter(); run(30); torch.cuda.synchronize()
            dt = (time.perf_counter() - t) / 30
        print(f"caller stream priority {prio}: {dt*1e3:.3f} ms per batch = {B/dt:.1f} frames/s", flush=True)
        del net
