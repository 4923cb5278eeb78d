import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from sfh_amd import synth
from sfh_amd.reconstructor import Reconstructor
W, H = 640, 360
dev = torch.device("cuda", 0)
for B in (1, 2, 4):
    court = synth.load_court_template("ncaa_nc4_640x360", 4, B).to(dev); poi = synth.load_court_poi("pitch", B).to(dev)
    net = Reconstructor(court, poi, target_size=(W, H), unet_size=(W, H), warp_size=(W, H), warp_with_nearest=True)
    net.load_state_dict(synth.synth_state_dict(net.state_dict(), 0)); net.to(dev).eval()
    x = synth.frames_to_float(synth.synth_frames_u8(B, H, W, seed=0)).to(dev)
    def piped(m):
        prev = None
        for _ in range(m):
            h = net.predict_async(x, consistency=False)
            if prev is not None: prev.result()
            prev = h
        prev.result()
    out = []
    with torch.no_grad():
        for rep in range(4):
            for flag in (True, False):
                net.pipeline_splitk = flag
                piped(10); torch.cuda.synchronize()
                n = 200 // B
                t0 = time.perf_counter(); piped(n); torch.cuda.synchronize()
                out.append((flag, (time.perf_counter() - t0) / n * 1e3))
    print(f"B={B}: " + "  ".join(f"{'split' if f else 'nosplit'} {t:.3f}" for f, t in out), flush=True)
