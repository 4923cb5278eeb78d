import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from sfh_amd import synth
from sfh_amd.reconstructor import Reconstructor
B, W, H = 16, 640, 360
dev = torch.device("cuda", 0)
court = synth.load_court_template("ncaa_nc4_640x360", 4, B).to(dev); poi = synth.load_court_poi("pitch", B).to(dev)
net = Reconstructor(court, poi, target_size=(W, H), unet_size=(W, H), warp_size=(W, H), warp_with_nearest=True)
net.load_state_dict(synth.synth_state_dict(net.state_dict(), 0)); net.to(dev).eval()
x = synth.frames_to_float(synth.synth_frames_u8(B, H, W, seed=0)).to(dev)
with torch.no_grad():
    for _ in range(3): net.predict(x, consistency=False)
    torch.cuda.synchronize()
    # (1) sync predict
    t = time.perf_counter()
    for _ in range(20): net.predict(x, consistency=False)
    torch.cuda.synchronize(); a = (time.perf_counter() - t) / 20
    # (2) guard off: enqueue only, sync at the end
    net.range_guard = False
    t = time.perf_counter()
    for _ in range(20): net.predict(x, consistency=False)
    enq = (time.perf_counter() - t) / 20
    torch.cuda.synchronize(); b = (time.perf_counter() - t) / 20
    print(f"predict() with guard {a*1e3:.3f} ms; guard off: host enqueue {enq*1e3:.3f} ms per batch, wall {b*1e3:.3f} ms per batch")
    # (3) pieces of the host path
    t = time.perf_counter()
    for _ in range(200): net._param_stamp()
    print(f"_param_stamp {1e6*(time.perf_counter()-t)/200:.1f} us")
    net.range_guard = True
    rg = net._h2_ranges
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(50): rg.read()
    print(f"range read-back (idle GPU) {1e6*(time.perf_counter()-t)/50:.1f} us")
