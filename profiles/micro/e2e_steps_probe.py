import sys, time
sys.path.insert(0, "/root/repo")
import torch, bench
from sfh_amd import synth
from sfh_amd.reconstructor import Reconstructor
B, W, H = 16, 640, 360
dev = torch.device("cuda", 0)
court = synth.load_court_template("ncaa_nc4_640x360", 4, B).to(dev); poi = synth.load_court_poi("pitch", B).to(dev)
net = Reconstructor(court, poi, target_size=(W, H), unet_size=(W, H), warp_size=(W, H), warp_with_nearest=True)
net.load_state_dict(synth.synth_state_dict(net.state_dict(), 0)); net.to(dev).eval()
x = synth.frames_to_float(synth.synth_frames_u8(B, H, W, seed=0)).to(dev)
d2d = bench._timed_predicts(net, x, 20, 3, True, consistency=True)
print("d2d ms/batch", d2d / 20 * 1e3)
for src in ((W, H), (3 * W, 3 * H)):
    for n in (10, 20, 40, 10):
        fps, ms = bench.e2e_bench(net, B, W, H, src, n, 3)
        print(src, "n", n, round(ms, 3), "ms/batch", round(fps, 1), "frames/s", flush=True)
