"""cProfile of the host side of predict() (B=16, 640x360): where do the ~26 us per launch go?"""
import cProfile, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from sfh_amd import synth
from sfh_amd.reconstructor import Reconstructor

dev = torch.device("cuda", 0)
B, W, H = 16, 640, 360
court = synth.load_court_template("ncaa_nc4_640x360", 4, B).to(dev)
poi = synth.load_court_poi("pitch", B).to(dev)
net = Reconstructor(court, poi, target_size=(W, H), unet_size=(W, H), warp_size=(W, H), warp_with_nearest=True)
net.load_state_dict(synth.synth_state_dict(net.state_dict(), 0))
net.to(dev).eval()
net.range_guard = False
x = synth.frames_to_float(synth.synth_frames_u8(B, H, W, seed=1)).to(dev)
with torch.no_grad():
    for _ in range(3):
        net.predict(x, consistency=False)
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(20):
        net.predict(x, consistency=False)
    pr.disable()
    torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
