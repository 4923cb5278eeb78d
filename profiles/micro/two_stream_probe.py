"""Do two UNets on two streams beat one?  Two models (own engines and workspaces), predict() alternately on two streams
without the per-call range read-back, against the same number of batches on one stream.  GPU box only."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from sfh_amd import synth  # noqa: E402
from sfh_amd.reconstructor import Reconstructor  # noqa: E402

B, W, H = 16, 640, 360
dev = torch.device("cuda", 0)
court = synth.load_court_template("ncaa_nc4_640x360", 4, B).to(dev)
poi = synth.load_court_poi("pitch", B).to(dev)
nets = []
for _ in range(2):
    net = Reconstructor(court, poi, target_size=(W, H), unet_size=(W, H), warp_size=(W, H), warp_with_nearest=True)
    net.load_state_dict(synth.synth_state_dict(net.state_dict(), 0))
    net.to(dev).eval()
    net.range_guard = False
    nets.append(net)
x = [synth.frames_to_float(synth.synth_frames_u8(B, H, W, seed=k)).to(dev) for k in range(2)]
streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]


def run(n, two):
    with torch.no_grad():
        for k in range(n):
            s = streams[k % 2] if two else streams[0]
            with torch.cuda.stream(s):
                nets[k % 2].predict(x[k % 2], consistency=False)
    torch.cuda.synchronize()


for two in (False, True, False, True):
    run(4, two)
    t0 = time.perf_counter()
    run(20, two)
    el = time.perf_counter() - t0
    print(f"{'two streams' if two else 'one stream '}: {B * 20 / el:8.1f} frames/s  {el / 20 * 1e3:.3f} ms per batch", flush=True)
