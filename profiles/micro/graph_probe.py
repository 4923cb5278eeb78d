"""Does capturing predict() in a HIP graph pay?  eager vs torch.cuda.CUDAGraph replay, B=16 640x360."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from sfh_amd import synth
from sfh_amd.reconstructor import Reconstructor

dev = torch.device("cuda", 0)
B, W, H = (int(sys.argv[1]) if len(sys.argv) > 1 else 16), 640, 360
court = synth.load_court_template("ncaa_nc4_640x360", 4, B).to(dev)
poi = synth.load_court_poi("pitch", B).to(dev)
net = Reconstructor(court, poi, target_size=(W, H), unet_size=(W, H), warp_size=(W, H), warp_with_nearest=True)
net.load_state_dict(synth.synth_state_dict(net.state_dict(), 0))
net.to(dev).eval()
x = synth.frames_to_float(synth.synth_frames_u8(B, H, W, seed=1)).to(dev)

def bench(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

with torch.no_grad():
    eager_guard = bench(lambda: net.predict(x, consistency=False))
    net.range_guard = False     # the read-back of the fp16-range word synchronises and cannot be captured
    eager = bench(lambda: net.predict(x, consistency=False))
    ref = net.predict(x, consistency=False)
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        net.predict(x, consistency=False)
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        out = net.predict(x, consistency=False)
    graph = bench(g.replay)
    g.replay()
    torch.cuda.synchronize()
    print("B=%d " % B + "precision %s: eager with range guard %.3f ms  eager %.3f ms  graph %.3f ms  same theta %s same mask %s overflow %s" % (
        net.precision, eager_guard, eager, graph, torch.equal(out["theta"], ref["theta"]), torch.equal(out["warp_mask"], ref["warp_mask"]), net.range_overflowed()))
