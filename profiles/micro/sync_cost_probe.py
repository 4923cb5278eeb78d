"""What does the per-call read-back of the fp16-range word cost, and which way of waiting is cheapest?
predict() B=16 640x360 back to back, ending each call with: nothing / .item() of a device int32 / torch.cuda.synchronize() /
event record + query spin / event.synchronize()."""
import os, sys, time
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
flag = torch.zeros(1, dtype=torch.int32, device=dev)
ev = torch.cuda.Event()


def bench(tail, n=30):
    with torch.no_grad():
        for _ in range(3):
            net.predict(x, consistency=False); tail()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            net.predict(x, consistency=False); tail()
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def spin():
    ev.record()
    while not ev.query():
        pass


def evsync():
    ev.record()
    ev.synchronize()


for name, tail in (("nothing", lambda: None), ("item()", lambda: flag.item()), ("synchronize()", torch.cuda.synchronize),
                   ("event query spin", spin), ("event.synchronize()", evsync), ("nothing", lambda: None)):
    print(f"{name:22s} {bench(tail):7.3f} ms per call", flush=True)
# CPU time of one predict() call (launch work only)
with torch.no_grad():
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    net.predict(x, consistency=False)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
print(f"host time to enqueue one predict(): {(t1 - t0) * 1e3:.3f} ms")
