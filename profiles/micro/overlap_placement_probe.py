"""Where under the NEXT batch's UNet should the ResNet-STN launches of a batch run?  Today (predict_async) all of them start with
the UNet's first launches (inc.*, d1.*: the HBM-heavy ones).  This probe replays the recorded launch lists of the two engines
on two streams and moves the second half of the ResNet (layer3 / layer4 + head) behind an event recorded before a chosen UNet
launch.  Prints ms per (UNet + ResNet) pair; no result checking (the same buffers are rewritten every iteration)."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
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
    net.predict(x, consistency=False)
torch.cuda.synchronize()
un, rn = net._engines
U, R = [fn for _, fn in un.steps], [fn for _, fn in rn.steps]
print("UNet launches", len(U), "ResNet launches", len(R), flush=True)
side = torch.cuda.Stream(dev)
main = torch.cuda.current_stream(dev)


def run(iters, split_at, second_before):
    """ResNet launches [0, split_at) start with the UNet's first launch; [split_at, end) wait for the event recorded before
    UNet launch `second_before` (None: no split)."""
    for _ in range(iters):
        start = torch.cuda.Event(); start.record(main)
        side.wait_event(start)
        with torch.cuda.stream(side):
            for fn in R[:split_at]:
                fn()
        for i, fn in enumerate(U):
            if second_before is not None and i == second_before:
                ev = torch.cuda.Event(); ev.record(main)
                side.wait_event(ev)
                with torch.cuda.stream(side):
                    for g in R[split_at:]:
                        g()
            fn()
        if second_before is None:
            with torch.cuda.stream(side):
                for g in R[split_at:]:
                    g()
        done = torch.cuda.Event(); done.record(side)
        main.wait_event(done)     # (the pair is complete before the next pair starts: the pipeline's steady state)


def timeit(**kw):
    with torch.no_grad():
        run(3, **kw); torch.cuda.synchronize(); t = time.perf_counter(); run(20, **kw); torch.cuda.synchronize()
    return (time.perf_counter() - t) / 20 * 1e3


with torch.no_grad():
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(10):
        for fn in U: fn()
    torch.cuda.synchronize(); tu = (time.perf_counter() - t) / 10 * 1e3
    t = time.perf_counter()
    for _ in range(10):
        for fn in R: fn()
    torch.cuda.synchronize(); tr = (time.perf_counter() - t) / 10 * 1e3
print(f"alone: UNet {tu:.3f} ms, ResNet {tr:.3f} ms, sum {tu + tr:.3f}", flush=True)
nR = len(R)
for rnd in range(2):
    print(f"all ResNet launches from the UNet's first launch on: {timeit(split_at=nR, second_before=None):.3f} ms", flush=True)
    for split in (nR // 3, nR // 2, 2 * nR // 3):
        for before in (len(U) - 8, len(U) - 5, len(U) - 3, len(U) // 2):
            print(f"  ResNet[{split}:] behind an event before UNet launch {before}/{len(U)}: {timeit(split_at=split, second_before=before):.3f} ms", flush=True)
