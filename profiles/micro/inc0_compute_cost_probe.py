"""Review item 6a (`inc.0` recomputed inside `inc.3`'s halo), the measurement behind the decision: what does the first layer's
arithmetic cost when its output does NOT go to HBM?  The first-layer kernel (sfh_conv3x3_c4h2_fwd: 3 -> 64 channels, BatchNorm,
ReLU, two-plane split = exactly the producer phase a fused inc.0 + inc.3 kernel would run on every halo) is timed on frame
sizes from 45x80 to 360x640 at batch 16: at the small sizes its 4-byte-per-element output (15-59 MB) stays in the 256 MB
Infinity Cache, so ns per output pixel there is the price of the MFMAs + epilogue alone; at 360x640 (0.94 GB) it is HBM-bound.
A fused kernel would pay the cache-resident price on 1.33x the pixels (18x34... 10x34 halo per 8x32 tile) inside a launch that
is already power- and issue-bound, to save inc.0's launch (0.22 ms) and <= 0.05 ms of inc.3's own DMA waits.
usage (GPU box): python profiles/micro/inc0_compute_cost_probe.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import ctypes  # noqa: E402
import torch  # noqa: E402
from sfh_amd import engine as E, _lib  # noqa: E402


def bench(fn, reps=20):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


torch.manual_seed(0)
lib = _lib.load()
B = 16
w = torch.randn(64, 3, 3, 3, device="cuda") * (2.0 / 27) ** 0.5
bn = torch.nn.BatchNorm2d(64).cuda().eval()
pc = E.PackedConv(w, torch.zeros(64, device="cuda"), bn, 3, 3, fmt=None, frame_h2=True)
per_px_cached = 1e9
for (h, wd) in ((45, 80), (90, 160), (180, 320), (360, 640)):
    x = torch.rand(B, 3, h, wd, device="cuda")
    xin = torch.empty(B, h, wd, 4, device="cuda")
    fh2 = torch.empty(B, h, wd, 4, device="cuda")
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    _lib.check(lib.sfh_frame_to_h2(ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(xin.data_ptr()), ctypes.c_void_p(fh2.data_ptr()),
                                   B, 3, h, wd, 2, None, None, st), "frame_to_h2")
    y = E.split_empty("h2", B, h, wd, 64, "cuda")
    ms = bench(lambda: pc.run(fh2, B, h, wd, y, exp_src=2))
    px = B * h * wd
    print(f"first layer 3 -> 64, {h}x{wd} x {B}: output {px * 64 * 4 / 1e6:7.1f} MB  {ms * 1e3:8.1f} us  {ms * 1e6 / px:6.3f} ns per pixel  "
          f"({px * 64 * 4 / ms / 1e9:5.2f} TB/s of output)", flush=True)
    if px * 64 * 4 < 250e6:      # output stays in the 256 MB Infinity Cache: the best such rate is the arithmetic's price
        per_px_cached = min(per_px_cached, ms * 1e6 / px)
full_px = B * 360 * 640
print(f"producer phase of a fused inc.0 + inc.3 kernel at the cache-resident rate: 1.33 x {full_px} pixels x {per_px_cached:.3f} ns = "
      f"{1.33 * full_px * per_px_cached / 1e6:.3f} ms of a fully occupied chip, against 0.22-0.26 ms (inc.0's launch) + <= 0.05 ms saved: "
      f"no net gain even before the producer phase competes with inc.3's own MFMAs and epilogue for the same issue slots")
