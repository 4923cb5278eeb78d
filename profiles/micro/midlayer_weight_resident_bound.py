"""The long-K mid layers of the UNet (d3.x at 45x80, d4.x at 22x40, batch 16) pull 2.3 - 5x their tensors over the fabric: every pixel
tile streams the layer's whole weight set (9.4 / 37.7 MB) through its XCD's 4 MB L2 (profiles/r06_tcc_per_launch.txt).  Are they
waiting for it?  UPPER BOUND of any weight-stationary order, without building one: the same kernel, the same MFMA work, with
weights that fit the L2s - 1/8 or 1/16 of the couts over 8x / 16x the frames.  GPU box only."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from sfh_amd import engine as E  # noqa: E402


def bench(fn, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def case(tag, cin, cout, B, h, w):
    torch.manual_seed(0)
    wt = torch.randn(cout, cin, 3, 3, device="cuda") * (2.0 / (9 * cin)) ** 0.5
    bn = torch.nn.BatchNorm2d(cout).cuda().eval()
    conv = E.PackedConv(wt, torch.zeros(cout, device="cuda"), bn, 3, cin, fmt="h2", tag="probe")
    x = E.f32_to_h2(torch.relu(torch.randn(B, h, w, cin, device="cuda")))
    y = E.split_empty("h2", B, h, w, cout, "cuda")
    ms = bench(lambda: conv.run(x, B, h, w, y))
    tf = 2.0 * B * h * w * cout * 9 * cin / ms / 1e9
    print(f"{tag:36s} {cin:5d} -> {cout:5d}  {B:4d} frames of {h}x{w}: {ms * 1e3:8.1f} us  {tf:7.1f} TFLOP/s   weights {cout * cin * 9 * 4 / 2 ** 20:6.1f} MB", flush=True)
    return ms


if __name__ == "__main__":
    for rep in range(2):
        a = case("d3.3 as the model runs it", 512, 512, 16, 45, 80)
        b = case("d3.3 bound: weights L2-resident", 512, 64, 128, 45, 80)
        c = case("d4.3 as the model runs it", 1024, 1024, 16, 22, 40)
        d = case("d4.3 bound: weights L2-resident", 1024, 64, 256, 22, 40)
        e = case("d4.0 as the model runs it", 512, 1024, 16, 22, 40)
        f = case("d4.0 bound: weights L2-resident", 512, 128, 128, 22, 40)
        print(f"  round {rep}: bound: d3.3 {1e3 * (a - b):+.1f} us, d4.3 {1e3 * (c - d):+.1f} us, d4.0 {1e3 * (e - f):+.1f} us per batch")
