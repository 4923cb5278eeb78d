"""How much of a 3x3 H2 conv launch's time is the partly filled last round of workgroups, and how much is the
per-workgroup prologue / epilogue?  One layer shape (cin -> cout) is run at a ladder of frame heights, i.e. of
grid sizes from a fraction of a round to tens of rounds (512 resident workgroups = two per CU): the TFLOP/s at
many full rounds is the rate the stage loop + prologue + epilogue sustain, the dips are the tail.  GPU box only.
usage: python profiles/micro/conv_rate_probe.py [--wg 0|64|128]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from sfh_amd import engine as E  # noqa: E402


def bench(fn, reps=6):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def run(cin, cout, w, heights, B=16, wg=0, pool=False, tile=None):
    torch.manual_seed(0)
    wt = torch.randn(cout, cin, 3, 3, device="cuda") * (2.0 / (9 * cin)) ** 0.5
    bias = torch.randn(cout, device="cuda") * 0.1
    bn = torch.nn.BatchNorm2d(cout).cuda().eval()
    conv = E.PackedConv(wt, bias, bn, 3, cin, fmt="h2", tag="probe")
    for h in heights:
        x = E.f32_to_h2(torch.relu(torch.randn(B, h, w, cin, device="cuda")))
        y = E.split_empty("h2", B, h, w, cout, "cuda")
        yp = E.split_empty("h2", B, h // 2, w // 2, cout, "cuda") if pool else None
        ms = bench(lambda: conv.run(x, B, h, w, y, dst_pool=yp, wg_couts=wg, tile=tile))
        zr = 1 + ((h + 1) & 1)
        tl = E.choose_tile_s3(B, h, w, 1, zr, cout // 64) if tile is None else tile
        th, tw = {0: (8, 32), 1: (16, 16), 2: (32, 8), 3: (8, 16), 4: (16, 8)}[tl]
        ntiles = -(-(B * (h + zr)) // th) * -(-w // tw)
        tf = 2.0 * B * h * w * cout * 9 * cin / ms / 1e9
        print(f"{cin:5d}->{cout:<5d} {h:4d}x{w:<4d} tile {th}x{tw} tiles {ntiles:6d} x {cout // 64} blocks of 64 "
              f"{ms:8.3f} ms {tf:7.1f} TFLOP/s", flush=True)
        del x, y


if __name__ == "__main__":
    wg = int(sys.argv[sys.argv.index("--wg") + 1]) if "--wg" in sys.argv else 0
    # the shapes of the model (d3.3 / u1.3, d2.3, d1.3, inc.3) and taller / shorter frames of the same width
    if "--w8-half" not in sys.argv:
        run(512, 512, 80, (11, 22, 45, 90, 180, 360), wg=wg)
        run(256, 256, 160, (22, 45, 90, 180, 360), wg=wg)
        run(128, 128, 320, (45, 90, 180, 360), wg=wg)
        run(64, 64, 640, (90, 180, 360), wg=wg)
        run(64, 64, 640, (360,), wg=wg, pool=True)
    if "--w8-half" in sys.argv:   # 128-pixel x 128-cout double-buffered workgroups (wg=128 + half-size tile) against the other shapes
        print("128 x 128 workgroups, two LDS buffers (tile 8x16)")
        run(512, 512, 80, (45, 360), wg=128, tile=3)
        run(256, 256, 160, (90, 360), wg=128, tile=3)
        run(128, 128, 320, (180, 360), wg=128, tile=3)
        print("128 x 64 workgroups (tile 8x16; double-buffered only in the experiment build recorded in r03_conv_rate_probe_w8half.txt)")
        run(64, 64, 640, (360,), wg=64, tile=3)
        run(64, 64, 640, (360,), wg=64, tile=3, pool=True)
        run(128, 128, 320, (180,), wg=64, tile=3)
        run(256, 256, 160, (90,), wg=64, tile=3)
        run(512, 512, 80, (45,), wg=64, tile=3)
        print("64 -> 128 (d1.0): 128 x 128 double-buffered against the product shape")
        run(64, 128, 320, (180,), wg=128, tile=3)
        run(64, 128, 320, (180,))
        run(64, 128, 320, (180,), wg=128, tile=3)
        run(64, 128, 320, (180,))
        print("the same layers, product shapes")
        run(512, 512, 80, (45, 360), wg=128)
        run(256, 256, 160, (90, 360), wg=128)
        run(128, 128, 320, (180, 360), wg=128)
        run(64, 64, 640, (360,))
        run(64, 64, 640, (360,), pool=True)
    if "--half-tiles" in sys.argv:   # 128-pixel tiles: half the LDS and accumulators per workgroup, more workgroups per CU
        print("half-size tiles (8x16)")
        run(64, 64, 640, (360,), tile=3)
        run(64, 64, 640, (360,), tile=3, pool=True)
        run(128, 128, 320, (180,), wg=64, tile=3)
        run(128, 128, 320, (180,), wg=64)
