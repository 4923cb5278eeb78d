"""Is every 3x3 layer of the C2 UNet on its best workgroup shape?  Times each layer shape (cin -> cout at its frame size, batch 16)
with every (tile, workgroup) combination the launcher offers - full-size tiles 8x32 / 16x16 / 32x8 with 64- or 128-cout
workgroups, and the 128 x 128 double-buffered half-size tile - next to what the engine picks by itself.  GPU box only."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from sfh_amd import engine as E, _lib  # noqa: E402


def bench(fn, reps=5):
    fn()
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


LAYERS = [  # name, cin, cout, H, W, pooled output
    ("inc.3", 64, 64, 360, 640, True), ("d1.0", 64, 128, 180, 320, False), ("d1.3", 128, 128, 180, 320, True),
    ("d2.0", 128, 256, 90, 160, False), ("d2.3", 256, 256, 90, 160, True), ("d3.0", 256, 512, 45, 80, False),
    ("d3.3", 512, 512, 45, 80, True), ("d4.0", 512, 1024, 22, 40, False), ("d4.3", 1024, 1024, 22, 40, False),
    ("u1.3", 512, 512, 45, 80, False), ("u2.3", 256, 256, 90, 160, False), ("u3.3", 128, 128, 180, 320, False),
    ("u4.3", 64, 64, 360, 640, False),
]
if "--c5" in sys.argv:   # the same layers at 1280x720 (BASELINE config 5): every frame size doubled
    LAYERS = [(n, ci, co, 2 * h, 2 * w, p) for (n, ci, co, h, w, p) in LAYERS]
COMBOS = [("8x32/64", 0, 64), ("16x16/64", 1, 64), ("32x8/64", 2, 64), ("8x32/128", 0, 128), ("16x16/128", 1, 128),
          ("32x8/128", 2, 128), ("8x16/128db", 3, 128)]
B = 16
for name, cin, cout, h, w, pool in LAYERS:
    torch.manual_seed(0)
    wt = torch.randn(cout, cin, 3, 3, device="cuda") * (2.0 / (9 * cin)) ** 0.5
    bn = torch.nn.BatchNorm2d(cout).cuda().eval()
    conv = E.PackedConv(wt, torch.zeros(cout, device="cuda"), bn, 3, cin, fmt="h2", tag="probe")
    x = E.f32_to_h2(torch.relu(torch.randn(B, h, w, cin, device="cuda")))
    y = E.split_empty("h2", B, h, w, cout, "cuda")
    yp = E.split_empty("h2", B, h // 2, w // 2, cout, "cuda") if pool else None
    res = {}
    bench(lambda: conv.run(x, B, h, w, y, dst_pool=yp))   # (the first measurement of a layer runs 5-10 % slow: discarded)
    for label, tile, wg in COMBOS:
        if wg == 128 and (cout % 128 or cin < (64 if tile == 3 else 128)):
            continue
        try:
            res[label] = bench(lambda: conv.run(x, B, h, w, y, dst_pool=yp, tile=tile, wg_couts=wg))
        except Exception as e:   # a shape the launcher does not offer for this layer
            res[label] = None
    res["engine"] = bench(lambda: conv.run(x, B, h, w, y, dst_pool=yp))
    fl = 2.0 * B * h * w * cout * 9 * cin
    best = min((v, k) for k, v in res.items() if v)
    print(f"{name:6s} {cin:4d}->{cout:<4d} {h}x{w}: " + "  ".join(
        f"{k} {fl / v / 1e9:5.0f}" if v else f"{k}   -  " for k, v in res.items()) + f"   best {best[1]}"
        + ("" if res['engine'] <= best[0] * 1.015 else f"  (engine {100 * (res['engine'] / best[0] - 1):.1f} % slower)"), flush=True)
    del x, y, yp
