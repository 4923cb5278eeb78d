"""sfh_s2d_split_colsum at the four Up levels of 640x360 x 16 against sfh_bn_apply on a tensor of the same bytes (its bandwidth yardstick).
usage: [SFH_AMD_LIB=<variant .so>] [S2D_ROWS=32] python profiles/micro/s2d_cost_probe.py   (S2D_ROWS: rows of the column-sum table the
workgroups spread their fp64 atomics over; build variants -DSFH_S2D_BLOCKS=<n>: another number of workgroups)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import sfh_amd  # noqa
from sfh_amd import _lib, engine as E
from sfh_amd.engine import _ptr, _stream
lib = _lib.load()
def bench(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
B = 16
ROWS = int(os.environ.get("S2D_ROWS", "32"))
for cout, h, w in ((64, 180, 320), (128, 90, 160), (256, 45, 80), (512, 22, 40)):
    du = torch.randn(B, 2 * h, 2 * w, cout, device="cuda")
    s = E.split_empty("h2", B, h, w, 4 * cout, "cuda")
    acc = torch.zeros(ROWS, cout, dtype=torch.float64, device="cuda")
    t = bench(lambda: _lib.check(lib.sfh_s2d_split_colsum(_ptr(du), B, h, w, cout, _ptr(s), _lib.FMT_H2, _ptr(acc), ROWS, None, _stream()), "s2d"))
    ys = E.split_empty("h2", B, 2 * h, 2 * w, cout, "cuda")
    mi = torch.cat([torch.zeros(cout), torch.ones(cout)]).cuda(); g = torch.ones(cout, device="cuda"); bb = torch.zeros(cout, device="cuda")
    t2 = bench(lambda: _lib.check(lib.sfh_bn_apply(_ptr(du), _ptr(mi), _ptr(g), _ptr(bb), None, 1, B * 4 * h * w, cout, None, _ptr(ys), 2 * w, _lib.FMT_H2, None, _stream()), "bn_apply"))
    gb = du.numel() * 8 / 1e9
    print(f"cout {cout:4d} {h}x{w}: s2d_split_colsum {t * 1e3:7.1f} us = {gb / t:5.2f} TB/s   bn_apply (same bytes) {t2 * 1e3:7.1f} us = {gb / t2:5.2f} TB/s", flush=True)
