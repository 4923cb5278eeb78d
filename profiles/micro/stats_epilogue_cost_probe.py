"""What the BatchNorm-sums epilogue of the training convs costs: the same H2 conv (fp32 destination, no ReLU: the training
forward's z = conv + bias) with and without sfh_conv_desc.stats_partial, and the backward-data form (bwd_z) - per layer shape
of the UNet at 640x360 x 16.   usage: python profiles/micro/stats_epilogue_cost_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import sfh_amd  # noqa
from sfh_amd import engine as E

def bench(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

B = 16
for cin, cout, h, w in ((64, 64, 360, 640), (128, 128, 180, 320), (256, 256, 90, 160), (512, 512, 45, 80), (1024, 1024, 22, 40)):
    torch.manual_seed(0)
    wt = torch.randn(cout, cin, 3, 3, device="cuda") * (2.0 / (9 * cin)) ** 0.5
    conv = E.PackedConv(wt, None, None, 3, cin, relu=False, fmt="h2", tag="probe", shared_unit_scale=True)
    x = E.f32_to_h2(torch.relu(torch.randn(B, h, w, cin, device="cuda")))
    z = torch.empty(B, h, w, cout, device="cuda")
    rows = 64
    while rows < 2048 and rows * 1024 < B * h * w: rows *= 2
    table = torch.zeros(rows, 2, cout, dtype=torch.float64, device="cuda")
    zprev = torch.randn(B, h, w, cout, device="cuda")
    mi = torch.cat([torch.zeros(cout), torch.ones(cout)]).cuda()
    gam, bet = torch.ones(cout, device="cuda"), torch.zeros(cout, device="cuda")
    t0 = bench(lambda: conv.run(x, B, h, w, z))
    t1 = bench(lambda: conv.run(x, B, h, w, z, stats=table))
    t2 = bench(lambda: conv.run(x, B, h, w, z, stats=table, bwd=(zprev, mi, gam, bet)))
    gf = 2.0 * B * h * w * cout * 9 * cin / 1e9
    print(f"{cin:5d}->{cout:<5d} {h:3d}x{w:<3d}  plain {t0:7.3f} ms ({gf / t0:6.1f} TFLOP/s)   + forward sums {t1:7.3f} ms ({100 * (t1 / t0 - 1):+5.1f} %)   + backward sums {t2:7.3f} ms ({100 * (t2 / t0 - 1):+5.1f} %)", flush=True)
