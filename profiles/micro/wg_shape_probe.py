"""64-cout (4 waves as 2 x 2, three workgroups per CU) against 128-cout (1 x 4, two per CU) workgroups on the model's
own 3x3 layer shapes (H2 operands, batch 16).  GPU box only."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from sfh_amd import engine as E  # noqa: E402
from conv_rate_probe import bench  # noqa: E402

SHAPES = [(64, 128, 180, 320), (128, 128, 180, 320), (128, 256, 90, 160), (256, 256, 90, 160), (256, 512, 45, 80),
          (512, 512, 45, 80), (512, 1024, 22, 40), (1024, 1024, 22, 40)]
for cin, cout, h, w in SHAPES:
    torch.manual_seed(0)
    wt = torch.randn(cout, cin, 3, 3, device="cuda") * (2.0 / (9 * cin)) ** 0.5
    bn = torch.nn.BatchNorm2d(cout).cuda().eval()
    conv = E.PackedConv(wt, None, bn, 3, cin, fmt="h2", tag="probe")
    x = E.f32_to_h2(torch.relu(torch.randn(16, h, w, cin, device="cuda")))
    y = E.split_empty("h2", 16, h, w, cout, "cuda")
    res = []
    for wg in (64, 128, 64, 128):
        try:
            ms = bench(lambda: conv.run(x, 16, h, w, y, wg_couts=wg), reps=10)
        except ValueError:      # the 128-cout shape needs at least four 32-channel stages
            ms = float("inf")
        res.append(ms)
    tf = lambda ms: 2.0 * 16 * h * w * cout * 9 * cin / ms / 1e9
    print(f"{cin:5d}->{cout:<5d} {h:4d}x{w:<4d}  64-cout {res[0]:.3f} / {res[2]:.3f} ms ({tf(min(res[0], res[2])):6.1f} TFLOP/s)   "
          f"128-cout {res[1]:.3f} / {res[3]:.3f} ms ({tf(min(res[1], res[3])):6.1f} TFLOP/s)", flush=True)
