"""Data-dependent clock check: the 512->512 45x80 S3 conv on random vs all-zero operands."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from sfh_amd import engine as E
def bench(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
B, cin, cout, h, w = 16, 512, 512, 45, 80
for zero in (False, True):
    wt = torch.zeros(cout, cin, 3, 3, device="cuda") if zero else torch.randn(cout, cin, 3, 3, device="cuda") * 0.02
    x = torch.zeros(B, h, w, cin, device="cuda") if zero else torch.relu(torch.randn(B, h, w, cin, device="cuda"))
    pc = E.PackedConv(wt, torch.zeros(cout, device="cuda"), None, 3, cin, s3=True)
    xs = E.f32_to_s3(x); y = E.s3_empty(B, h, w, cout, "cuda")
    for _ in range(20): pc.run(xs, B, h, w, y)   # sustained load before timing
    t = bench(lambda: pc.run(xs, B, h, w, y), reps=20)
    print("zero" if zero else "random", f"{t:.3f} ms {2.0*B*h*w*cout*9*cin/t/1e9:.1f} TF-equiv")
