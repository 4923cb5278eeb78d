"""Repeat the conv_bn_act forward/backward unit case in one process (with allocator churn in between) and
report which quantity differs from the first iteration - looking for run-to-run nondeterminism."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from sfh_amd import training as T

class H(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.conv = torch.nn.Conv2d(64, 128, 3, padding=1)
        self.bn = torch.nn.BatchNorm2d(128)

g = torch.Generator().manual_seed(15)
m = H()
with torch.no_grad():
    m.bn.weight.uniform_(0.5, 1.5, generator=g); m.bn.bias.uniform_(-.3, .3, generator=g)
x = torch.randn(2, 64, 13, 18, generator=g)
dy = torch.randn(2, 128, 13, 18, generator=g)
m.cuda().train()
xs = x.permute(0, 2, 3, 1).contiguous().cuda()
dys = dy.permute(0, 2, 3, 1).contiguous().cuda()
ref = None
bad = 0
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 40):
    # poison freed blocks of the sizes this case allocates (small and large allocator pools)
    junk = [torch.full((n,), float("nan"), device="cuda") for n in (1 << 20, 1 << 18, 60000, 30000, 15000, 4096, 1024, 256) for _ in range(6)]
    del junk
    tape = T.Tape()
    y = T.conv_bn_act(tape, T._Names(m), m.conv, m.bn, [(xs, 64, 0, 0)], 2, 13, 18)
    tape.add_grad(y, dys.clone())
    tape.backward()
    torch.cuda.synchronize()
    cur = {"y": y.clone(), "dx": tape.pop_grad(xs).clone(), **{k: v.clone() for k, v in tape.param_grads.items()}}
    if ref is None:
        ref = cur
        continue
    for k in ref:
        d = (cur[k] - ref[k]).abs().max().item()
        s = ref[k].abs().max().item() + 1e-30
        if not (d / s < 1e-5):
            bad += 1
            print("iter %d: %s differs: max abs %.3e (scale %.3e), nan=%s" % (it, k, d, s, torch.isnan(cur[k]).any().item()))
print("done, %d deviations" % bad)
