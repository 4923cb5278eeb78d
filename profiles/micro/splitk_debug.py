"""debug: split-K partial slabs vs the unsplit fp32 output (GPU box)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from sfh_amd import engine as E, synth

for fmt in ("s3", "h2"):
    g = synth._rng(12, "dbg")
    B, H, W, cin, cout = 2, 12, 20, 256, 128
    w = torch.from_numpy((g.normal(0, 1, (cout, cin, 3, 3)) * (2.0 / (9 * cin)) ** 0.5).astype(np.float32)).cuda()
    bn = torch.nn.BatchNorm2d(cout).cuda().eval()
    with torch.no_grad():
        bn.running_mean.uniform_(-0.1, 0.1); bn.running_var.uniform_(0.5, 1.5); bn.weight.uniform_(0.5, 1.5); bn.bias.uniform_(-0.3, 0.3)
    x = torch.from_numpy(g.normal(0, 1, (B, H, W, cin)).astype(np.float32)).cuda()
    res = torch.from_numpy(g.normal(0, 1, (B, H, W, cout)).astype(np.float32)).cuda()
    xs = E.f32_to_split(x, fmt)
    for use_bn in (False, True):
        for relu in (False, True):
            for use_res, er in ((False, 2), (True, 2), (True, 1)):
                for ed in (2, 0):
                    pc = E.PackedConv(w, None, bn if use_bn else None, 3, cin, relu=relu, fmt=fmt)
                    rs = E.f32_to_split(res, fmt, exp=er) if use_res else None
                    outs = []
                    for ks in (0, 3):
                        y = E.split_empty(fmt, B, H, W, cout, "cuda")
                        slabs = torch.full((ks, B, H, W, cout), float("nan"), device="cuda") if ks else None
                        pc.run(xs, B, H, W, y, residual=rs, exp_res=er, exp_dst=ed, ksplit=ks, slabs=slabs)
                        outs.append(E.s3_to_f32(y, exp=ed))
                    torch.cuda.synchronize()
                    print(fmt, "bn", use_bn, "relu", relu, "res", use_res, "exp_res", er, "exp_dst", ed, "maxdiff", float((outs[0] - outs[1]).abs().max()), flush=True)
