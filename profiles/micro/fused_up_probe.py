"""Debug probe: composed Up weights and the 2x2 quadrant conv against a torch restatement."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn.functional as F
from sfh_amd import _lib, engine as E
from sfh_amd.engine import _ptr, _stream, PackedConv
lib = _lib.load()
g = torch.Generator().manual_seed(1)
B, h, w, cx, c1, c0, cout = 2, 6, 10, 128, 64, 64, 64
up = torch.nn.ConvTranspose2d(cx, c1, 2, stride=2)
conv = torch.nn.Conv2d(c0 + c1, cout, 3, padding=1)
bn = torch.nn.BatchNorm2d(cout)
with torch.no_grad():
    bn.weight.uniform_(0.5, 1.5, generator=g); bn.bias.uniform_(-.3, .3, generator=g)
    bn.running_mean.uniform_(-.1, .1, generator=g); bn.running_var.uniform_(.5, 1.5, generator=g)
bn.eval()
x = torch.randn(B, cx, h, w, generator=g)
skip = torch.randn(B, c0, 2 * h, 2 * w, generator=g)
with torch.no_grad():
    want = torch.relu(bn(conv(torch.cat([skip, up(x)], 1))))
    part_ref = F.conv2d(skip, conv.weight[:, :c0], None, padding=1)
for m in (up, conv, bn):
    m.cuda()
fu = PackedConv.fused_up(conv, bn, up, c0)
torch.cuda.synchronize()
# (1) composed weights vs a torch restatement
wc, wt = conv.weight.detach().cpu(), up.weight.detach().cpu()
w2 = torch.zeros(4 * cout, cx, 2, 2)
for py in range(2):
    for px in range(2):
        q = py * 2 + px
        for ky in range(3):
            ty = py + ky - 1; sy = -1 if ty < 0 else ty >> 1; dy = ty & 1; a = sy + 1 - py
            for kx in range(3):
                tx = px + kx - 1; sx = -1 if tx < 0 else tx >> 1; dx = tx & 1; b = sx + 1 - px
                w2[q * cout:(q + 1) * cout, :, a, b] += torch.einsum("ou,xu->ox", wc[:, c0:, ky, kx], wt[:, :, dy, dx])
# reference result through w2: out[2Y+py][2X+px] = sum_ab w2[q][...][a][b] x[Y+py-1+a][X+px-1+b]
xp = F.pad(x, [1, 1, 1, 1])
ref2 = torch.zeros(B, cout, 2 * h, 2 * w)
for py in range(2):
    for px in range(2):
        q = py * 2 + px
        o = F.conv2d(xp[:, :, py:py + h + 1, px:px + w + 1], w2[q * cout:(q + 1) * cout])   # (B,cout,h,w)
        ref2[:, :, py::2, px::2] = o
with torch.no_grad():
    upart = F.conv2d(up.cpu()(x) - up.bias.view(1, -1, 1, 1).cpu(), wc[:, c0:], None, padding=1)
print("composition identity (torch only): %.2e" % (ref2 - upart).abs().max().item())
# (2) the HIP pair
sk = E.f32_to_s3(skip.permute(0, 2, 3, 1).contiguous().cuda())
xl = E.f32_to_s3(x.permute(0, 2, 3, 1).contiguous().cuda())
k1 = PackedConv(conv.weight.detach()[:, :c0].contiguous(), None, None, 3, c0, relu=False, s3=True)
k1.scale.copy_(fu.scale[:cout])
part = torch.empty((B, 2 * h, 2 * w, cout), device="cuda")
k1.run(sk, B, 2 * h, 2 * w, part)
torch.cuda.synchronize()
print("K1 partial err: %.2e" % (part.permute(0, 3, 1, 2).cpu() - part_ref * fu.scale[:cout].cpu().view(1, -1, 1, 1)).abs().max().item())
out = E.s3_empty(B, 2 * h, 2 * w, cout, "cuda")
fu.run(xl, B, h, w, out, residual=part)
torch.cuda.synchronize()
got = E.s3_to_f32(out).permute(0, 3, 1, 2).cpu()
err = (got - want).abs()
print("fused err: %.2e (max |want| %.2f)" % (err.max().item(), want.abs().max().item()))
e2 = err.amax(dim=(0, 1))
print("err by row:", [round(v, 3) for v in e2.amax(dim=1).tolist()])
print("err by col:", [round(v, 3) for v in e2.amax(dim=0).tolist()])
# ---- which window offsets / quadrant order does the kernel actually implement?
zero = torch.zeros_like(part)
fu.relu = False
out2 = E.s3_empty(B, 2 * h, 2 * w, cout, "cuda")
ones_s, zero_s = fu.scale.clone(), fu.shift.clone()
fu.scale.fill_(1.0); fu.shift.zero_(); fu.shift_border = None
fu.run(xl, B, h, w, out2, residual=zero)
torch.cuda.synchronize()
raw = E.s3_to_f32(out2).permute(0, 3, 1, 2).cpu()      # should equal ref2
print("raw K2 vs ref2: %.2e" % (raw - ref2).abs().max().item())
import itertools
xp2 = F.pad(x, [2, 2, 2, 2])
for py in range(2):
    for px in range(2):
        best = None
        for qq, oy, ox in itertools.product(range(4), range(-1, 2), range(-1, 2)):
            o = F.conv2d(xp2[:, :, 1 + oy + py:1 + oy + py + h + 1, 1 + ox + px:1 + ox + px + w + 1], w2[qq * cout:(qq + 1) * cout])
            e = (raw[:, :, py::2, px::2] - o).abs().max().item()
            if best is None or e < best[0]:
                best = (e, qq, oy, ox)
        print("quadrant (%d,%d): best match err %.2e with weights of q=%d, extra offset (%d,%d)" % ((py, px) + best))
