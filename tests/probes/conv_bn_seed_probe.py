"""Which quantity of tests/test_gpu_training.py::test_conv_bn_act_backward exceeded its 2e-5 bound for some draws
of the default-initialised nn.Conv2d (round 1 closed that by seeding torch's global RNG)?  For a set of fixed
seeds: max-abs-difference / max-abs-reference (the test's metric) of every checked quantity, for the HIP path in
both train precisions and for torch's own CPU fp32 autograd, all against the fp64 oracle.
usage: python tests/probes/conv_bn_seed_probe.py [nseeds]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch


class H(torch.nn.Module):
    def __init__(self, stride, ks):
        super().__init__()
        self.conv = torch.nn.Conv2d(64, 128, ks, stride=stride, padding=ks // 2, bias=(stride == 1))
        self.bn = torch.nn.BatchNorm2d(128)


def rel(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()


def run(seed, stride, ks, res, shape, precision):
    from sfh_amd import training as T
    os.environ["SFH_TRAIN_PRECISION"] = precision
    B, Hh, W = shape
    torch.manual_seed(seed)
    g = torch.Generator().manual_seed(11 + stride + ks)
    m = H(stride, ks)
    with torch.no_grad():
        m.bn.weight.uniform_(0.5, 1.5, generator=g)
        m.bn.bias.uniform_(-0.3, 0.3, generator=g)
    x = torch.randn(B, 64, Hh, W, generator=g)
    ho, wo = (Hh - 1) // stride + 1, (W - 1) // stride + 1
    r = torch.randn(B, 128, ho, wo, generator=g) if res else None
    dy = torch.randn(B, 128, ho, wo, generator=g)
    out = {}
    refs = {}
    for tag, dt in (("f64", torch.float64), ("cpu32", torch.float32)):
        mm = H(stride, ks).to(dt)
        mm.load_state_dict({k: v.to(dt) for k, v in m.state_dict().items()})
        mm.train()
        xr = x.to(dt).requires_grad_(True)
        rr = r.to(dt).requires_grad_(True) if res else None
        z = mm.conv(xr)
        y = mm.bn(z)
        y = torch.relu(y + rr if res else y)
        y.backward(dy.to(dt))
        q = {"y": y, "dx": xr.grad, **{k: p.grad for k, p in mm.named_parameters() if k != "conv.bias"}}
        if res:
            q["dres"] = rr.grad
        q["z_absmin_var"] = z.detach().var(dim=(0, 2, 3), unbiased=False).min()
        refs[tag] = q
    m.cuda().train()
    tape = T.Tape()
    nh = lambda t: t.permute(0, 2, 3, 1).contiguous().cuda()
    nc = lambda t: t.permute(0, 3, 1, 2).cpu()
    xs, rs = nh(x), (nh(r) if res else None)
    y = T.conv_bn_act(tape, T._Names(m), m.conv, m.bn, [(xs, 64, 0, 0)], B, Hh, W, residual=rs)
    tape.add_grad(y, nh(dy))
    tape.backward()
    torch.cuda.synchronize()
    gq = {"y": nc(y), "dx": nc(tape.pop_grad(xs)), **{k: v for k, v in tape.param_grads.items() if k != "conv.bias"}}
    cb = tape.param_grads.get("conv.bias")
    if res:
        gq["dres"] = nc(tape.pop_grad(rs))
    for k in gq:
        out[k] = (rel(gq[k], refs["f64"][k]), rel(refs["cpu32"][k], refs["f64"][k]))
    if cb is not None:   # the test bounds this one absolutely (exact value: 0)
        out["conv.bias(abs)"] = (float(cb.abs().max()), 0.0)
    out["min_channel_var"] = float(refs["f64"]["z_absmin_var"])
    return out


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    worst = {}
    for prec in ("bf16x6", "fp32"):
        for (stride, ks, res) in [(1, 3, False), (1, 3, True), (2, 3, False), (2, 1, False), (1, 1, True)]:
            for shape in [(2, 13, 18), (3, 8, 40)]:
                for seed in range(n):
                    o = run(1000 + seed, stride, ks, res, shape, prec)
                    for k, v in o.items():
                        if k == "min_channel_var":
                            continue
                        key = (prec, k)
                        if key not in worst or v[0] > worst[key][0]:
                            worst[key] = (v[0], v[1], dict(seed=1000 + seed, stride=stride, ks=ks, res=res, shape=shape,
                                                           min_channel_var=o["min_channel_var"]))
    for (prec, k), (e, e32, where) in sorted(worst.items()):
        print(json.dumps({"train_precision": prec, "quantity": k, "worst_hip_vs_f64": e, "cpu_fp32_vs_f64_same_case": e32,
                          "case": where}))
