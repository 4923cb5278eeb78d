"""Training-mode parity: HIP forward/backward vs torch autograd over the CPU oracle."""
import warnings

import numpy as np
import pytest
import torch

from sfh_amd import modules, synth
from oracle import torch_ref, train_ref

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def T():
    from sfh_amd import training
    return training


def _relerr(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()


def _grad_stats(got, want):
    errs = {}
    for k, w in want.items():
        w = w.double()
        if k.endswith("double_conv.0.bias") or k.endswith("double_conv.3.bias") or w.norm() < 1e-12:
            continue   # conv bias in front of a batch-statistics BatchNorm: exactly zero gradient up to rounding
        errs[k] = ((got[k].detach().cpu().double() - w).norm() / w.norm()).item()
    return errs


def _check_grads(got, want, tol):
    assert sorted(got) == sorted(want)
    bad = {k: _relerr(got[k], want[k]) for k in want if _relerr(got[k], want[k]) > tol}
    assert not bad, f"{len(bad)} of {len(want)} gradients off: {dict(list(bad.items())[:6])}"


def _mini_net():
    from sfh_amd.reconstructor import Reconstructor
    net = Reconstructor(None, None, use_warper=False, use_resnet=False, target_size=(96, 64), unet_size=(96, 64))
    sd = synth.synth_state_dict(net.state_dict(), 41)
    net.load_state_dict(sd)
    return net, sd


def test_wgrad_kernel_vs_autograd(T):
    """backward-filter on the fp32 matrix cores: 3x3 two-source (concat + pad), 1x1, odd sizes."""
    from sfh_amd import _lib
    from sfh_amd.engine import _ptr, _stream
    lib = _lib.load()
    g = torch.Generator().manual_seed(3)
    B, H, W = 2, 11, 37
    x0 = torch.randn(B, 64, H, W, generator=g)
    x1 = torch.randn(B, 64, 8, 34, generator=g)
    dz = torch.randn(B, 128, H, W, generator=g)
    pt, pl = 1, 2
    for ks in (3, 1):
        xin = torch.cat([x0, torch.nn.functional.pad(x1, [pl, W - 34 - pl, pt, H - 8 - pt])], 1).requires_grad_(True)
        w = torch.zeros(128, 128, ks, ks, requires_grad=True)
        torch.nn.functional.conv2d(xin, w, padding=ks // 2).backward(dz)
        raw = T._wgrad(lib, dz.permute(0, 2, 3, 1).contiguous().cuda(),
                       [(x0.permute(0, 2, 3, 1).contiguous().cuda(), 64, 0, 0, 0),
                        (x1.permute(0, 2, 3, 1).contiguous().cuda(), 64, 64, pt, pl)], B, H, W, ks, 128)
        got = raw.view(128, ks, ks, 128).permute(0, 3, 1, 2)
        assert _relerr(got, w.grad) < 2e-5


@pytest.mark.parametrize("prec", ["bf16x6", "f16x3"])
@pytest.mark.parametrize("ks", [3, 1])
@pytest.mark.parametrize("shape", [(2, 11, 37), (1, 6, 70), (3, 9, 20), (2, 12, 8), (1, 40, 10)])
def test_wgrad_s3_kernel_vs_fp64(T, shape, prec, ks, monkeypatch):
    """backward-filter on the bf16 matrix cores (split-bf16 operands, transposing LDS reads): two-source
    (concat + pad), all three tile shapes, partial tiles, against an fp64 reference and the fp32-MFMA kernel."""
    from sfh_amd import _lib, engine as E
    monkeypatch.setenv("SFH_TRAIN_PRECISION", prec)
    lib = _lib.load()
    B, H, W = shape
    g = torch.Generator().manual_seed(5 + H)
    h1, w1, pt, pl = H - 3, W - 3, 1, 2
    x0 = torch.randn(B, 64, H, W, generator=g) * torch.exp(torch.randn(B, 64, H, W, generator=g))
    x1 = torch.randn(B, 32, h1, w1, generator=g)
    dz = torch.randn(B, 128, H, W, generator=g)
    xin = torch.cat([x0, torch.nn.functional.pad(x1, [pl, W - w1 - pl, pt, H - h1 - pt])], 1).double().requires_grad_(True)
    w = torch.zeros(128, 96, ks, ks, dtype=torch.float64, requires_grad=True)
    torch.nn.functional.conv2d(xin, w, padding=ks // 2).backward(dz.double())
    nh = lambda t: t.permute(0, 2, 3, 1).contiguous().cuda()
    tape = T.Tape()
    t0, t1, dzc = nh(x0), nh(x1), nh(dz)
    srcs = [(t0, 64, 0, 0, 0), (t1, 32, 64, pt, pl)]
    assert T.wgrad_s3_ok(ks, 1, 128, srcs)
    raw = T._wgrad_s3(lib, tape, E.f32_to_split(dzc, tape.fmt), 128, srcs, B, H, W, 96, ks)
    ref32 = T._wgrad(lib, dzc, srcs, B, H, W, ks, 96)
    torch.cuda.synchronize()
    got = raw.view(128, ks, ks, 96).permute(0, 3, 1, 2)
    old = ref32.view(128, ks, ks, 96).permute(0, 3, 1, 2)
    e_new, e_old = _relerr(got, w.grad), _relerr(old, w.grad)
    assert e_new < 2e-5, (e_new, e_old)
    assert e_new < 4 * e_old + 1e-6, (e_new, e_old)       # fp32-grade accuracy


def test_unet_train_forward_backward(T):
    net, sd = _mini_net()
    B, H, W = 4, 64, 96
    x = synth.frames_to_float(synth.synth_frames_u8(B, H, W, seed=41))
    g = torch.Generator().manual_seed(7)
    dlogits = torch.randn(B, 4, H, W, generator=g) / (H * W)

    # oracle: autograd over the functional restatement, BatchNorm in training mode
    ref = train_ref.leaf_state(sd)
    logits_ref, xtop_ref, _ = train_ref.forward_unet_train(x, ref, unet_size=(W, H), target_size=(W, H))
    logits_ref.backward(dlogits)

    net.cuda().train()
    tape = T.Tape()
    u = T.UNetTrainer(net).forward(tape, x.cuda())
    logits, xtop = u["logits"], tape.f32(u["x_top"])     # (x_top feeds a transposed conv only: split copy, no fp32 storage)
    u["heads"][0][1](dlogits.cuda())
    tape.backward()
    torch.cuda.synchronize()

    assert _relerr(logits, logits_ref) < 2e-5
    assert _relerr(xtop.permute(0, 3, 1, 2), xtop_ref) < 2e-5
    want = {k: v.grad for k, v in ref.items() if v.requires_grad}
    assert sorted(tape.param_grads) == sorted(want)
    # statistical check (see test_full_training_forward_backward for why); exact layer checks below
    errs = np.sort(np.array(list(_grad_stats(tape.param_grads, want).values())))
    assert np.median(errs) < 2e-2 and errs[int(0.8 * len(errs))] < 5e-2, (np.median(errs), errs[-5:])
    # running statistics and the batch counter advance like nn.BatchNorm2d(momentum=0.1)
    new = net.state_dict()
    for k, v in ref.items():
        if k.endswith("running_mean") or k.endswith("running_var"):
            assert _relerr(new[k], v) < 1e-5, k
        if k.endswith("num_batches_tracked"):
            assert int(new[k]) == int(sd[k]) + 1


# ------------------------------------------------------------------------------- single layers
class _Holder(torch.nn.Module):
    def __init__(self, **mods):
        super().__init__()
        for k, v in mods.items():
            setattr(self, k, v)


def _nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous().cuda()


def _nchw(t):
    return t.permute(0, 3, 1, 2).cpu()


@pytest.mark.parametrize("stride,ks,res", [(1, 3, False), (1, 3, True), (2, 3, False), (2, 1, False), (1, 1, True)])
@pytest.mark.parametrize("shape", [(2, 13, 18), (3, 8, 40)])
@pytest.mark.parametrize("seed", [20240917, 1063, 1092, 1094])
@pytest.mark.parametrize("prec", ["bf16x6", "f16x3"])
def test_conv_bn_act_backward(T, stride, ks, res, shape, seed, prec, monkeypatch):
    """conv (+bias) -> BatchNorm(train) (+residual) -> ReLU: outputs, input and parameter gradients.

    The conv's default initialisation draws from torch's global generator.  Round 1 saw this test fail once and
    closed it by seeding that generator; the cause (tests/probes/conv_bn_seed_probe.py, 100 seeds x 10 cases x 2
    precisions): in 3 of 2000 runs ONE pre-activation sits within rounding distance of zero, so the ReLU decision
    of the fp32-grade run and of the fp64 oracle differ for that element - the output moves by 1e-6, but that
    element's whole gradient toggles and the max-norm error of dx / dW jumps to 4e-3..5e-2.  torch's own CPU fp32
    autograd shows the same jump (seed 1092: 3e-1 on the residual gradient).  It is a property of ReLU, not of the
    kernels: the test therefore sends no gradient into elements whose fp64 pre-activation is closer to zero than
    1e-4, and the three seeds that hit such an element stay in the parametrisation.  Every other run of the sweep
    was below 1.6e-6 against the 2e-5 bound."""
    B, H, W = shape
    monkeypatch.setenv("SFH_TRAIN_PRECISION", prec)
    torch.manual_seed(seed)
    g = torch.Generator().manual_seed(11 + stride + ks)
    conv = torch.nn.Conv2d(64, 128, ks, stride=stride, padding=ks // 2, bias=(stride == 1))
    bn = torch.nn.BatchNorm2d(128)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5, generator=g)
        bn.bias.uniform_(-0.3, 0.3, generator=g)
    holder = _Holder(conv=conv, bn=bn)
    x = torch.randn(B, 64, H, W, generator=g)
    ho, wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    r = torch.randn(B, 128, ho, wo, generator=g) if res else None
    dy = torch.randn(B, 128, ho, wo, generator=g)
    # oracle in float64
    ref = _Holder(conv=torch.nn.Conv2d(64, 128, ks, stride=stride, padding=ks // 2, bias=(stride == 1)),
                  bn=torch.nn.BatchNorm2d(128)).double()
    ref.load_state_dict({k: v.double() for k, v in holder.state_dict().items()})
    ref.train()
    xr = x.double().requires_grad_(True)
    rr = r.double().requires_grad_(True) if res else None
    pre = ref.bn(ref.conv(xr))
    pre = pre + rr if res else pre
    near_zero = pre.detach().abs() < 1e-4
    dy = dy * (~near_zero).float()          # no gradient into ReLU decisions that rounding can flip
    yr = torch.relu(pre)
    yr.backward(dy.double())

    holder.cuda().train()
    tape = T.Tape()
    xs, rs = _nhwc(x), (_nhwc(r) if res else None)
    y = T.conv_bn_act(tape, T._Names(holder), holder.conv, holder.bn, [(xs, 64, 0, 0)], B, H, W, residual=rs)
    tape.add_grad(y, _nhwc(dy))
    tape.backward()
    torch.cuda.synchronize()
    assert _relerr(_nchw(y), yr) < 1e-5
    assert _relerr(_nchw(tape.pop_grad(xs)), xr.grad) < 2e-5
    if res:
        assert _relerr(_nchw(tape.pop_grad(rs)), rr.grad) < 2e-5
    for k, p in ref.named_parameters():
        if k == "conv.bias":   # exactly zero in exact arithmetic (BatchNorm removes the mean)
            assert tape.param_grads[k].abs().max().item() < 1e-4
        else:
            assert _relerr(tape.param_grads[k], p.grad) < 2e-5, k
    assert _relerr(holder.bn.running_var, ref.bn.running_var) < 1e-6



@pytest.mark.parametrize("prec", ["bf16x6", "f16x3"])
def test_concat_pool_convT_backward(T, prec, monkeypatch):
    """One UNet level: skip -> max-pool -> conv/BN/ReLU -> transposed conv -> pad -> cat([skip, up]) ->
    conv/BN/ReLU, odd sizes (pad bottom 1), gradients of the skip tensor from both of its consumers."""
    monkeypatch.setenv("SFH_TRAIN_PRECISION", prec)
    B, H, W = 2, 13, 22
    g = torch.Generator().manual_seed(23)

    def build():
        return _Holder(low=torch.nn.Conv2d(64, 128, 3, padding=1), bnl=torch.nn.BatchNorm2d(128),
                       up=torch.nn.ConvTranspose2d(128, 64, 2, stride=2),
                       conv=torch.nn.Conv2d(128, 64, 3, padding=1), bn=torch.nn.BatchNorm2d(64))

    holder = build()
    skip = torch.randn(B, 64, H, W, generator=g)
    dy = torch.randn(B, 64, H, W, generator=g)
    ref = build().double()
    ref.load_state_dict({k: v.double() for k, v in holder.state_dict().items()})
    ref.train()
    sr = skip.double().requires_grad_(True)
    lo = torch.relu(ref.bnl(ref.low(torch.nn.functional.max_pool2d(sr, 2))))     # (B,128,6,11)
    ur = torch.nn.functional.pad(ref.up(lo), [0, 0, 0, 1])                        # (B,64,12,22) -> 13 rows
    yr = torch.relu(ref.bn(ref.conv(torch.cat([sr, ur], 1))))
    yr.backward(dy.double())

    holder.cuda().train()
    tape = T.Tape()
    names = T._Names(holder)
    s = _nhwc(skip)
    p = T.maxpool2(tape, s)
    lo_g = T.conv_bn_act(tape, names, holder.low, holder.bnl, [(p, 64, 0, 0)], B, H // 2, W // 2)
    u = T.conv_transpose2x2(tape, names, holder.up, lo_g)
    y = T.conv_bn_act(tape, names, holder.conv, holder.bn, [(s, 64, 0, 0), (u, 64, 0, 0)], B, H, W)
    tape.add_grad(y, _nhwc(dy))
    tape.backward()
    torch.cuda.synchronize()
    assert _relerr(_nchw(y), yr) < 1e-5
    for k, q in ref.named_parameters():
        if k not in ("conv.bias", "low.bias"):
            assert _relerr(tape.param_grads[k], q.grad) < 2e-5, k
    assert _relerr(_nchw(tape.pop_grad(s)), sr.grad) < 2e-5


# --------------------------------------------------------------------------- warp / POI backward
def test_warp_and_poi_backward_theta(T):
    from sfh_amd import engine as E
    from oracle import warp_ref
    B, h, w = 3, 45, 80
    court = synth.load_court_template("ncaa_nc4_640x360", 4, B)[:, :, ::4, ::4].contiguous()   # (B,1,90,160)
    poi = synth.load_court_poi("pitch", B)
    g = torch.Generator().manual_seed(9)
    th = torch.tensor(synth.REALISTIC_THETAS[[0, 1, 0]]) + 0.01 * torch.randn(B, 3, 3, generator=g)
    th = (th / th[:, 2:3, 2:3]).reshape(B, 1, 3, 3).contiguous()
    dwarp = torch.randn(B, h, w, generator=g)
    dpoi = torch.randn(B, poi.shape[1], 2, generator=g)
    tr = th.clone().requires_grad_(True)
    warp_ref.homography_warp(tr, court, h, w, "bilinear").backward(dwarp)
    g_warp = tr.grad.clone()
    tr.grad = None
    (warp_ref.transform_points(torch.inverse(tr), poi) / 2.0 + 0.5).backward(dpoi)
    g_poi = tr.grad.clone()
    thc = th.cuda().reshape(B, 9).contiguous()
    got_w = T.warp_backward_theta(thc, court.cuda(), h, w, dwarp.cuda(), shared_template=False)
    got_p = T.poi_backward_theta(thc, poi.cuda(), dpoi.cuda())
    torch.cuda.synchronize()
    assert _relerr(got_w, g_warp.reshape(B, 9)) < 1e-3     # fp32 autograd sums over 3600 pixels on the CPU side
    assert _relerr(got_p, g_poi.reshape(B, 9)) < 1e-3


# --------------------------------------------------------------------------------- whole model
@pytest.mark.parametrize("prec", ["bf16x6", "f16x3"])
def test_full_training_forward_backward(T, prec, monkeypatch):
    """net.train(); preds = net(x); losses (train.py:181-224); loss.backward(): outputs, loss values and
    parameter gradients against torch autograd over the CPU oracle.  With batch-statistics BatchNorm the
    gradient is discontinuous in the forward rounding (a ReLU / max-pool decision that flips in a layer
    with few pixels moves whole rows), so the CPU fp32 oracle itself is several percent away from an
    fp64 run on a few tensors; the per-tensor check is therefore statistical, and the exact per-layer
    checks above carry the tight tolerances."""
    from sfh_amd.reconstructor import Reconstructor
    monkeypatch.setenv("SFH_TRAIN_PRECISION", prec)
    B, H, W = 4, 96, 128
    court = synth.load_court_template("ncaa_nc4_640x360", 4, B)[:, :, :H, :W].contiguous()
    poi = synth.load_court_poi("pitch", B)
    net = Reconstructor(court, poi, target_size=(W, H), unet_size=(W, H), warp_size=(W, H))
    sd = synth.synth_state_dict(net.state_dict(), 43)
    net.load_state_dict(sd)
    x = synth.frames_to_float(synth.synth_frames_u8(B, H, W, seed=43))
    g = torch.Generator().manual_seed(43)
    batch = {"mask": torch.randint(0, 4, (B, H, W), generator=g), "weight": torch.rand(B, generator=g) + 0.5,
             "poi": torch.rand(B, poi.shape[1], 2, generator=g),
             "nonzeros": (torch.rand(B, poi.shape[1], generator=g) > 0.3).float()}
    batch["num_nonzero"] = batch["nonzeros"].sum(1).clamp(min=1.0)

    ref = train_ref.leaf_state(sd)
    preds_ref = train_ref.forward_train(x, ref, court, poi, warp_size=(W, H), unet_size=(W, H), target_size=(W, H))
    loss_ref = train_ref.losses(preds_ref, batch)
    loss_ref["total"].backward()
    want = {k: v.grad for k, v in ref.items() if v.requires_grad}

    net.court_img, net.court_poi = court.cuda(), poi.cuda()
    net.cuda().train()
    preds = net(x.cuda())
    loss = train_ref.losses(preds, {k: v.cuda() for k, v in batch.items()})
    loss["total"].backward()
    torch.cuda.synchronize()

    assert _relerr(preds["logits"], preds_ref["logits"]) < 1e-4
    assert (preds["theta"].cpu() - preds_ref["theta"]).abs().max().item() < 1e-4
    assert (preds["poi"].cpu() - preds_ref["poi"]).abs().max().item() < 1e-3
    assert (preds["warp_mask"].cpu() - preds_ref["warp_mask"]).abs().max().item() < 2e-3
    for k in loss_ref:
        # the consistency target is trunc(warp_mask * 4) (train.py:218): bilinear values that sit on a
        # class boundary flip with the last bit of the warp, hence the wider tolerance there
        tol = 2e-3 if k in ("consist", "total") else 1e-4
        assert abs(loss[k].item() - loss_ref[k].item()) < tol * max(1.0, abs(loss_ref[k].item())), k
    got = {k: p.grad for k, p in net.named_parameters()}
    assert all(v is not None for v in got.values())
    errs = _grad_stats(got, want)
    vals = np.sort(np.array(list(errs.values())))
    assert len(vals) > 150
    assert np.median(vals) < 2e-2, np.median(vals)
    assert vals[int(0.8 * len(vals))] < 8e-2, vals[int(0.8 * len(vals))]
    new = net.state_dict()
    for k, v in ref.items():
        if k.endswith("running_mean") or k.endswith("running_var"):
            assert _relerr(new[k], v) < 1e-4, k


# ----------------------------------------------- vectors produced by the reference's own classes
@pytest.fixture(scope="module")
def golden_train():
    import os
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "train_blocks.npz"))


def _golden_case(T, g, tag, mod, seed, build):
    """build(tape, names, mod, xs) -> y on the HIP training kernels; compare with the reference's
    train-mode outputs / gradients (oracle/make_fixtures.py:make_train_goldens)."""
    sd = synth.synth_state_dict(mod.state_dict(), seed)
    mod.load_state_dict(sd)
    mod.cuda().train()
    xs = [_nhwc(torch.from_numpy(g[f"{tag}.x{i}"])) for i in range(2) if f"{tag}.x{i}" in g]
    tape = T.Tape()
    y = build(tape, T._Names(mod), mod, xs)
    tape.add_grad(y, _nhwc(torch.from_numpy(g[f"{tag}.dy"])))
    tape.backward()
    torch.cuda.synchronize()
    assert _relerr(_nchw(y), torch.from_numpy(g[f"{tag}.y"])) < 1e-5
    for i, t in enumerate(xs):
        assert _relerr(_nchw(tape.pop_grad(t)), torch.from_numpy(g[f"{tag}.dx{i}"])) < 5e-5, (tag, i)
    for k in g.files:
        if k.startswith(tag + ".grad."):
            name = k[len(tag) + 6:]
            want = torch.from_numpy(g[k])
            if name.endswith("double_conv.0.bias") or name.endswith("double_conv.3.bias"):
                # conv bias in front of BatchNorm: zero up to rounding (the reference's value is noise)
                assert want.abs().max() < 1e-3 and tape.param_grads[name].abs().max().item() < 1e-3
            else:
                assert _relerr(tape.param_grads[name], want) < 5e-5, (tag, name)
        if k.startswith(tag + ".buf."):
            assert _relerr(mod.state_dict()[k[len(tag) + 5:]], torch.from_numpy(g[k])) < 1e-6


def _dconv(T, tape, names, block, srcs, B, h, w):
    (cv1, bn1), (cv2, bn2) = block.convs()
    y1 = T.conv_bn_act(tape, names, cv1, bn1, srcs, B, h, w)
    return T.conv_bn_act(tape, names, cv2, bn2, [(y1, y1.shape[3], 0, 0)], B, h, w)


@pytest.mark.parametrize("prec", ["bf16x6", "f16x3"])
def test_training_blocks_vs_reference_golden(T, golden_train, prec, monkeypatch):
    monkeypatch.setenv("SFH_TRAIN_PRECISION", prec)
    g = golden_train

    def dc(tape, names, m, xs):
        B, H, W, C = xs[0].shape
        return _dconv(T, tape, names, m, [(xs[0], C, 0, 0)], B, H, W)

    def down(tape, names, m, xs):
        p = T.maxpool2(tape, xs[0])
        B, H, W, C = p.shape
        return _dconv(T, tape, names, m.block, [(p, C, 0, 0)], B, H, W)

    def up(tape, names, m, xs):
        u = T.conv_transpose2x2(tape, names, m.up, xs[0])
        B, H, W, C = xs[1].shape
        dy, dx = H - u.shape[1], W - u.shape[2]
        return _dconv(T, tape, names, m.conv, [(xs[1], C, 0, 0), (u, u.shape[3], dy // 2, dx // 2)], B, H, W)

    def basic(tape, names, m, xs):
        B, H, W, C = xs[0].shape
        one = lambda t: [(t, t.shape[3], 0, 0)]
        idn = T.conv_bn_act(tape, names, m.downsample[0], m.downsample[1], one(xs[0]), B, H, W, relu=False)
        t = T.conv_bn_act(tape, names, m.conv1, m.bn1, one(xs[0]), B, H, W)
        return T.conv_bn_act(tape, names, m.conv2, m.bn2, one(t), B, t.shape[1], t.shape[2], residual=idn)

    _golden_case(T, g, "t_dc_64_128", modules.DoubleConv(64, 128), 31, dc)
    _golden_case(T, g, "t_down_64_128", modules.Down(64, 128), 32, down)
    _golden_case(T, g, "t_up_128_64", modules.Up(128, 64, False), 33, up)
    ds = torch.nn.Sequential(torch.nn.Conv2d(64, 128, 1, stride=2, bias=False), torch.nn.BatchNorm2d(128))
    _golden_case(T, g, "t_basic_64_128_s2", modules.BasicBlock(64, 128, 2, ds), 34, basic)


# ------------------------------------------------------------ losses, optimizer, whole step on HIP
def _batch(B, H, W, npts, seed):
    g = torch.Generator().manual_seed(seed)
    b = {"mask": torch.randint(0, 4, (B, H, W), generator=g), "weight": torch.rand(B, generator=g) + 0.5,
         "poi": torch.rand(B, npts, 2, generator=g), "nonzeros": (torch.rand(B, npts, generator=g) > 0.3).float()}
    b["num_nonzero"] = b["nonzeros"].sum(1).clamp(min=1.0)
    return b


@pytest.mark.parametrize("rec,cls", [("SmoothL1", "CE"), ("MSE", "focal")])
def test_loss_kernels_vs_torch(T, rec, cls):
    from sfh_amd import _lib
    from sfh_amd.engine import _ptr, _stream
    import torch.nn.functional as F
    lib = _lib.load()
    B, H, W, N = 3, 21, 34, 33
    g = torch.Generator().manual_seed(77)
    logits = (torch.randn(B, 4, H, W, generator=g) * 2).requires_grad_(True)
    warp = (torch.rand(B, H, W, generator=g) * 0.74).requires_grad_(True)
    poi = torch.rand(B, N, 2, generator=g).requires_grad_(True)
    b = _batch(B, H, W, N, 78)
    lam = (2.0, 2.0, 8.0, 1.0)
    ce = (lambda lg, t: F.cross_entropy(lg, t, reduction="none")) if cls == "CE" else train_ref.focal_loss
    seg = train_ref.per_sample_weighted(ce(logits, b["mask"]), b["weight"]) * lam[0]
    gt_f = b["mask"].float() / 4.0
    rl = F.smooth_l1_loss(warp, gt_f, reduction="none") if rec == "SmoothL1" else F.mse_loss(warp, gt_f, reduction="none")
    recl = train_ref.per_sample_weighted(rl, b["weight"]) * lam[1]
    rep = train_ref.reprojection_loss(poi, b["poi"], b["nonzeros"], b["num_nonzero"]) * lam[2]
    cons = ce(logits, (warp * 4).to(torch.long)).mean() * lam[3]
    (seg + recl + rep + cons).backward()

    lc, wc, pc = logits.detach().cuda(), warp.detach().cuda(), poi.detach().cuda()
    bc = {k: v.cuda() for k, v in b.items()}
    losses = torch.zeros(4, dtype=torch.float64, device="cuda")
    dl, dw, dp = torch.empty_like(lc), torch.empty_like(wc), torch.empty_like(pc)
    _lib.check(lib.sfh_train_losses(_ptr(lc), _ptr(bc["mask"]), _ptr(bc["weight"]), _ptr(wc), 4, B, H, W, lam[0], lam[1],
                                    1 if rec == "MSE" else 0, lam[3], 3 if cls == "focal" else 0, _ptr(dl), _ptr(dw),
                                    _ptr(losses), _stream()), "losses")
    import ctypes
    _lib.check(lib.sfh_reproj_loss(_ptr(pc), _ptr(bc["poi"]), _ptr(bc["nonzeros"]), _ptr(bc["num_nonzero"]), B, N, lam[2],
                                   _ptr(dp), ctypes.c_void_p(losses.data_ptr() + 24), _stream()), "reproj")
    torch.cuda.synchronize()
    want = torch.tensor([seg.item(), recl.item(), cons.item(), rep.item()], dtype=torch.float64)
    assert (losses.cpu() - want).abs().max().item() < 1e-5 * want.abs().max().item()
    assert _relerr(dl, logits.grad) < 1e-5
    assert _relerr(dw, warp.grad) < 1e-5
    assert _relerr(dp, poi.grad) < 1e-5


def test_rmsprop_kernel_vs_torch(T):
    """clip_grad_value_(0.1) + RMSprop(lr, weight_decay, momentum 0.9) over three steps, odd tensor sizes."""
    g = torch.Generator().manual_seed(5)
    shapes = [(64, 3, 3, 3), (64,), (9, 512), (70001,), (1,)]
    net = torch.nn.ParameterList([torch.nn.Parameter(torch.randn(s, generator=g)) for s in shapes])
    ref = [p.detach().clone().requires_grad_(True) for p in net]
    opt = torch.optim.RMSprop(ref, lr=1e-3, weight_decay=1e-4, momentum=0.9)
    net.cuda()
    holder = torch.nn.Module()
    holder.ps = net
    ts = T.TrainStep.__new__(T.TrainStep)
    # only the optimizer part of TrainStep, on a bare parameter list
    ts.net, ts.hp = holder, dict(lr=1e-3, wd=1e-4, mu=0.9, alpha=0.99, eps=1e-8, clip=0.1)
    T.TrainStep._init_optimizer(ts, list(net))
    from sfh_amd import _lib
    from sfh_amd.engine import _ptr, _stream
    lib = _lib.load()
    for it in range(3):
        grads = [torch.randn(s, generator=g) * (0.3 if it else 0.05) for s in shapes]
        for p, gr in zip(ref, grads):
            p.grad = gr.clone()
        torch.nn.utils.clip_grad_value_(ref, 0.1)
        opt.step()
        for dst, gr in zip(ts.grads, grads):
            dst.copy_(gr)
        hp = ts.hp
        _lib.check(lib.sfh_rmsprop_step(_ptr(ts.table), _ptr(ts.chunks), ts.nchunks, hp["lr"], hp["alpha"], hp["eps"],
                                        hp["wd"], hp["mu"], hp["clip"], 1.0, _stream()), "rmsprop")
    torch.cuda.synchronize()
    for p, r in zip(net, ref):
        assert (p.detach().cpu() - r.detach()).abs().max().item() < 2e-6


@pytest.mark.parametrize("prec", ["bf16x6", "f16x3"])
def test_train_step_on_hip_matches_autograd_path(T, prec, monkeypatch):
    """TrainStep (losses + backward without torch autograd) against net(x) + torch losses + autograd -
    both on the same HIP forward/backward kernels - and the loss goes down over a few steps."""
    from sfh_amd.reconstructor import Reconstructor
    monkeypatch.setenv("SFH_TRAIN_PRECISION", prec)
    B, H, W = 4, 96, 128
    court = synth.load_court_template("ncaa_nc4_640x360", 4, B)[:, :, :H, :W].contiguous().cuda()
    poi = synth.load_court_poi("pitch", B).cuda()

    def make():
        net = Reconstructor(court, poi, target_size=(W, H), unet_size=(W, H), warp_size=(W, H))
        net.load_state_dict(synth.synth_state_dict(net.state_dict(), 47))
        return net.cuda().train()

    x = synth.frames_to_float(synth.synth_frames_u8(B, H, W, seed=47)).cuda()
    batch = {k: v.cuda() for k, v in _batch(B, H, W, poi.shape[1], 48).items()}
    lam = (2.0, 2.0, 8.0, 1.0)

    a = make()
    preds = a(x)
    la = train_ref.losses(preds, batch, lambdas=lam)
    la["total"].backward()
    ga = {k: p.grad for k, p in a.named_parameters()}

    b = make()
    ts = T.TrainStep(b, lr=1e-4)
    lb = ts.loss_and_grads(x, batch).cpu()
    torch.cuda.synchronize()
    want = torch.tensor([la["seg"].item(), la["rec"].item(), la["consist"].item(), la["reproj"].item()], dtype=torch.float64)
    assert (lb - want).abs().max().item() < 2e-3 * want.abs().max().item()
    gb = {k: g for (k, _), g in zip(b.named_parameters(), ts.grads)}
    errs = np.sort(np.array(list(_grad_stats(gb, {k: v.cpu() for k, v in ga.items()}).values())))
    assert np.median(errs) < 1e-2, (np.median(errs), errs[-5:])

    first = None
    for it in range(6):
        tot = ts.step(x, batch).sum().item()
        assert np.isfinite(tot)
        first = tot if first is None else first
    assert tot < first, (first, tot)


@pytest.mark.parametrize("kw,mode", [
    ({"unet_uv": True, "resnet_input": "img+mask+uv"}, "img+mask+uv"),
    ({"resnet_input": "mask"}, "mask"),
    ({"resnet_input": "img"}, "img"),
    ({"use_resnet": False, "use_warper": False}, None),
    ({"unet_bilinear": True}, "img+mask"),
    ({"unet_size": (80, 48), "unet_uv": True}, "img+mask"),     # bilinear input resize + nearest logits / uv resize
])
def test_training_variants(T, kw, mode):
    """uv head, the other resnet_input modes (models/reconstructor.py:84-97,173-183), UNet-only training,
    the bilinear Up variant and the resize paths around the UNet (:134-156)."""
    from sfh_amd.reconstructor import Reconstructor
    B, H, W = 4, 64, 96
    court = synth.load_court_template("ncaa_nc4_640x360", 4, B)[:, :, :H, :W].contiguous()
    poi = synth.load_court_poi("pitch", B)
    kw = dict(kw)
    usize = kw.pop("unet_size", (W, H))
    net = Reconstructor(court, poi, target_size=(W, H), unet_size=usize, warp_size=(W, H), **kw)
    sd = synth.synth_state_dict(net.state_dict(), 53)
    net.load_state_dict(sd)
    x = synth.frames_to_float(synth.synth_frames_u8(B, H, W, seed=53))
    g = torch.Generator().manual_seed(53)

    ref = train_ref.leaf_state(sd)
    with train_ref.bn_training():
        pr = torch_ref.forward(x, ref, court, poi, warp_size=(W, H), unet_size=usize, target_size=(W, H),
                               resnet_input=mode or "img+mask", use_resnet=mode is not None,
                               bilinear=bool(kw.get("unet_bilinear")))
    douts = {k: torch.randn(v.shape, generator=g) / v.numel() ** 0.5 for k, v in pr.items()}
    sum((pr[k] * douts[k]).sum() for k in pr).backward()
    want = {k: v.grad for k, v in ref.items() if v.requires_grad and v.grad is not None}

    net.court_img, net.court_poi = court.cuda(), poi.cuda()
    net.cuda().train()
    preds = net(x.cuda())
    assert sorted(preds) == sorted(pr)
    sum((preds[k] * douts[k].cuda()).sum() for k in preds).backward()
    torch.cuda.synchronize()
    for k in pr:
        tol = 2e-3 if k == "warp_mask" else 2e-4
        assert (preds[k].detach().cpu() - pr[k].detach()).abs().max().item() < tol * max(1.0, pr[k].abs().max().item()), k
    got = {k: p.grad for k, p in net.named_parameters() if p.grad is not None}
    assert sorted(got) == sorted(want)
    errs = np.sort(np.array(list(_grad_stats(got, want).values())))
    assert np.median(errs) < 2e-2 and errs[int(0.8 * len(errs))] < 1e-1, (np.median(errs), errs[-5:])


def test_upsample_and_nearest_resize_backward(T):
    """backward of nn.Upsample(2x, bilinear, align_corners=True) and of F.interpolate(mode='nearest')"""
    from sfh_amd import _lib
    from sfh_amd.engine import _ptr, _stream
    import torch.nn.functional as F
    lib = _lib.load()
    g = torch.Generator().manual_seed(3)
    for (B, C, H, W) in [(2, 8, 5, 7), (1, 4, 1, 3), (2, 4, 2, 2), (1, 8, 11, 20)]:
        x = torch.randn(B, C, H, W, generator=g, dtype=torch.float64, requires_grad=True)
        dy = torch.randn(B, C, 2 * H, 2 * W, generator=g)
        F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True).backward(dy.double())
        dyc = _nhwc(dy)
        dx = torch.empty((B, H, W, C), device="cuda")
        _lib.check(lib.sfh_upsample2x_bilinear_nhwc_bwd(_ptr(dyc), _ptr(dx), B, H, W, C, _stream()), "up_bwd")
        assert _relerr(_nchw(dx), x.grad) < 1e-5, (B, C, H, W)
    for (hs, ws, hd, wd) in [(48, 80, 64, 96), (64, 96, 48, 80), (7, 9, 23, 31), (5, 5, 5, 5), (30, 17, 11, 40)]:
        x = torch.randn(2, 3, hs, ws, generator=g, dtype=torch.float64, requires_grad=True)
        dy = torch.randn(2, 3, hd, wd, generator=g)
        xf = x.detach().float().requires_grad_(True)
        F.interpolate(xf, size=(hd, wd), mode="nearest").backward(dy)
        dx = torch.empty((2, 3, hs, ws), device="cuda")
        dyc = dy.cuda()
        _lib.check(lib.sfh_resize_nearest_nchw_bwd(_ptr(dyc), _ptr(dx), 6, hs, ws, hd, wd, _stream()), "nearest_bwd")
        assert _relerr(dx, xf.grad) < 1e-6, (hs, ws, hd, wd)


def test_post_step_weights_vs_cpu_restatement(T):
    """One full iteration (train.py:155-237) on HIP (TrainStep) against the CPU restatement (oracle forward +
    losses + autograd + clip_grad_value_ + torch.optim.RMSprop): loss values, and the post-step weights.
    RMSprop's first step moves every weight by about lr*10*sign(g), so the comparison is on the update:
    the updates must agree for all but the few percent of weights whose gradient is near zero or sits in a
    tensor hit by the BatchNorm backward chaos (see test_full_training_forward_backward)."""
    from sfh_amd.reconstructor import Reconstructor
    B, H, W = 4, 96, 128
    lr, lam = 1e-4, (2.0, 2.0, 8.0, 1.0)
    court = synth.load_court_template("ncaa_nc4_640x360", 4, B)[:, :, :H, :W].contiguous()
    poi = synth.load_court_poi("pitch", B)
    net = Reconstructor(court, poi, target_size=(W, H), unet_size=(W, H), warp_size=(W, H))
    sd = synth.synth_state_dict(net.state_dict(), 59)
    net.load_state_dict(sd)
    x = synth.frames_to_float(synth.synth_frames_u8(B, H, W, seed=59))
    batch = _batch(B, H, W, poi.shape[1], 60)

    ref = train_ref.leaf_state(sd)
    params = [v for v in ref.values() if v.requires_grad]
    opt = torch.optim.RMSprop(params, lr=lr, weight_decay=1e-8, momentum=0.9)
    pr = train_ref.forward_train(x, ref, court, poi, warp_size=(W, H), unet_size=(W, H), target_size=(W, H))
    lref = train_ref.losses(pr, batch, lambdas=lam)
    lref["total"].backward()
    torch.nn.utils.clip_grad_value_(params, 0.1)
    opt.step()

    net.court_img, net.court_poi = court.cuda(), poi.cuda()
    net.cuda().train()
    ts = T.TrainStep(net, lr=lr, weight_decay=1e-8)
    lh = ts.step(x.cuda(), {k: v.cuda() for k, v in batch.items()}).cpu()
    torch.cuda.synchronize()
    want = torch.tensor([lref["seg"].item(), lref["rec"].item(), lref["consist"].item(), lref["reproj"].item()], dtype=torch.float64)
    assert (lh - want).abs().max().item() < 2e-3 * want.abs().max().item()
    new = net.state_dict()
    agree = total = 0
    for k, v in ref.items():
        if not v.requires_grad:
            continue
        d_ref = (v.detach() - sd[k]).double()
        d_hip = (new[k].cpu() - sd[k]).double()
        agree += ((d_ref - d_hip).abs() <= 0.05 * (10 * lr)).sum().item()
        total += v.numel()
    assert agree / total > 0.95, agree / total


def test_loss_trajectory_vs_cpu_restatement(T):
    """Three full iterations (train.py:155-237) on the HIP TrainStep and on the CPU restatement from the same
    weights and batch: the loss terms stay together step after step (profiles/r01_train_loss_curve.txt holds
    the 12-step version)."""
    from sfh_amd.reconstructor import Reconstructor
    B, H, W = 4, 96, 128
    lam = (2.0, 2.0, 8.0, 1.0)
    court = synth.load_court_template("ncaa_nc4_640x360", 4, B)[:, :, :H, :W].contiguous()
    poi = synth.load_court_poi("pitch", B)
    net = Reconstructor(court, poi, target_size=(W, H), unet_size=(W, H), warp_size=(W, H))
    sd = synth.synth_state_dict(net.state_dict(), 61)
    net.load_state_dict(sd)
    x = synth.frames_to_float(synth.synth_frames_u8(B, H, W, seed=61))
    batch = _batch(B, H, W, poi.shape[1], 62)
    ref = train_ref.leaf_state(sd)
    params = [v for v in ref.values() if v.requires_grad]
    opt = torch.optim.RMSprop(params, lr=1e-4, weight_decay=1e-8, momentum=0.9)
    cpu = []
    for _ in range(3):
        pr = train_ref.forward_train(x, ref, court, poi, warp_size=(W, H), unet_size=(W, H), target_size=(W, H))
        l = train_ref.losses(pr, batch, lambdas=lam)
        opt.zero_grad()
        l["total"].backward()
        torch.nn.utils.clip_grad_value_(params, 0.1)
        opt.step()
        cpu.append(sum(l[k].item() for k in ("seg", "rec", "consist", "reproj")))
    net.court_img, net.court_poi = court.cuda(), poi.cuda()
    net.cuda().train()
    ts = T.TrainStep(net, lr=1e-4, weight_decay=1e-8)
    xb, bb = x.cuda(), {k: v.cuda() for k, v in batch.items()}
    hip = [ts.step(xb, bb).sum().item() for _ in range(3)]
    assert cpu[2] < cpu[0] and hip[2] < hip[0]
    for a, b in zip(hip, cpu):
        assert abs(a - b) < 1e-2 * b, (hip, cpu)


def test_train_step_checkpoint_resume(T):
    """model state_dict + TrainStep.state_dict() -> a fresh process-equivalent object continues bit-close to
    the uninterrupted run (BatchNorm running stats, RMSprop averages and momentum all travel)."""
    import copy
    from sfh_amd.reconstructor import Reconstructor
    B, H, W = 2, 64, 96
    court = synth.load_court_template("ncaa_nc4_640x360", 4, B)[:, :, :H, :W].contiguous().cuda()
    poi = synth.load_court_poi("pitch", B).cuda()

    def make(sd):
        net = Reconstructor(court, poi, target_size=(W, H), unet_size=(W, H), warp_size=(W, H))
        net.load_state_dict(sd)
        return net.cuda().train()

    x = synth.frames_to_float(synth.synth_frames_u8(B, H, W, seed=71)).cuda()
    batch = {k: v.cuda() for k, v in _batch(B, H, W, poi.shape[1], 72).items()}
    sd0 = synth.synth_state_dict(Reconstructor(court, poi, target_size=(W, H), unet_size=(W, H), warp_size=(W, H)).state_dict(), 71)
    a = make(sd0)
    ta = T.TrainStep(a, lr=1e-4)
    for _ in range(2):
        ta.step(x, batch)
    ck_model = copy.deepcopy({k: v.detach().clone() for k, v in a.state_dict().items()})
    ck_opt = ta.state_dict()
    la = [ta.step(x, batch).sum().item() for _ in range(2)]

    def resume(drop=None):
        b = make(ck_model)
        tb = T.TrainStep(b, lr=3e-5)              # another lr on purpose: the checkpoint's hyper-parameters win
        st = {k: (dict(v) if isinstance(v, dict) else v) for k, v in ck_opt.items()}
        if drop:
            st[drop] = {n: torch.zeros_like(t) for n, t in st[drop].items()}
        tb.load_state_dict(st)
        assert tb.global_step == 2 and tb.hp["lr"] == 1e-4
        lb = [tb.step(x, batch).sum().item() for _ in range(2)]
        dl = max(abs(u - v) / abs(u) for u, v in zip(la, lb))
        # distance between the two runs' UPDATES since the checkpoint, relative to the update itself
        num = den = 0.0
        for (k, pa), pb in zip(a.named_parameters(), b.parameters()):
            ua, ub = pa.detach().double() - ck_model[k].double(), pb.detach().double() - ck_model[k].double()
            num += float((ua - ub).pow(2).sum())
            den += float(ua.pow(2).sum())
        return dl, (num / den) ** 0.5

    dl, dw = resume()
    print("resume: loss deviation %.2e, update deviation %.2e" % (dl, dw))
    # Same arithmetic up to the summation order of the split-K atomics in backward-filter.  RMSprop divides by
    # sqrt(square_avg): elements whose gradient is rounding noise still move by about lr per step, in a direction
    # that noise decides, so the two runs are close in the norm of the update, not element by element.
    assert dl < 2e-4 and dw < 1e-2, (dl, dw)      # measured 6e-5 / 9e-4; without a state tensor 2e-2 / 0.7-0.9
    # the check has teeth: a resume that loses either optimizer state tensor is far outside that bound
    for drop in ("momentum_buffer", "square_avg"):
        dl_bad, dw_bad = resume(drop)
        print("resume without %s: loss deviation %.2e, update deviation %.2e" % (drop, dl_bad, dw_bad))
        assert dw_bad > 0.3 and dw_bad > 6 * dw, (drop, dw_bad, dw)
    bad = dict(ck_opt)
    bad["square_avg"] = {n: t[..., :1] if t.dim() else t for n, t in ck_opt["square_avg"].items()}
    with pytest.raises(RuntimeError):
        T.TrainStep(make(ck_model)).load_state_dict(bad)      # wrong-shaped entry: no silent broadcast


def test_predict_after_train_steps_uses_fresh_weights(T):
    """TrainStep.step writes weights and BatchNorm statistics through raw device pointers (no torch version
    bump): predict() afterwards must re-pack - equal to a fresh model loaded from the trained state_dict.
    Also: replaced parameter storage (net.float() / p.data = ...) is picked up by the optimizer table."""
    from sfh_amd.reconstructor import Reconstructor
    B, H, W = 2, 64, 96
    court = synth.load_court_template("ncaa_nc4_640x360", 4, B)[:, :, :H, :W].contiguous().cuda()
    poi = synth.load_court_poi("pitch", B).cuda()
    kw = dict(target_size=(W, H), unet_size=(W, H), warp_size=(W, H), warp_with_nearest=True)
    net = Reconstructor(court, poi, **kw)
    net.load_state_dict(synth.synth_state_dict(net.state_dict(), 81))
    net.cuda()
    x = synth.frames_to_float(synth.synth_frames_u8(B, H, W, seed=81)).cuda()
    batch = {k: v.cuda() for k, v in _batch(B, H, W, poi.shape[1], 82).items()}
    with torch.no_grad():
        before = net.eval().predict(x, consistency=True)
    ts = T.TrainStep(net.train(), lr=1e-3)
    # freeze BatchNorm bookkeeping that WOULD bump a torch version, so only the raw-pointer writes remain
    for _ in range(3):
        ts.step(x, batch)
    with torch.no_grad():
        after = net.eval().predict(x, consistency=True)
    fresh = Reconstructor(court, poi, **kw)
    fresh.load_state_dict({k: v.detach().clone() for k, v in net.state_dict().items()})
    with torch.no_grad():
        want = fresh.cuda().eval().predict(x, consistency=True)
    assert not torch.equal(before["theta"], after["theta"])
    for k in ("theta", "logits", "warp_mask", "consist_score"):
        assert torch.equal(after[k], want[k]), k
    # parameter storage replaced behind the optimizer's back: the table follows
    net.train()
    p0 = next(net.parameters())
    old_ptr = p0.data_ptr()
    p0.data = p0.data.clone()
    assert p0.data_ptr() != old_ptr
    w_before = p0.detach().clone()
    ts.step(x, batch)
    torch.cuda.synchronize()
    assert not torch.equal(p0.detach(), w_before)             # the NEW storage was updated


def test_train_step_f16x3_range_fallback(T, monkeypatch):
    """SFH_TRAIN_PRECISION=f16x3: a step whose activations leave the fp16 range (first BatchNorm scaled by 2^16, the
    next conv divided by it) is repeated with bf16x6 operands FROM THE SAME BatchNorm statistics: losses, gradients
    and running statistics equal those of a plain bf16x6 step; an ordinary model does not fall back."""
    from sfh_amd.reconstructor import Reconstructor
    B, H, W = 2, 64, 96
    court = synth.load_court_template("ncaa_nc4_640x360", 4, B)[:, :, :H, :W].contiguous().cuda()
    poi = synth.load_court_poi("pitch", B).cuda()
    g = torch.Generator().manual_seed(7)
    x = synth.frames_to_float(synth.synth_frames_u8(B, H, W, seed=7)).cuda()
    batch = {"mask": torch.randint(0, 4, (B, H, W), generator=g).cuda(), "weight": torch.ones(B).cuda(),
             "poi": torch.rand(B, poi.shape[1], 2, generator=g).cuda(), "nonzeros": torch.ones(B, poi.shape[1]).cuda()}
    batch["num_nonzero"] = batch["nonzeros"].sum(1)

    def run(prec, blow_up):
        monkeypatch.setenv("SFH_TRAIN_PRECISION", prec)
        net = Reconstructor(court, poi, target_size=(W, H), unet_size=(W, H), warp_size=(W, H))
        sd = synth.synth_state_dict(net.state_dict(), 7)
        if blow_up:
            sd["inc.double_conv.1.weight"] = sd["inc.double_conv.1.weight"] * 65536.0
            sd["inc.double_conv.1.bias"] = sd["inc.double_conv.1.bias"] * 65536.0
        net.load_state_dict(sd)
        net.cuda().train()
        ts = T.TrainStep(net, lr=1e-5, weight_decay=1e-8, seg_lambda=1.0, rec_lambda=1.0, reproj_lambda=1.0, consist_lambda=1.0)
        losses = ts.loss_and_grads(x, batch)
        torch.cuda.synchronize()
        return losses.clone(), ts.gflat.clone(), {k: v.clone() for k, v in net.state_dict().items() if "running" in k or "tracked" in k}, ts

    l0, g0, b0, ts0 = run("f16x3", False)
    assert ts0.range_fallbacks == 0
    with pytest.warns(UserWarning, match="fp16 range"):
        l1, g1, b1, ts1 = run("f16x3", True)
    assert ts1.range_fallbacks == 1
    l2, g2, b2, _ = run("bf16x6", True)
    # (split-K / reduction atomics make two runs of the same step differ in the last bits)
    assert torch.allclose(l1, l2, rtol=1e-6, atol=0) and _relerr(g1, g2) < 1e-4
    for k in b1:      # a step repeated WITHOUT the restored statistics would have moved them twice (momentum 0.1)
        assert torch.allclose(b1[k].double(), b2[k].double(), rtol=1e-5, atol=1e-7), k


def test_autograd_node_repeats_a_step_whose_gradient_leaves_the_fp16_range(T, monkeypatch):
    """net.train(); net(x); loss.backward() (the reference's train.py structure) with SFH_TRAIN_PRECISION=f16x3: a
    backward pass that does not fit the two-plane fp16 format - here a loss scaled until the gradients overflow it,
    then a NaN loss - is repeated with bf16x6 operands from the BatchNorm statistics of BEFORE the step: the
    gradients equal those of a plain bf16x6 step, the running statistics advance once, a NaN loss gives non-finite
    gradients like the reference; no exception reaches the training loop."""
    from sfh_amd.reconstructor import Reconstructor
    B, H, W = 2, 64, 96
    court = synth.load_court_template("ncaa_nc4_640x360", 4, B)[:, :, :H, :W].contiguous().cuda()
    poi = synth.load_court_poi("pitch", B).cuda()
    x = synth.frames_to_float(synth.synth_frames_u8(B, H, W, seed=9)).cuda()
    mask = torch.randint(0, 4, (B, H, W), generator=torch.Generator().manual_seed(9)).cuda()

    def run(prec, logit_gain, theta_gain, poison=False):
        monkeypatch.setenv("SFH_TRAIN_PRECISION", prec)
        net = Reconstructor(court, poi, target_size=(W, H), unet_size=(W, H), warp_size=(W, H))
        net.load_state_dict(synth.synth_state_dict(net.state_dict(), 9))
        net.cuda().train()
        preds = net(x)
        loss = torch.nn.functional.cross_entropy(preds["logits"], mask) * logit_gain + preds["theta"].square().sum() * theta_gain
        if poison:
            loss = loss * float("nan")
        loss.backward()
        torch.cuda.synchronize()
        g = torch.cat([p.grad.reshape(-1) for p in net.parameters()])
        stats = {k: v.clone() for k, v in net.state_dict().items() if "running" in k or "tracked" in k}
        return g, stats, net.__dict__.get("train_range_fallbacks", 0)

    # theta gradients 2^40 above what one scale with the head gradients can carry: the step does not fit the format
    g_ref, s_ref, n_ref = run("bf16x6", 1.0, 2.0 ** 40)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        g_h2, s_h2, n_h2 = run("f16x3", 1.0, 2.0 ** 40)
    assert n_ref == 0 and n_h2 == 1
    assert _relerr(g_h2, g_ref) < 1e-4
    for k in s_ref:
        assert torch.allclose(s_h2[k].double(), s_ref[k].double(), rtol=1e-5, atol=1e-7), k
    # an ordinary step stays on the two-plane path
    _, _, n_ok = run("f16x3", 1.0, 1.0)
    assert n_ok == 0
    # a NaN loss: non-finite gradients, not an exception; statistics advanced once
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        g_nan, s_nan, n_nan = run("f16x3", 1.0, 1.0, poison=True)
    assert n_nan == 1 and not torch.isfinite(g_nan).all()
    for k in s_ref:
        if k.endswith("num_batches_tracked"):
            assert int(s_nan[k]) == int(s_ref[k])


def test_bn_snapshot_follows_replaced_buffers(T):
    """_BNSnapshot looks the buffers up at every save(): after net.to() / .float() the model holds NEW buffer objects,
    and a restore must write into those."""
    from sfh_amd.reconstructor import Reconstructor
    court = synth.load_court_template("ncaa_nc4_640x360", 4, 1)[:, :, :32, :32].contiguous().cuda()
    net = Reconstructor(court, synth.load_court_poi("pitch", 1).cuda(), target_size=(32, 32), unet_size=(32, 32), warp_size=(32, 32))
    net.cuda().train()
    snap = T._BNSnapshot(net)
    g1 = snap.save()
    bn = net.inc.double_conv[1]
    bn.running_mean.add_(1.0)
    snap.restore()
    assert float(bn.running_mean.abs().max()) == 0.0
    net.double().float()                       # replaces every buffer object
    assert all(a is not b for a, b in zip(snap.bufs, net.buffers()) if a.is_floating_point())
    g2 = snap.save()
    assert g2 == g1 + 1 and all(a is b for a, b in zip(snap.bufs, [b for b in net.buffers() if b.is_cuda]))
    net.inc.double_conv[1].running_mean.add_(2.0)
    snap.restore()
    assert float(net.inc.double_conv[1].running_mean.abs().max()) == 0.0


@pytest.mark.parametrize("shape,c0,c1,cout", [((2, 13, 37), 64, 0, 64), ((1, 22, 40), 128, 64, 128), ((3, 9, 20), 256, 0, 256),
                                              ((16, 45, 80), 128, 0, 128)])
def test_conv_epilogue_leaves_the_batchnorm_sums(T, shape, c0, c1, cout, monkeypatch):
    """sfh_conv_desc.stats_partial (training-only instances of the split-operand conv kernel): the per-channel sums of z
    and z^2 the epilogue leaves, added up by sfh_bn_stats_partials, equal what the separate pass (sfh_bn_stats) finds in the
    z the same launch wrote - partial tiles, frames that end inside a tile, two sources, 64- and 128-cout workgroups,
    a grid large enough for the single-buffered instance."""
    from sfh_amd import _lib, engine as E
    from sfh_amd.engine import PackedConv, _ptr, _stream
    monkeypatch.setenv("SFH_TRAIN_PRECISION", "f16x3")
    lib = _lib.load()
    B, H, W = shape
    g = torch.Generator().manual_seed(11 + H)
    w = (torch.randn(cout, c0 + c1, 3, 3, generator=g) * 0.05).cuda()
    b = torch.randn(cout, generator=g).cuda()
    x0 = torch.randn(B, H, W, c0, generator=g).cuda() + 0.5
    x1 = torch.randn(B, H - 3, W - 2, c1, generator=g).cuda() if c1 else None
    pc = PackedConv(w, b, None, 3, c0, c1, relu=False, tag="train_fwd", fmt="h2", shared_unit_scale=True)
    assert pc.stats_ok
    z = torch.empty(B, H, W, cout, device="cuda")
    stats = torch.zeros(256, 2, cout, dtype=torch.float64, device="cuda")
    pc.run(E.f32_to_split(x0, "h2"), B, H, W, z, src1=E.f32_to_split(x1, "h2") if c1 else None, pad1=(1, 1), stats=stats)
    acc = torch.zeros(2 * cout, dtype=torch.float64, device="cuda")
    _lib.check(lib.sfh_bn_stats_partials(_ptr(stats), 256, cout, _ptr(acc), _stream()), "bn_stats_partials")
    ref = torch.zeros(2 * cout, dtype=torch.float64, device="cuda")
    _lib.check(lib.sfh_bn_stats(_ptr(z), B * H * W, cout, _ptr(ref), _stream()), "bn_stats")
    torch.cuda.synchronize()
    zz = z.double().reshape(-1, cout)
    exact = torch.cat([zz.sum(0), (zz * zz).sum(0)])
    assert float(((acc - exact).abs() / (exact.abs() + 1e-9)).max()) < 1e-12
    assert float(((ref - exact).abs() / (exact.abs() + 1e-9)).max()) < 1e-12
    # a launch without the table takes the ordinary instance and writes the same z
    z2 = torch.empty_like(z)
    pc.run(E.f32_to_split(x0, "h2"), B, H, W, z2, src1=E.f32_to_split(x1, "h2") if c1 else None, pad1=(1, 1))
    assert torch.equal(z, z2)


@pytest.mark.parametrize("prec", ["bf16x6", "f16x3"])
def test_batchnorm_split_copy_without_the_fp32_copy(T, prec, monkeypatch):
    """bn_apply / bn_bwd_apply with y == NULL / dz == NULL (layers whose consumers read the split copy only): the split
    copy holds the same bits as with the fp32 copy, odd row length and a pixel count that is no multiple of 64."""
    from sfh_amd import _lib, engine as E
    from sfh_amd.engine import _ptr, _stream
    monkeypatch.setenv("SFH_TRAIN_PRECISION", prec)
    lib = _lib.load()
    tape = T.Tape()
    B, H, W, C = 3, 7, 37, 96
    g = torch.Generator().manual_seed(4)
    z = torch.randn(B, H, W, C, generator=g).cuda()
    dy = torch.randn(B, H, W, C, generator=g).cuda()
    mi = torch.cat([torch.randn(C, generator=g) * 0.1, torch.rand(C, generator=g) + 0.5]).cuda()
    gam, bet = (torch.rand(C, generator=g) + 0.5).cuda(), (torch.randn(C, generator=g) * 0.1).cuda()
    acc = torch.randn(2 * C, generator=g).double().cuda()
    npix = B * H * W
    outs = []
    for with_f32 in (True, False):
        y = torch.empty_like(z) if with_f32 else None
        ys = E.split_empty(tape.fmt, B, H, W, C, "cuda")
        _lib.check(lib.sfh_bn_apply(_ptr(z), _ptr(mi), _ptr(gam), _ptr(bet), None, 1, npix, C, _ptr(y), _ptr(ys), W,
                                    tape.fmt_code, None, _stream()), "bn_apply")
        dz = torch.empty_like(z) if with_f32 else None
        dzs = E.split_empty(tape.fmt, B, H, W, C, "cuda")
        a32 = torch.empty(2 * C, device="cuda")
        _lib.check(lib.sfh_bn_bwd_apply(_ptr(dy), None, _ptr(z), _ptr(mi), _ptr(gam), _ptr(bet), _ptr(acc), 1, npix, C,
                                        _ptr(dz), None, _ptr(dzs), W, tape.fmt_code, None, _ptr(a32), _stream()), "bn_bwd_apply")
        outs.append((y, ys, dz, dzs, a32))
    torch.cuda.synchronize()
    (y, ys, dz, dzs, a32), (_, ys2, _, dzs2, a32b) = outs
    assert torch.equal(ys.view(torch.int16), ys2.view(torch.int16)) and torch.equal(dzs.view(torch.int16), dzs2.view(torch.int16))
    assert torch.equal(a32, a32b) and torch.equal(a32, acc.float())
    want = torch.relu((z - mi[:C]) * mi[C:] * gam + bet)
    assert torch.equal(y, want) or float((y - want).abs().max()) < 1e-6
    assert float((E.s3_to_f32(ys) - y).abs().max()) < 2e-6 * float(y.abs().max())


@pytest.mark.parametrize("shape,cin,cout", [((2, 13, 37), 64, 64), ((1, 22, 40), 128, 256), ((16, 45, 80), 128, 128)])
def test_backward_data_epilogue_leaves_the_batchnorm_backward_sums(T, shape, cin, cout, monkeypatch):
    """sfh_conv_desc.bwd_z: the backward-data launch of a conv whose input is a BatchNorm + ReLU output with no other
    consumer leaves sum g and sum g * xhat of that layer (g = dy * (y > 0), y recomputed from z as bn_apply does) -
    equal to what sfh_bn_bwd_reduce finds in the dy the same launch wrote."""
    from sfh_amd import _lib, engine as E
    from sfh_amd.engine import PackedConv, _ptr, _stream
    monkeypatch.setenv("SFH_TRAIN_PRECISION", "f16x3")
    lib = _lib.load()
    B, H, W = shape
    g = torch.Generator().manual_seed(23 + H)
    w = (torch.randn(cout, cin, 3, 3, generator=g) * 0.05).cuda()          # the consumer conv: cin -> cout
    dz2 = torch.randn(B, H, W, cout, generator=g).cuda()                    # its dz
    z1 = torch.randn(B, H, W, cin, generator=g).cuda()                      # pre-BatchNorm tensor of the producer layer
    mi = torch.cat([torch.randn(cin, generator=g) * 0.2, torch.rand(cin, generator=g) + 0.5]).cuda()
    gam, bet = (torch.rand(cin, generator=g) + 0.5).cuda(), (torch.randn(cin, generator=g) * 0.3).cuda()
    bd = PackedConv.backward_data(w, 3, fmt="h2")
    assert bd.stats_ok and bd.cout == cin
    dy = torch.empty(B, H, W, cin, device="cuda")
    table = torch.zeros(128, 2, cin, dtype=torch.float64, device="cuda")
    bd.run(E.f32_to_split(dz2, "h2"), B, H, W, dy, stats=table, bwd=(z1, mi, gam, bet))
    acc = torch.zeros(2 * cin, dtype=torch.float64, device="cuda")
    _lib.check(lib.sfh_bn_stats_partials(_ptr(table), 128, cin, _ptr(acc), _stream()), "bn_stats_partials")
    ref = torch.zeros(2 * cin, dtype=torch.float64, device="cuda")
    _lib.check(lib.sfh_bn_bwd_reduce(_ptr(dy), None, _ptr(z1), _ptr(mi), _ptr(gam), _ptr(bet), 1, B * H * W, cin, _ptr(ref),
                                     _stream()), "bn_bwd_reduce")
    torch.cuda.synchronize()
    scale = torch.cat([dy.abs().double().reshape(-1, cin).sum(0)] * 2) + 1e-9
    assert float(((acc - ref).abs() / scale).max()) < 1e-12
    dy2 = torch.empty_like(dy)
    bd.run(E.f32_to_split(dz2, "h2"), B, H, W, dy2)
    assert torch.equal(dy, dy2)


@pytest.mark.parametrize("fmt", ["h2", "s3"])
@pytest.mark.parametrize("shape,cout", [((2, 11, 20), 64), ((1, 5, 70), 32), ((3, 7, 9), 8), ((16, 22, 40), 256)])
def test_conv_transpose_backward_first_pass_in_one_kernel(T, shape, cout, fmt):
    """sfh_s2d_split_colsum: bias gradient, space-to-depth and the split copy of ConvTranspose2d's output gradient in one
    pass - the split planes bit for bit those of sfh_space_to_depth2 + sfh_f32_to_h2 / _s3, the column sums those of
    sfh_colsum (fp64, different order of additions)."""
    from sfh_amd import _lib, engine as E
    from sfh_amd.engine import _ptr, _stream
    lib = _lib.load()
    B, h, w = shape
    g = torch.Generator().manual_seed(5 + h + cout)
    du = (torch.randn(B, 2 * h, 2 * w, cout, generator=g) * torch.rand(1, 1, 1, cout, generator=g) * 3).cuda()
    s = torch.empty(B, h, w, 4 * cout, device="cuda")
    _lib.check(lib.sfh_space_to_depth2(_ptr(du), _ptr(s), B, 2 * h, 2 * w, cout, _stream()), "space_to_depth2")
    want = E.f32_to_split(s, fmt)
    ref = T._colsum(lib, du).double()
    got = E.split_empty(fmt, B, h, w, 4 * cout, "cuda")
    acc = torch.zeros(cout, dtype=torch.float64, device="cuda")
    over = torch.zeros(1, dtype=torch.int32, device="cuda")
    _lib.check(lib.sfh_s2d_split_colsum(_ptr(du), B, h, w, cout, _ptr(got), {"s3": _lib.FMT_S3, "h2": _lib.FMT_H2}[fmt], _ptr(acc), 1, _ptr(over), _stream()),
               "s2d_split_colsum")
    torch.cuda.synchronize()
    assert torch.equal(got.view(torch.int16), want.view(torch.int16))
    assert int(over) == 0
    assert float((acc - ref).abs().max()) <= 1e-6 * float(du.abs().double().sum(dim=(0, 1, 2)).max())
    assert float((acc - du.double().sum(dim=(0, 1, 2))).abs().max()) <= 1e-9 * float(du.abs().double().sum(dim=(0, 1, 2)).max())
    # the sums spread over the rows of a table (what the training step passes: same-address fp64 atomics are slow)
    table = torch.zeros(32, cout, dtype=torch.float64, device="cuda")
    _lib.check(lib.sfh_s2d_split_colsum(_ptr(du), B, h, w, cout, _ptr(got), {"s3": _lib.FMT_S3, "h2": _lib.FMT_H2}[fmt], _ptr(table), 32,
                                        _ptr(over), _stream()), "s2d_split_colsum")
    torch.cuda.synchronize()
    assert float((table.sum(0) - acc).abs().max()) <= 1e-9 * float(du.abs().double().sum(dim=(0, 1, 2)).max())
    # a value outside the H2 range raises the overflow word, as the separate conversion does
    if fmt == "h2":
        du[0, 1, 1, 3] = 1e6
        _lib.check(lib.sfh_s2d_split_colsum(_ptr(du), B, h, w, cout, _ptr(got), _lib.FMT_H2, _ptr(acc), 1, _ptr(over), _stream()),
                   "s2d_split_colsum")
        torch.cuda.synchronize()
        assert int(over) == 1


@pytest.mark.parametrize("fmt", ["h2", "s3"])
@pytest.mark.parametrize("shape,C", [((2, 12, 20), 64), ((1, 45, 81), 32), ((3, 7, 9), 128), ((16, 45, 80), 512), ((1, 2, 2), 64)])
def test_skip_tensor_batchnorm_relu_maxpool_in_one_pass(T, shape, C, fmt):
    """sfh_bn_apply_pool / sfh_pool2_bwd_bn_reduce (odd sizes: the cropped row / column still gets its y and its gradient):
    forward planes bit for bit those of sfh_bn_apply -> sfh_maxpool2_fwd -> sfh_f32_to_h2 / _s3; backward dx bit for bit
    that of sfh_maxpool2_bwd (fresh and accumulating), sums those of sfh_bn_bwd_reduce over the finished dx."""
    from sfh_amd import _lib, engine as E
    from sfh_amd.engine import _ptr, _stream
    lib = _lib.load()
    B, H, W = shape
    code = {"s3": _lib.FMT_S3, "h2": _lib.FMT_H2}[fmt]
    g = torch.Generator().manual_seed(31 + H + C)
    z = torch.randn(B, H, W, C, generator=g).cuda()
    z = torch.where(torch.rand(B, H, W, 1, generator=g).cuda() < 0.5, torch.round(z * 2) / 2, z)   # exact ties in half the windows
    mi = torch.cat([torch.randn(C, generator=g) * 0.2, torch.rand(C, generator=g) + 0.5]).cuda()
    gam, bet = (torch.rand(C, generator=g) + 0.5).cuda(), (torch.randn(C, generator=g) * 0.3).cuda()
    npix = B * H * W
    # reference: three passes
    y = torch.empty_like(z)
    ys_ref = E.split_empty(fmt, B, H, W, C, "cuda")
    _lib.check(lib.sfh_bn_apply(_ptr(z), _ptr(mi), _ptr(gam), _ptr(bet), None, 1, npix, C, _ptr(y), _ptr(ys_ref), W, code, None,
                                _stream()), "bn_apply")
    p = torch.empty(B, H // 2, W // 2, C, device="cuda")
    _lib.check(lib.sfh_maxpool2_fwd(_ptr(y), _ptr(p), B, H, W, C, _stream()), "maxpool2_fwd")
    ps_ref = E.f32_to_split(p, fmt)
    ys, ps = E.split_empty(fmt, B, H, W, C, "cuda"), E.split_empty(fmt, B, H // 2, W // 2, C, "cuda")
    over = torch.zeros(1, dtype=torch.int32, device="cuda")
    _lib.check(lib.sfh_bn_apply_pool(_ptr(z), _ptr(mi), _ptr(gam), _ptr(bet), B, H, W, C, _ptr(ys), _ptr(ps), code, _ptr(over),
                                     _stream()), "bn_apply_pool")
    torch.cuda.synchronize()
    assert torch.equal(ys.view(torch.int16), ys_ref.view(torch.int16))
    assert torch.equal(ps.view(torch.int16), ps_ref.view(torch.int16))
    assert int(over) == 0
    if fmt == "s3":
        return
    dp = torch.randn(B, H // 2, W // 2, C, generator=g).cuda()
    other = torch.randn(B, H, W, C, generator=g).cuda()       # the gradient from y's other consumer
    for accumulate in (1, 0):
        want = other.clone() if accumulate else (torch.empty_like(other) if H % 2 == 0 and W % 2 == 0 else torch.zeros_like(other))
        _lib.check(lib.sfh_maxpool2_bwd(_ptr(y), _ptr(dp), _ptr(want), B, H, W, C,
                                        1 if (accumulate or H % 2 or W % 2) else 0, _stream()), "maxpool2_bwd")
        ref = torch.zeros(2 * C, dtype=torch.float64, device="cuda")
        _lib.check(lib.sfh_bn_bwd_reduce(_ptr(want), None, _ptr(z), _ptr(mi), _ptr(gam), _ptr(bet), 1, npix, C, _ptr(ref),
                                         _stream()), "bn_bwd_reduce")
        got = other.clone() if accumulate else torch.full_like(other, float("nan"))
        acc = torch.zeros(2 * C, dtype=torch.float64, device="cuda")
        _lib.check(lib.sfh_pool2_bwd_bn_reduce(_ptr(z), _ptr(mi), _ptr(gam), _ptr(bet), _ptr(dp), B, H, W, C, accumulate, _ptr(got),
                                               _ptr(acc), _stream()), "pool2_bwd_bn_reduce")
        torch.cuda.synchronize()
        assert torch.equal(got, want)
        scale = torch.cat([want.abs().double().reshape(-1, C).sum(0)] * 2) * 8 + 1e-9
        assert float(((acc - ref).abs() / scale).max()) < 1e-12


@pytest.mark.parametrize("prec", ["bf16x6", "f16x3"])
def test_fused_training_passes_give_the_gradients_of_the_separate_ones(T, prec, monkeypatch):
    """Round 5: the one-pass forms (BatchNorm + ReLU + max-pool of the skip tensors; max-pool backward + BatchNorm sums;
    ConvTranspose2d backward's bias gradient + space-to-depth + split copy; the first layer's BatchNorm backward inside its
    backward-filter kernel) are what a training step runs, and they
    give the gradients of the separate passes (same values per element; fp64 sums in another order), also at odd sizes."""
    from sfh_amd.reconstructor import Reconstructor
    monkeypatch.setenv("SFH_TRAIN_PRECISION", prec)
    B, H, W = 2, 90, 136                      # 45 x 68 -> 22 x 34 -> 11 x 17: cropped rows / columns at two levels
    court = synth.load_court_template("ncaa_nc4_640x360", 4, B)[:, :, :H, :W].contiguous().cuda()
    poi = synth.load_court_poi("pitch", B).cuda()
    x = synth.frames_to_float(synth.synth_frames_u8(B, H, W, seed=7)).cuda()
    batch = {k: v.cuda() for k, v in _batch(B, H, W, poi.shape[1], 8).items()}
    tags = []
    real = T._lib.check
    monkeypatch.setattr(T._lib, "check", lambda rc, tag="": (tags.append(tag), real(rc, tag))[1])

    def grads(fused):
        monkeypatch.setattr(T, "POOL_FUSED", fused)
        monkeypatch.setattr(T, "S2D_FUSED", fused)
        monkeypatch.setattr(T, "C4_BN_FUSED", fused)
        monkeypatch.setattr(T, "UP_SUMS_FUSED", fused)
        monkeypatch.setattr(T, "OUTCONV_SUMS_FUSED", fused)
        net = Reconstructor(court, poi, target_size=(W, H), unet_size=(W, H), warp_size=(W, H))
        net.load_state_dict(synth.synth_state_dict(net.state_dict(), 3))
        net.cuda().train()
        ts = T.TrainStep(net, lr=1e-4)
        del tags[:]
        losses = ts.loss_and_grads(x, batch).cpu()
        torch.cuda.synchronize()
        return losses, {k: g.clone() for (k, _), g in zip(net.named_parameters(), ts.grads)}, list(tags)

    l1, g1, t1 = grads(True)
    l0, g0, t0 = grads(False)
    assert t1.count("bn_apply_pool") == 4 and t1.count("pool2_bwd_bn_reduce") == 4 and t1.count("s2d_split_colsum") == 4
    assert "maxpool2_fwd" not in t1 and "maxpool2_bwd" not in t1 and "colsum" not in t1
    assert t0.count("maxpool2_fwd") == 4 and "bn_apply_pool" not in t0 and "s2d_split_colsum" not in t0
    # four skip tensors; in f16x3 also the four inputs of transposed convs (the sums ride in a conv epilogue of the H2 kernel)
    # and the last DoubleConv's from the OutConv backward
    assert t1.count("bn_bwd_reduce") == t0.count("bn_bwd_reduce") - (9 if prec == "f16x3" else 5)
    assert t1.count("outconv_bwd_bn") == 1 and "outconv_bwd" not in t1 and t0.count("outconv_bwd") == 1
    assert t1.count("conv_wgrad_c4_bn") == 1 and "conv_wgrad_c4_bn" not in t0 and t1.count("bn_bwd_apply") == t0.count("bn_bwd_apply") - 1
    # (the forward values are the same per element; the batch statistics come from fp64 atomics whose order varies run to run)
    assert float((l1 - l0).abs().max()) < 1e-5 * float(l0.abs().max())
    for k in g0:
        d = float((g1[k].double() - g0[k].double()).norm()) / (float(g0[k].double().norm()) + 1e-30)
        assert d < 1e-5 or float(g0[k].abs().max()) < 1e-12, (k, d)


# ------------------------------------------------------------------ round 6: the reference's other optimizers and the UV loss
@pytest.mark.parametrize("name", ["SGD", "Adam"])
def test_sgd_and_adam_kernels_vs_torch(T, name):
    """train.py:89-92: clip_grad_value_(0.1) + torch.optim.SGD(lr, momentum 0.9, weight_decay) / torch.optim.Adam(lr,
    betas (0.9, 0.999), weight_decay) over three steps on odd tensor sizes, like test_rmsprop_kernel_vs_torch."""
    g = torch.Generator().manual_seed(6)
    shapes = [(64, 3, 3, 3), (64,), (9, 512), (70001,), (1,)]
    net = torch.nn.ParameterList([torch.nn.Parameter(torch.randn(s, generator=g)) for s in shapes])
    ref = [p.detach().clone().requires_grad_(True) for p in net]
    if name == "SGD":
        opt = torch.optim.SGD(ref, lr=1e-2, weight_decay=1e-4, momentum=0.9)
    else:
        opt = torch.optim.Adam(ref, lr=1e-3, betas=(0.9, 0.999), weight_decay=1e-4)
    net.cuda()
    holder = torch.nn.Module()
    holder.ps = net
    ts = T.TrainStep.__new__(T.TrainStep)
    ts.net, ts.hp = holder, dict(lr=1e-2 if name == "SGD" else 1e-3, wd=1e-4, mu=0.9, alpha=0.99, eps=1e-8, clip=0.1)
    T.TrainStep._init_optimizer(ts, list(net))
    from sfh_amd import _lib
    from sfh_amd.engine import _ptr, _stream
    lib = _lib.load()
    for it in range(3):
        grads = [torch.randn(s, generator=g) * (0.3 if it else 0.05) for s in shapes]
        for p, gr in zip(ref, grads):
            p.grad = gr.clone()
        torch.nn.utils.clip_grad_value_(ref, 0.1)
        opt.step()
        for dst, gr in zip(ts.grads, grads):
            dst.copy_(gr)
        hp = ts.hp
        if name == "SGD":
            _lib.check(lib.sfh_sgd_step(_ptr(ts.table), _ptr(ts.chunks), ts.nchunks, hp["lr"], hp["wd"], hp["mu"], hp["clip"],
                                        1.0, _stream()), "sgd")
        else:
            _lib.check(lib.sfh_adam_step(_ptr(ts.table), _ptr(ts.chunks), ts.nchunks, hp["lr"], 0.9, 0.999, hp["eps"],
                                         hp["wd"], hp["clip"], 1.0, it + 1, _stream()), "adam")
    torch.cuda.synchronize()
    for p, r in zip(net, ref):
        assert (p.detach().cpu() - r.detach()).abs().max().item() < 2e-6
    # Adam's step counter starts at 1
    assert lib.sfh_adam_step(_ptr(ts.table), _ptr(ts.chunks), ts.nchunks, 1e-3, 0.9, 0.999, 1e-8, 0.0, 0.1, 1.0, 0, _stream()) != 0


@pytest.mark.parametrize("crit", ["MSE", "SmoothL1"])
@pytest.mark.parametrize("shape", [(1, 2, 21, 34), (12, 2, 9, 12)])
def test_uv_loss_kernel_vs_the_reference_rule(T, crit, shape):
    """train.py:136-144,203-208 with models/losses.py:33-41 AS WRITTEN: on the (B,2,H,W) uv maps torch.mean(loss, dim=(1, 2))
    leaves (B, W) and the (B,) weights broadcast along the LAST axis - legal for B == 1 and for B == W only."""
    import torch.nn.functional as F
    from sfh_amd import _lib
    from sfh_amd.engine import _ptr, _stream
    lib = _lib.load()
    B, C, H, W = shape
    g = torch.Generator().manual_seed(21)
    uv = (torch.rand(shape, generator=g) * 3 - 1).requires_grad_(True)      # differences beyond 1: both SmoothL1 branches
    gt = torch.rand(shape, generator=g)
    wgt = torch.rand(B, generator=g) + 0.5
    lam = 2.0
    fn = F.mse_loss if crit == "MSE" else F.smooth_l1_loss
    want = train_ref.per_sample_weighted(fn(uv, gt, reduction="none"), wgt) * lam
    want.backward()
    duv = torch.empty(shape, device="cuda")
    loss = torch.zeros(1, dtype=torch.float64, device="cuda")
    uvc, gtc, wc = uv.detach().cuda(), gt.cuda(), wgt.cuda()
    _lib.check(lib.sfh_uv_loss(_ptr(uvc), _ptr(gtc), _ptr(wc), B, B, C, H, W, lam, 1 if crit == "MSE" else 0, _ptr(duv),
                               _ptr(loss), _stream()), "uv_loss")
    torch.cuda.synchronize()
    assert abs(float(loss) - float(want)) < 1e-6 * max(1.0, abs(float(want)))
    assert _relerr(duv, uv.grad) < 1e-6
    # any other batch size is the reference's broadcasting error, not a silently different loss
    bad = torch.ones(5, device="cuda")
    assert lib.sfh_uv_loss(_ptr(uvc), _ptr(gtc), _ptr(bad), 5, B, C, H, W, lam, 1, _ptr(duv), _ptr(loss), _stream()) != 0
    if W != 5:
        with pytest.raises(RuntimeError):     # torch's own broadcasting error for the same call
            train_ref.per_sample_weighted(torch.zeros(5, C, H, W), torch.ones(5))


@pytest.mark.parametrize("optimizer,mode", [("SGD", "img+mask+uv"), ("Adam", "img+mask"), ("RMSprop", "img+mask+uv")])
def test_train_step_with_uv_head_and_the_other_optimizers(T, optimizer, mode):
    """TrainStep on a unet_uv model (train.py:136-144,203-208) with each optimizer of train.py:87-95, one frame per step (the
    reference's uv loss broadcasts only for B == 1 or B == W): the five losses against the CPU restatement, and the
    post-step weights against torch's optimizer fed with the CPU restatement's clipped gradients."""
    import torch.nn.functional as F
    from sfh_amd.reconstructor import Reconstructor
    B, H, W = 1, 64, 96
    court = synth.load_court_template("ncaa_nc4_640x360", 4, B)[:, :, :H, :W].contiguous()
    poi = synth.load_court_poi("pitch", B)
    net = Reconstructor(court, poi, target_size=(W, H), unet_size=(W, H), warp_size=(W, H), unet_uv=True, resnet_input=mode)
    sd = synth.synth_state_dict(net.state_dict(), 57)
    net.load_state_dict(sd)
    x = synth.frames_to_float(synth.synth_frames_u8(B, H, W, seed=57))
    batch = _batch(B, H, W, poi.shape[1], 58)
    g = torch.Generator().manual_seed(59)
    batch["uv"] = torch.rand(B, 2, H, W, generator=g)
    lam = (2.0, 2.0, 8.0, 1.0)
    lr = 1e-3

    ref = train_ref.leaf_state(sd)
    params = [v for v in ref.values() if v.requires_grad]
    opt = {"SGD": lambda: torch.optim.SGD(params, lr=lr, weight_decay=1e-8, momentum=0.9),
           "Adam": lambda: torch.optim.Adam(params, lr=lr, betas=(0.9, 0.999), weight_decay=1e-8),
           "RMSprop": lambda: torch.optim.RMSprop(params, lr=lr, weight_decay=1e-8, momentum=0.9)}[optimizer]()
    pr = train_ref.forward_train(x, ref, court, poi, warp_size=(W, H), unet_size=(W, H), target_size=(W, H), resnet_input=mode)
    lref = train_ref.losses(pr, batch, lambdas=lam)
    luv = train_ref.per_sample_weighted(F.mse_loss(pr["uv"], batch["uv"], reduction="none"), batch["weight"]) * 2.0
    (lref["total"] + luv).backward()
    torch.nn.utils.clip_grad_value_(params, 0.1)
    opt.step()

    net.court_img, net.court_poi = court.cuda(), poi.cuda()
    net.cuda().train()
    ts = T.TrainStep(net, lr=lr, weight_decay=1e-8, optimizer=optimizer, uv_loss="MSE", uv_lambda=2.0)
    lh = ts.step(x.cuda(), {k: v.cuda() for k, v in batch.items()}).cpu()
    torch.cuda.synchronize()
    assert lh.numel() == 5
    want = torch.tensor([lref["seg"].item(), lref["rec"].item(), lref["consist"].item(), lref["reproj"].item(), luv.item()],
                        dtype=torch.float64)
    assert (lh - want).abs().max().item() < 2e-3 * want.abs().max().item(), (lh, want)
    new = net.state_dict()
    agree = total = 0
    step = {"SGD": lr * 0.1, "Adam": lr, "RMSprop": 10 * lr}[optimizer]      # size of a first step on a clipped gradient
    for k, v in ref.items():
        if not v.requires_grad:
            continue
        d_ref = (v.detach() - sd[k]).double()
        d_hip = (new[k].cpu() - sd[k]).double()
        agree += ((d_ref - d_hip).abs() <= 0.05 * step).sum().item()
        total += d_ref.numel()
    assert agree / total > 0.9, agree / total
    # the optimizer state travels under the optimizer's name
    st = ts.state_dict()
    assert st["optimizer"] == optimizer and st["global_step"] == 1
    other = T.TrainStep(net, optimizer="SGD" if optimizer != "SGD" else "Adam")
    with pytest.raises(RuntimeError, match="belongs to"):
        other.load_state_dict(st)


def test_train_step_gradient_overflow_lowers_the_scale_for_good(T, monkeypatch):
    """a step whose GRADIENTS leave the fp16 range (forward clean) is repeated on the same two-plane fp16
    kernels with a lower power-of-two gradient scale, kept for the steps that follow - not in bf16x6.  Planted by raising
    the scale 2^10 above where TrainStep puts it: the layer gradients (two to three orders of magnitude above the head
    gradients) then pass 16376."""
    from sfh_amd.reconstructor import Reconstructor
    monkeypatch.setenv("SFH_TRAIN_PRECISION", "f16x3")
    B, H, W = 2, 64, 96
    court = synth.load_court_template("ncaa_nc4_640x360", 4, B)[:, :, :H, :W].contiguous().cuda()
    poi = synth.load_court_poi("pitch", B).cuda()
    x = synth.frames_to_float(synth.synth_frames_u8(B, H, W, seed=9)).cuda()
    batch = {k: v.cuda() for k, v in _batch(B, H, W, poi.shape[1], 10).items()}

    def run(shift):
        net = Reconstructor(court, poi, target_size=(W, H), unet_size=(W, H), warp_size=(W, H))
        net.load_state_dict(synth.synth_state_dict(net.state_dict(), 9))
        net.cuda().train()
        ts = T.TrainStep(net, lr=1e-5)
        ts.grad_scale_shift = shift
        losses = ts.loss_and_grads(x, batch)
        torch.cuda.synchronize()
        stats = {k: v.clone() for k, v in net.state_dict().items() if "running" in k}
        return losses.clone(), ts.gflat.clone(), stats, ts, net

    l0, g0, b0, ts0, _ = run(0)
    assert (ts0.range_rescales, ts0.range_fallbacks, ts0.grad_scale_shift) == (0, 0, 0)
    l1, g1, b1, ts1, net1 = run(12)
    assert ts1.range_rescales >= 1 and ts1.range_fallbacks == 0 and ts1.grad_scale_shift <= 6
    assert torch.allclose(l0, l1, rtol=1e-6, atol=0) and _relerr(g1, g0) < 1e-3
    for k in b0:      # the repeated step started from the restored BatchNorm statistics (moved once, not twice)
        assert torch.allclose(b0[k].double(), b1[k].double(), rtol=1e-5, atol=1e-7), k
    # sticky: the next step repeats nothing, and the shift travels with the optimizer state
    before = ts1.range_rescales
    ts1.step(x, batch)
    torch.cuda.synchronize()
    assert ts1.range_rescales == before and ts1.range_fallbacks == 0
    assert ts1.state_dict()["grad_scale_shift"] == ts1.grad_scale_shift


@pytest.mark.parametrize("seed", [0, 1])
def test_train_step_on_trained_like_checkpoints(T, seed, monkeypatch):
    """One TrainStep iteration from a checkpoint with TRAINED statistics (synth.trained_like_state_dict: conv rows over three
    decades of scale, negative / dead gammas, 1 % weight outliers, one layer scaled by 2^+12 (seed 0) / 2^-12 (seed 1)) against
    the CPU restatement.  Under batch-statistics BatchNorm the 2^-12 layer's pre-activation gradient is 4096x its
    neighbours': the f16x3 backward pass must hold it by lowering its gradient scale (range_rescales), not by leaving the
    two-plane fp16 kernels (range_fallbacks == 0)."""
    from sfh_amd.reconstructor import Reconstructor
    monkeypatch.setenv("SFH_TRAIN_PRECISION", "f16x3")
    B, H, W = 4, 96, 128
    lr, lam = 1e-4, (2.0, 2.0, 8.0, 1.0)
    court = synth.load_court_template("ncaa_nc4_640x360", 4, B)[:, :, :H, :W].contiguous()
    poi = synth.load_court_poi("pitch", B)
    net = Reconstructor(court, poi, target_size=(W, H), unet_size=(W, H), warp_size=(W, H))
    sd, info = synth.trained_like_state_dict(net.state_dict(), seed, return_info=True)
    net.load_state_dict(sd)
    x = synth.frames_to_float(synth.synth_frames_u8(B, H, W, seed=61 + seed))
    batch = _batch(B, H, W, poi.shape[1], 62 + seed)

    ref = train_ref.leaf_state(sd)
    params = [v for v in ref.values() if v.requires_grad]
    opt = torch.optim.RMSprop(params, lr=lr, weight_decay=1e-8, momentum=0.9)
    pr = train_ref.forward_train(x, ref, court, poi, warp_size=(W, H), unet_size=(W, H), target_size=(W, H))
    lref = train_ref.losses(pr, batch, lambdas=lam)
    lref["total"].backward()
    torch.nn.utils.clip_grad_value_(params, 0.1)
    opt.step()

    net.court_img, net.court_poi = court.cuda(), poi.cuda()
    net.cuda().train()
    ts = T.TrainStep(net, lr=lr, weight_decay=1e-8)
    lh = ts.step(x.cuda(), {k: v.cuda() for k, v in batch.items()}).cpu()
    torch.cuda.synchronize()
    want = torch.tensor([lref["seg"].item(), lref["rec"].item(), lref["consist"].item(), lref["reproj"].item()], dtype=torch.float64)
    new = net.state_dict()
    agree = total = 0
    for k, v in ref.items():
        if not v.requires_grad:
            continue
        d_ref = (v.detach() - sd[k]).double()
        d_hip = (new[k].cpu() - sd[k]).double()
        agree += ((d_ref - d_hip).abs() <= 0.05 * (10 * lr)).sum().item()
        total += v.numel()
    print(f"trained-like seed {seed} ({info['scaled_layer']} x 2^{info['scaled_layer_exp']}): losses hip {lh.tolist()} cpu {want.tolist()}; "
          f"updates agreeing {agree / total:.4f}; range_rescales {ts.range_rescales} range_fallbacks {ts.range_fallbacks} "
          f"grad_scale_shift {ts.grad_scale_shift}")
    assert ts.range_fallbacks == 0
    assert (lh - want).abs().max().item() < 2e-3 * want.abs().max().item(), (lh, want)
    assert agree / total > 0.9, agree / total
