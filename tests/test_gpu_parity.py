"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle and the
committed golden vectors.  Tolerances (BASELINE.json north_star): homography and warp
within 1e-4 abs, nearest-mode warp / argmax / POI pixel integer-exact."""
import json
import math
import os
import warnings

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import torch_ref, warp_ref  # noqa: E402
from sfh_amd import synth, modules  # noqa: E402

torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))


@pytest.fixture(scope="module")
def E():
    from sfh_amd import engine, _lib
    _lib.load()
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return engine


def _nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous().cuda()


def _nchw(y):
    return y.permute(0, 3, 1, 2).contiguous().cpu()


def _maxerr(a, b):
    return (torch.as_tensor(a).float() - torch.as_tensor(b).float()).abs().max().item()


def _mods_to_cuda(m, seed):
    sd = synth.synth_state_dict(m.state_dict(), seed)
    m.load_state_dict(sd)
    return m.cuda().eval(), sd


# ---------------------------------------------------------------- conv building blocks
def _run_double_conv(E, block, x_nhwc, B, H, W, c0, src1=None, c1=0, pool0=False, pad1=(0, 0), tile=None):
    (cv1, bn1), (cv2, bn2) = block.convs()
    l1 = E.PackedConv(cv1.weight, cv1.bias, bn1, 3, c0, c1)
    l2 = E.PackedConv(cv2.weight, cv2.bias, bn2, 3, cv1.out_channels)
    mid = torch.empty((B, H, W, cv1.out_channels), device="cuda")
    out = torch.empty((B, H, W, cv2.out_channels), device="cuda")
    l1.run(x_nhwc, B, H, W, mid, src1=src1, pool0=pool0, pad1=pad1, tile=tile)
    l2.run(mid, B, H, W, out, tile=tile)
    torch.cuda.synchronize()
    return out


@pytest.mark.parametrize("tile", [None, 0, 1, 2])
def test_double_conv_golden(E, golden_blocks, tile):
    g = golden_blocks
    m, _ = _mods_to_cuda(modules.DoubleConv(3, 64), 11)
    x = torch.from_numpy(g["dc_3_64.x"])
    x4 = torch.zeros(2, 20, 24, 4)
    x4[..., :3] = x.permute(0, 2, 3, 1)
    y = _run_double_conv(E, m, x4.cuda(), 2, 20, 24, 3, tile=tile)
    assert _maxerr(_nchw(y), g["dc_3_64.y"]) < 2e-5
    m, _ = _mods_to_cuda(modules.DoubleConv(64, 128, 64), 12)
    x = torch.from_numpy(g["dc_64_128_m64.x"])
    y = _run_double_conv(E, m, _nhwc(x), 1, 17, 23, 64, tile=tile)
    assert _maxerr(_nchw(y), g["dc_64_128_m64.y"]) < 5e-5


def _frame_h2(x_nchw, exp=2):
    """the FH2 frame tensor of sfh_frame_to_h2 (held as float32 (B,H,W,4): 16 bytes per pixel) + the fp32 NHWC copy"""
    import ctypes
    from sfh_amd import _lib
    lib = _lib.load()
    B, C, H, W = x_nchw.shape
    x = x_nchw.contiguous().cuda()
    nhwc4 = torch.empty((B, H, W, 4), device="cuda")
    fh2 = torch.empty((B, H, W, 4), device="cuda")
    word = torch.zeros(1, dtype=torch.int32, device="cuda")
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    _lib.check(lib.sfh_frame_to_h2(p(x), p(nhwc4), p(fh2), B, C, H, W, exp, None, p(word),
                                   ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "frame_to_h2")
    return fh2, nhwc4, word


@pytest.mark.parametrize("hw", [(20, 24), (45, 80), (37, 70)])
def test_first_layer_on_the_fp16_matrix_cores(E, golden_blocks, hw):
    """DoubleConv's first conv (3 -> 64, unet/unet_parts.py:15) in "f16x3" arithmetic (sfh_conv3x3_c4h2_fwd over the FH2
    frame tensor): against the reference-class golden, against the fp32-MFMA kernel it replaces on odd sizes (tile
    seams, frame borders, the zero rows between frames), and a NaN pixel reaches exactly its 3x3 neighbourhood - the
    padding taps 9 .. 15 of the K axis must not pull neighbours in through zero weights."""
    g = golden_blocks
    m, _ = _mods_to_cuda(modules.DoubleConv(3, 64), 11)
    (cv1, bn1), (cv2, bn2) = m.convs()
    lh = E.PackedConv(cv1.weight, cv1.bias, bn1, 3, 3, frame_h2=True)
    lf = E.PackedConv(cv1.weight, cv1.bias, bn1, 3, 3)
    l2 = E.PackedConv(cv2.weight, cv2.bias, bn2, 3, 64, fmt="h2")
    assert lh.c4h2 and lf.c4 and not lf.c4h2
    if hw == (20, 24):
        x = torch.from_numpy(g["dc_3_64.x"])
    else:
        x = torch.rand((3, 3) + hw, generator=torch.Generator().manual_seed(hw[0]))
    B, _, H, W = x.shape
    fh2, nhwc4, word = _frame_h2(x)
    assert torch.equal(nhwc4[..., :3].cpu(), x.permute(0, 2, 3, 1)) and float(nhwc4[..., 3].abs().max()) == 0.0
    mid_h = E.split_empty("h2", B, H, W, 64, "cuda")
    mid_f = E.split_empty("h2", B, H, W, 64, "cuda")
    lh.run(fh2, B, H, W, mid_h)
    lf.run(nhwc4, B, H, W, mid_f)
    yh = torch.empty((B, H, W, 64), device="cuda")
    yf = torch.empty((B, H, W, 64), device="cuda")
    a, b = E.s3_to_f32(mid_h), E.s3_to_f32(mid_f)
    torch.cuda.synchronize()
    assert float((a - b).abs().max()) < 2e-5 * max(1.0, float(b.abs().max()))
    assert abs(float(word.view(torch.float32).item()) - 4.0 * float(x.abs().max())) < 1e-5   # the range word: max |x| * 2^2
    if hw == (20, 24):
        l2.run(mid_h, B, H, W, yh)
        torch.cuda.synchronize()
        assert _maxerr(_nchw(yh), g["dc_3_64.y"]) < 2e-5
    # fp32 destination of the same kernel, and the NaN footprint
    lh.run(fh2, B, H, W, yh)
    xn = x.clone()
    xn[1, 2, 7, 9] = float("nan")
    fh2n, _, wn = _frame_h2(xn)
    lh.run(fh2n, B, H, W, yf)
    torch.cuda.synchronize()
    assert float((yh - a).abs().max()) < 1e-5 * max(1.0, float(a.abs().max()))
    assert int(wn.item()) >= 0x7F800000 or int(wn.item()) < 0        # the range word holds a non-finite pattern
    # the H2 split SATURATES a NaN (the range word reports it and the model re-runs in bf16x6); what this kernel must
    # guarantee is locality: outputs outside the pixel's 3x3 neighbourhood are those of the clean frame
    changed = ((yf - yh).abs().amax(dim=3) > 0).cpu()
    where = changed.nonzero()
    assert len(where) > 0 and (where[:, 0] == 1).all()
    assert int(where[:, 1].min()) >= 6 and int(where[:, 1].max()) <= 8 and int(where[:, 2].min()) >= 8 and int(where[:, 2].max()) <= 10


# ---------------------------------------------------------------- split-bf16 (S3) conv path
def _run_double_conv_s3(E, block, x_nhwc, B, H, W, c0, src1=None, c1=0, pad1=(0, 0), tile=None, pool=False, fmt="s3"):
    (cv1, bn1), (cv2, bn2) = block.convs()
    l1 = E.PackedConv(cv1.weight, cv1.bias, bn1, 3, c0, c1, fmt=fmt)
    l2 = E.PackedConv(cv2.weight, cv2.bias, bn2, 3, cv1.out_channels, fmt=fmt)
    mid = E.split_empty(fmt, B, H, W, cv1.out_channels, "cuda")
    out = torch.empty((B, H, W, cv2.out_channels), device="cuda")
    l1.run(E.f32_to_split(x_nhwc, fmt), B, H, W, mid, src1=None if src1 is None else E.f32_to_split(src1, fmt), pad1=pad1,
           tile=tile)
    pooled = torch.empty((B, H // 2, W // 2, cv2.out_channels), device="cuda") if pool else None
    l2.run(mid, B, H, W, out, tile=tile, dst_pool=pooled)
    torch.cuda.synchronize()
    return out, pooled


@pytest.mark.parametrize("fmt", ["s3", "h2"])
@pytest.mark.parametrize("tile", [None, 0, 1, 2, 3, 4])
def test_double_conv_s3_golden(E, golden_blocks, tile, fmt):
    g = golden_blocks
    m, _ = _mods_to_cuda(modules.DoubleConv(64, 128, 64), 12)
    x = torch.from_numpy(g["dc_64_128_m64.x"])
    y, yp = _run_double_conv_s3(E, m, _nhwc(x), 1, 17, 23, 64, tile=tile, pool=True, fmt=fmt)
    assert _maxerr(_nchw(y), g["dc_64_128_m64.y"]) < 5e-5
    want_pool = torch.nn.functional.max_pool2d(_nchw(y), 2)
    assert torch.equal(_nchw(yp), want_pool)          # fused MaxPool2d(2) of the same values: exact


@pytest.mark.parametrize("fmt", ["s3", "h2"])
def test_up_transposed_conv_concat_s3_golden(E, golden_blocks, fmt):
    g = golden_blocks
    m, _ = _mods_to_cuda(modules.Up(128, 64, False), 14)
    x1 = torch.from_numpy(g["up_128_64.x1"])
    x2 = torch.from_numpy(g["up_128_64.x2"])
    up = E.PackedConv(m.up.weight, m.up.bias, None, 1, 128, relu=False, transposed=True, fmt=fmt)
    upb = E.split_empty(fmt, 1, 20, 18, 64, "cuda")
    up.run(E.f32_to_split(_nhwc(x1), fmt), 1, 10, 9, upb)
    torch.cuda.synchronize()
    ref_up = torch.nn.functional.conv_transpose2d(x1, m.up.weight.cpu(), m.up.bias.cpu(), stride=2)
    assert _maxerr(_nchw(E.s3_to_f32(upb)), ref_up) < 2e-5
    y, _ = _run_double_conv_s3(E, m.conv, _nhwc(x2), 1, 21, 19, 64, src1=E.s3_to_f32(upb), c1=64, fmt=fmt)
    assert _maxerr(_nchw(y), g["up_128_64.y"]) < 5e-5


def test_s3_split_is_exact(E):
    g = synth._rng(4, "s3")
    x = torch.from_numpy((g.normal(0, 1, (2, 5, 7, 64)) * np.exp(g.uniform(-20, 20, (2, 5, 7, 64)))).astype(np.float32)).cuda()
    s = E.f32_to_s3(x)
    assert torch.equal(E.s3_to_f32(s), x)
    # (B,H,C/32,3,4,W,8): sum the planes, then bring (block, group, x, lane) back to (x, channel)
    rec = s.float().sum(3).permute(0, 1, 4, 2, 3, 5).reshape(x.shape)
    assert torch.equal(rec, x)


@pytest.mark.parametrize("case", [(16, 12, 20, 512, 512), (3, 23, 40, 256, 128), (2, 45, 80, 64, 64), (5, 7, 9, 32, 192),
                                  (1, 25, 41, 96, 64), (16, 22, 40, 1024, 64)])
def test_conv_small_map_kernel_gives_the_standard_kernel_s_bits(E, case):
    """Round 5: sfh_conv_small_fwd (12x20-pixel x 32-cout workgroups, 15 pixel groups of 4x4, halo AND weights through LDS; for
    launches whose standard grid leaves the chip under-filled: ResNet layer4 at batch 16) against conv_s3_kernel on the same
    operands and the same packed weights: every output accumulates the same products in the same order => identical bits.
    H2 destination with an H2 residual + ReLU, fp32 destination without; frames that are whole tiles (12x20), partial tiles in
    both directions (23x40, 45x80, 7x9, 25x41), stored channels beyond the used ones; both LDS-buffering variants (grids of
    at most / more than 256 workgroups); the range word reports the same maximum; and against an fp64 conv."""
    B, H, W, cin, cout = case
    g = synth._rng(7, f"small{case}")
    x = torch.from_numpy(g.normal(0, 1, (B, H, W, cin + 32)).astype(np.float32)).cuda()     # 32 stored channels the conv ignores
    w = torch.from_numpy((g.normal(0, 1, (cout, cin, 3, 3)) * (2.0 / (9 * cin)) ** 0.5).astype(np.float32)).cuda()
    b = torch.from_numpy(g.normal(0, 0.1, (cout,)).astype(np.float32)).cuda()
    res = torch.from_numpy(g.normal(0, 1, (B, H, W, cout)).astype(np.float32)).cuda()
    bn = torch.nn.BatchNorm2d(cout).cuda().eval()
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5)
        bn.running_mean.uniform_(-0.2, 0.2)
        bn.running_var.uniform_(0.5, 1.5)
    pc = E.PackedConv(w, b, bn, 3, cin, fmt="h2")
    xs, rs = E.f32_to_split(x, "h2"), E.f32_to_split(res, "h2")
    outs, words = {}, {}
    for small in (False, True):
        y = E.split_empty("h2", B, H, W, cout, "cuda")
        word = torch.zeros(1, dtype=torch.int32, device="cuda")
        pc.run(xs, B, H, W, y, residual=rs, small=small, range_word=word.data_ptr(), exp_src=2, exp_dst=1, exp_res=2)
        yf = torch.empty((B, H, W, cout), device="cuda")
        pc.relu = False
        pc.run(xs, B, H, W, yf, small=small, exp_src=2)
        pc.relu = True
        torch.cuda.synchronize()
        outs[small], words[small] = (y.clone(), yf), int(word.item())
    assert torch.equal(outs[True][0], outs[False][0]) and torch.equal(outs[True][1], outs[False][1])
    assert words[True] == words[False] and words[True] > 0
    xin = x[..., :cin].cpu().double().permute(0, 3, 1, 2)
    z = torch.nn.functional.conv2d(xin, w.cpu().double(), b.cpu().double(), padding=1)
    bnd = bn.double().cpu()
    want = bnd(z)
    tol = 3e-5 if cin <= 512 else 1e-4      # (fp32 accumulation over K = 9 * cin products; the bit-equality above is the test)
    assert _maxerr(_nchw(outs[True][1]).double(), want) < tol
    want_r = torch.relu(want + E.s3_to_f32(rs).cpu().double().permute(0, 3, 1, 2))
    assert _maxerr(_nchw(E.s3_to_f32(outs[True][0], 1)).double(), want_r) < tol
    bn.float().cuda()
    # the engine's own rule: only grids that leave more than an eighth of the CUs idle AND gain workgroups from the finer tiling
    if case == (16, 12, 20, 512, 512):
        calls = []
        lib = E._lib.load()
        real = lib.sfh_conv_small_fwd
        try:
            E._lib._lib.sfh_conv_small_fwd = lambda *a: (calls.append(1), real(*a))[1]
            pc.run(xs, B, H, W, E.split_empty("h2", B, H, W, cout, "cuda"))
        finally:
            E._lib._lib.sfh_conv_small_fwd = real
        assert calls == [1]            # ResNet layer4 at batch 16: 192 standard workgroups -> the small-map kernel
    with pytest.raises(ValueError):
        pc.run(xs, B, H, W, E.split_empty("h2", B, H, W, cout, "cuda"), small=True,
               dst_pool=E.split_empty("h2", B, H // 2, W // 2, cout, "cuda"))


@pytest.mark.parametrize("tile", [0, 1, 2])
@pytest.mark.parametrize("hw", [(17, 23), (45, 80), (22, 40)])
def test_conv_h2_eight_wave_workgroup_matches_four_wave(E, tile, hw):
    """The 8-wave workgroup (256 pixels x 128 couts, double-buffered; sfh_conv_desc.wg_couts = 128) against the
    4-wave one on the same operands: same products, same accumulation order per output => identical bits; and
    against an fp64 conv.  3x3 with pool output, residual, two sources with a pad offset; long K."""
    H, W = hw
    g = synth._rng(5, f"w8{H}x{W}")
    B, c0, c1, cout = 2, 128, 64, 256
    x0 = torch.from_numpy(g.normal(0, 1, (B, H, W, c0)).astype(np.float32)).cuda()
    x1 = torch.from_numpy(g.normal(0, 1, (B, H - 1, W - 1, c1)).astype(np.float32)).cuda()
    w = torch.from_numpy((g.normal(0, 1, (cout, c0 + c1, 3, 3)) * 0.03).astype(np.float32)).cuda()
    b = torch.from_numpy(g.normal(0, 0.1, (cout,)).astype(np.float32)).cuda()
    res = torch.from_numpy(g.normal(0, 1, (B, H, W, cout)).astype(np.float32)).cuda()
    pc = E.PackedConv(w, b, None, 3, c0, c1, fmt="h2")
    xs0, xs1 = E.f32_to_split(x0, "h2"), E.f32_to_split(x1, "h2")
    outs = {}
    for wg in (64, 128):
        y = E.split_empty("h2", B, H, W, cout, "cuda")
        yp = E.split_empty("h2", B, H // 2, W // 2, cout, "cuda")
        pc.run(xs0, B, H, W, y, src1=xs1, pad1=(1, 0), dst_pool=yp, residual=E.f32_to_split(res, "h2"), tile=tile, wg_couts=wg)
        torch.cuda.synchronize()
        outs[wg] = (E.s3_to_f32(y), E.s3_to_f32(yp))
    assert torch.equal(outs[64][0], outs[128][0]) and torch.equal(outs[64][1], outs[128][1])
    xin = torch.zeros(B, c0 + c1, H, W, dtype=torch.float64)
    xin[:, :c0] = x0.cpu().double().permute(0, 3, 1, 2)
    xin[:, c0:, 1:, :W - 1] = x1.cpu().double().permute(0, 3, 1, 2)
    want = torch.nn.functional.conv2d(xin, w.cpu().double(), b.cpu().double(), padding=1)
    # the residual went through the H2 format (22 bits) before it was added
    want = torch.relu(want + E.s3_to_f32(E.f32_to_split(res, "h2")).cpu().double().permute(0, 3, 1, 2))
    assert _maxerr(_nchw(outs[128][0]).double(), want) < 3e-5
    # fp32 destination, no residual, single source (the partial of a fused Up block)
    pc2 = E.PackedConv(w[:, :c0].contiguous(), None, None, 3, c0, relu=False, fmt="h2")
    f = {}
    for wg in (64, 128):
        yf = torch.empty((B, H, W, cout), device="cuda")
        pc2.run(xs0, B, H, W, yf, tile=tile, wg_couts=wg)
        torch.cuda.synchronize()
        f[wg] = yf
    assert torch.equal(f[64], f[128])


@pytest.mark.parametrize("fmt", ["h2", "s3"])
@pytest.mark.parametrize("hw,stride,ks", [((12, 20), 1, 3), ((23, 40), 1, 2), ((23, 41), 2, 2), ((9, 7), 1, 8)])
def test_conv_split_k_matches_the_unsplit_launch(E, fmt, hw, stride, ks):
    """sfh_conv_desc.ksplit: the K loop over 256 input channels in ks parts (uneven for 3), fp32 partial slabs,
    sfh_splitk_finish adds shift + residual + ReLU and writes the split format (its own exponent in H2): equal to the
    one-launch conv up to the accumulation order, and the range word sees the same maximum."""
    g = synth._rng(12, f"splitk{hw}{stride}{ks}")
    B, (H, W), cin, cout = 2, hw, 256, 128
    ho, wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    w = torch.from_numpy((g.normal(0, 1, (cout, cin, 3, 3)) * (2.0 / (9 * cin)) ** 0.5).astype(np.float32)).cuda()
    bn = torch.nn.BatchNorm2d(cout).cuda().eval()
    with torch.no_grad():
        bn.running_mean.uniform_(-0.1, 0.1); bn.running_var.uniform_(0.5, 1.5); bn.weight.uniform_(0.5, 1.5); bn.bias.uniform_(-0.3, 0.3)
    pc = E.PackedConv(w, None, bn, 3, cin, stride=stride, fmt=fmt)
    x = torch.from_numpy(g.normal(0, 1, (B, H, W, cin)).astype(np.float32)).cuda()
    res = torch.from_numpy(g.normal(0, 1, (B, ho, wo, cout)).astype(np.float32)).cuda()
    xs = E.f32_to_split(x, fmt)
    rs = E.f32_to_split(res, fmt, exp=1)
    rw = torch.zeros(2, dtype=torch.int32, device="cuda")
    outs = []
    for k, word in ((0, 0), (ks, 1)):
        y = E.split_empty(fmt, B, ho, wo, cout, "cuda")
        slabs = torch.full((k, B, ho, wo, cout), float("nan"), device="cuda") if k else None
        pc.run(xs, B, H, W, y, residual=rs, exp_res=1, exp_dst=0, range_word=rw.data_ptr() + 4 * word, ksplit=k, slabs=slabs)
        outs.append(E.s3_to_f32(y, exp=0))
    torch.cuda.synchronize()
    a, b = outs
    assert float((a - b).abs().max()) <= 2e-6 * max(1.0, float(a.abs().max()))
    assert float(a.abs().max()) > 0.5 and bool((a >= 0).all())
    if fmt == "h2":
        m = rw.cpu().numpy().view(np.float32)
        assert m[0] > 0 and abs(m[0] - m[1]) <= 2e-6 * m[0]
    want = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double().cpu(), w.double().cpu(), stride=stride, padding=1)
    sc = (bn.weight / torch.sqrt(bn.running_var + 1e-5)).double().cpu().view(1, -1, 1, 1)
    want = torch.relu((want - bn.running_mean.double().cpu().view(1, -1, 1, 1)) * sc + bn.bias.double().cpu().view(1, -1, 1, 1)
                      + E.s3_to_f32(rs, exp=1).double().cpu().permute(0, 3, 1, 2))
    assert float((b.double().cpu().permute(0, 3, 1, 2) - want).abs().max()) < 2e-5


def test_conv_h2_wg_couts_argument_is_checked(E):
    w = torch.zeros(64, 128, 3, 3, device="cuda")
    pc = E.PackedConv(w, None, None, 3, 128, fmt="h2")
    x = E.split_empty("h2", 1, 16, 16, 128, "cuda")
    y = E.split_empty("h2", 1, 16, 16, 64, "cuda")
    with pytest.raises(ValueError, match="wg_couts"):
        pc.run(x, 1, 16, 16, y, wg_couts=128)      # 64 couts: no 128-cout workgroup


def test_h2_split_format(E):
    """H2 (two fp16 planes of v * 2^2): 22 significand bits over the whole normal range, absolute error <= 2^-27
    below it, saturation + the overflow word beyond +-16376; the layout is the S3 layout with two planes."""
    g = synth._rng(4, "h2")
    mag = np.exp(g.uniform(np.log(2.0 ** -4), np.log(16000.0), (2, 5, 7, 64)))
    x = torch.from_numpy((np.sign(g.normal(0, 1, mag.shape)) * mag).astype(np.float32)).cuda()
    ovf = torch.zeros(1, dtype=torch.int32, device="cuda")
    s = E.f32_to_h2(x, ovf)
    back = E.s3_to_f32(s)
    rel = ((back - x).abs() / x.abs()).max().item()
    assert rel <= 2.0 ** -21, rel                                  # half an ulp of 22 bits, with slack for plane1's rounding
    assert int(ovf.item()) == 0
    # the planes themselves: (B,H,C/32,2,4,W,8) -> sum -> (x, channel) order, times 2^-6
    rec = (s.float().sum(3).permute(0, 1, 4, 2, 3, 5).reshape(x.shape)) * 2.0 ** -2
    assert torch.equal(rec, back)
    # tiny values: absolute error bound
    t = torch.from_numpy((g.normal(0, 1, (1, 3, 5, 32)) * 1e-4).astype(np.float32)).cuda()
    assert (E.s3_to_f32(E.f32_to_h2(t)) - t).abs().max().item() <= 2.0 ** -27
    # exactly representable values survive bit for bit (zeros, small integers / 64)
    q = torch.from_numpy(g.integers(-2000, 2000, (1, 3, 5, 32)).astype(np.float32) / 4).cuda()
    assert torch.equal(E.s3_to_f32(E.f32_to_h2(q)), q)
    # beyond the fp16 range: saturated, and the overflow word is raised
    big = x.clone()
    big[0, 0, 0, 0] = 50000.0
    sb = E.f32_to_h2(big, ovf)
    assert int(ovf.item()) == 1
    assert abs(E.s3_to_f32(sb)[0, 0, 0, 0].item() - 65504.0 / 4) < 1e-3
    # NaN / Inf cannot be carried by the format: they raise the same word (the caller repeats the work in a format
    # with fp32's range), they are not silently clamped
    for bad in (float("nan"), float("inf"), -float("inf")):
        ovf.zero_()
        big = x.clone()
        big[1, 2, 3, 4] = bad
        E.f32_to_h2(big, ovf)
        assert int(ovf.item()) == 1, bad
    # the exponent belongs to the tensor: at 2^-3 the same format carries |v| up to 65504 * 8 with the same 22 bits,
    # and the range word receives the largest |v * 2^e| (here 3.0e5 / 8), or the bit pattern of a NaN
    rw = torch.zeros(1, dtype=torch.int32, device="cuda")
    wide = x * 16.0
    wide[0, 1, 2, 3] = -3.0e5
    ovf.zero_()
    sw = E.f32_to_h2(wide, ovf, exp=-3, range_word=rw)
    back = E.s3_to_f32(sw, exp=-3)
    assert int(ovf.item()) == 0 and ((back - wide).abs() / wide.abs()).max().item() <= 2.0 ** -21
    assert np.array([rw.item()], np.int32).view(np.float32)[0] == np.float32(3.0e5 / 8)
    E.f32_to_h2(wide * 0.5, ovf, exp=-3, range_word=rw)            # a smaller maximum leaves the word alone
    assert np.array([rw.item()], np.int32).view(np.float32)[0] == np.float32(3.0e5 / 8)
    E.f32_to_h2(wide, ovf, exp=2, range_word=rw)                   # 1.2e6 at exponent 2: saturated, the word says by how much
    assert int(ovf.item()) == 1 and np.array([rw.item()], np.int32).view(np.float32)[0] == np.float32(1.2e6)
    big = x.clone()
    big[1, 2, 3, 4] = float("nan")
    E.f32_to_h2(big, None, range_word=rw)
    assert (int(rw.item()) & 0x7FFFFFFF) > 0x7F800000


def test_down_pool_on_load_golden(E, golden_blocks):
    g = golden_blocks
    m, _ = _mods_to_cuda(modules.Down(64, 128), 13)
    x = torch.from_numpy(g["down_64_128.x"])
    y = _run_double_conv(E, m.block, _nhwc(x), 1, 10, 9, 64, pool0=True)
    assert _maxerr(_nchw(y), g["down_64_128.y"]) < 5e-5


def test_up_transposed_conv_concat_golden(E, golden_blocks):
    g = golden_blocks
    m, _ = _mods_to_cuda(modules.Up(128, 64, False), 14)
    x1 = torch.from_numpy(g["up_128_64.x1"])
    x2 = torch.from_numpy(g["up_128_64.x2"])
    up = E.PackedConv(m.up.weight, m.up.bias, None, 1, 128, relu=False, transposed=True)
    upb = torch.empty((1, 20, 18, 64), device="cuda")
    up.run(_nhwc(x1), 1, 10, 9, upb)
    torch.cuda.synchronize()
    ref_up = torch.nn.functional.conv_transpose2d(x1, m.up.weight.cpu(), m.up.bias.cpu(), stride=2)
    assert _maxerr(_nchw(upb), ref_up) < 2e-5
    y = _run_double_conv(E, m.conv, _nhwc(x2), 1, 21, 19, 64, src1=upb, c1=64, pad1=(0, 0))
    assert _maxerr(_nchw(y), g["up_128_64.y"]) < 5e-5


def test_outconv_argmax_golden(E, golden_blocks):
    from sfh_amd import _lib
    g = golden_blocks
    m, _ = _mods_to_cuda(modules.OutConv(64, 4), 16)
    x = torch.from_numpy(g["outc_64_4.x"])
    xn = _nhwc(x)
    logits = torch.empty((1, 4, 9, 13), device="cuda")
    am = torch.empty((1, 9, 13), dtype=torch.uint8, device="cuda")
    lib = _lib.load()
    _lib.check(lib.sfh_outconv_fwd(E._ptr(xn), 64, E._ptr(m.conv.weight.detach()), E._ptr(m.conv.bias.detach()), 4,
                                   1, 9, 13, E._ptr(logits), E._ptr(am), None, 0, None, 0, E._stream()), "outconv")
    torch.cuda.synchronize()
    assert _maxerr(logits.cpu(), g["outc_64_4.y"]) < 1e-5
    assert torch.equal(am.cpu(), torch_ref.preds_to_masks(logits.cpu()))


@pytest.mark.parametrize("fmt", ["s3", "h2"])
@pytest.mark.parametrize("hw", [(24, 40), (22, 37), (45, 80)])
def test_conv_variants_s3_vs_oracle(E, hw, fmt):
    """the same ResNet block shapes on the split-bf16 kernel: stride-2 3x3 and 1x1, S3 residual."""
    H, W = hw
    blk = modules.BasicBlock(64, 128, 2, torch.nn.Sequential(torch.nn.Conv2d(64, 128, 1, stride=2, bias=False),
                                                             torch.nn.BatchNorm2d(128)))
    blk, sd = _mods_to_cuda(blk, 21)
    sd = {"b." + k: v for k, v in sd.items()}
    x = torch.from_numpy(synth._rng(21, f"x{H}x{W}").uniform(-1, 1, (2, 64, H, W)).astype(np.float32))
    want = torch_ref._basic_block(x, sd, "b", 2)
    xs = E.f32_to_split(_nhwc(x), fmt)
    ho, wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    c1 = E.PackedConv(blk.conv1.weight, None, blk.bn1, 3, 64, stride=2, fmt=fmt)
    c2 = E.PackedConv(blk.conv2.weight, None, blk.bn2, 3, 128, fmt=fmt)
    dn = E.PackedConv(blk.downsample[0].weight, None, blk.downsample[1], 1, 64, relu=False, stride=2, fmt=fmt)
    t = E.split_empty(fmt, 2, ho, wo, 128, "cuda")
    idn = E.split_empty(fmt, 2, ho, wo, 128, "cuda")
    out = torch.empty((2, ho, wo, 128), device="cuda")
    c1.run(xs, 2, H, W, t)
    dn.run(xs, 2, H, W, idn)
    c2.run(t, 2, ho, wo, out, residual=None)
    torch.cuda.synchronize()
    # residual add in S3 needs an S3 destination: run conv2 once more into S3 with the residual
    out3 = E.split_empty(fmt, 2, ho, wo, 128, "cuda")
    c2.run(t, 2, ho, wo, out3, residual=idn)
    torch.cuda.synchronize()
    assert _maxerr(_nchw(E.s3_to_f32(out3)), want) < 5e-5


@pytest.mark.parametrize("hw", [(24, 40), (22, 37), (45, 80)])
def test_conv_variants_vs_oracle(E, hw):
    """stride-2 3x3, stride-2 1x1, residual epilogue - the ResNet block shapes."""
    H, W = hw
    torch.manual_seed(0)
    blk = modules.BasicBlock(64, 128, 2, torch.nn.Sequential(torch.nn.Conv2d(64, 128, 1, stride=2, bias=False),
                                                             torch.nn.BatchNorm2d(128)))
    blk, sd = _mods_to_cuda(blk, 21)
    sd = {"b." + k: v for k, v in sd.items()}
    x = torch.from_numpy(synth._rng(21, f"x{H}x{W}").uniform(-1, 1, (2, 64, H, W)).astype(np.float32))
    want = torch_ref._basic_block(x, sd, "b", 2)
    xn = _nhwc(x)
    ho, wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    c1 = E.PackedConv(blk.conv1.weight, None, blk.bn1, 3, 64, stride=2)
    c2 = E.PackedConv(blk.conv2.weight, None, blk.bn2, 3, 128)
    dn = E.PackedConv(blk.downsample[0].weight, None, blk.downsample[1], 1, 64, relu=False, stride=2)
    t = torch.empty((2, ho, wo, 128), device="cuda")
    idn = torch.empty_like(t)
    out = torch.empty_like(t)
    c1.run(xn, 2, H, W, t)
    dn.run(xn, 2, H, W, idn)
    c2.run(t, 2, ho, wo, out, residual=idn)
    torch.cuda.synchronize()
    assert _maxerr(_nchw(out), want) < 5e-5


# ---------------------------------------------------------------- ResNet-STN
@pytest.mark.parametrize("precision", ["f16x3", "bf16x6", "fp32"])
@pytest.mark.parametrize("name,key,seed", [("resnet34", "resnet34_7.theta", 17), ("resnet18", "resnet18_7.theta", 18),
                                           ("resnet50", "resnet50_7.theta", 19),
                                           ("wide_resnet50_2", "wide_resnet50_2_7.theta", 20)])
def test_resnet_stn_golden(E, golden_blocks, name, key, seed, precision):
    g = golden_blocks
    rn, _ = _mods_to_cuda(modules.ResNetSTN(name, 7), seed)
    x = torch.from_numpy(g["resnet34_7.x"])
    rg = E.H2Ranges(torch.device("cuda")) if precision == "f16x3" else None
    eng = E.ResNetEngine(rn, 7, torch.device("cuda"), precision, ranges=rg)
    y = E.nchw_to_nhwc(x.cuda(), 8)
    theta = eng.run(y, 2, 72, 128)
    torch.cuda.synchronize()
    rescales = 0
    while rg is not None:
        # these randomly initialised ResNets (no trained BatchNorm) grow to activations of several thousand in
        # layer3 / layer4 (resnet34: 7346; the Bottleneck depths go beyond the 16376 of the default exponent): the
        # kernels leave the magnitudes in the range words, the saturated tensors get a smaller exponent and the
        # engine resumes at the first of them - what Reconstructor._guarded does
        bits = rg.read()
        bad, nonfinite = rg.saturated(bits)
        assert not nonfinite
        if not bad:
            break
        rg.reset_words()
        theta = eng.rerun(eng.first_step(rg.lower(bad, bits)))
        rescales += 1
        assert rescales < 40
    assert _maxerr(theta.cpu(), g[key]) < 1e-4
    if rg is not None:
        assert min(rg.headroom().values()) >= 1.0 and len(rg.peak) > 10


@pytest.mark.parametrize("fmt", ["h2", "s3"])
@pytest.mark.parametrize("hw", [(180, 320), (23, 41), (7, 5)])
def test_stem_maxpool_written_straight_into_the_split_format(E, fmt, hw):
    """sfh_maxpool3x3s2_split_fwd (the ResNetSTN's MaxPool2d(3, 2, 1), models/resnet.py:176, pooled and split in one pass)
    gives the bits of the two-launch path it replaces - and of torch's max_pool2d - also on odd sizes, with a non-default
    exponent, and it raises the range word like f32_to_h2 does."""
    import ctypes
    from sfh_amd import _lib
    lib = _lib.load()
    H, W = hw
    B, C = 2, 64
    x = torch.randn((B, H, W, C), generator=torch.Generator().manual_seed(H)).mul_(50.0).cuda()
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    y32 = torch.empty((B, Ho, Wo, C), device="cuda")
    _lib.check(lib.sfh_maxpool3x3s2_fwd(p(x), p(y32), B, H, W, C, st), "maxpool")
    want32 = torch.nn.functional.max_pool2d(x.permute(0, 3, 1, 2), 3, 2, 1).permute(0, 2, 3, 1)
    assert torch.equal(y32, want32)
    exp = -1 if fmt == "h2" else 2
    want = E.f32_to_split(y32, fmt, exp=exp) if fmt == "h2" else E.f32_to_split(y32, fmt)
    got = E.split_empty(fmt, B, Ho, Wo, C, "cuda")
    word = torch.zeros(1, dtype=torch.int32, device="cuda")
    _lib.check(lib.sfh_maxpool3x3s2_split_fwd(p(x), p(got), B, H, W, C, E._SPLIT[fmt][2], exp, None,
                                              p(word) if fmt == "h2" else None, st), "maxpool_split")
    torch.cuda.synchronize()
    assert torch.equal(got.view(torch.int16), want.view(torch.int16))
    if fmt == "h2":
        assert abs(float(word.view(torch.float32).item()) - float(y32.abs().max()) * 2.0 ** exp) < 1e-3


# ---------------------------------------------------------------- warp / POI / CE
def _thetas():
    ident = np.eye(3, dtype=np.float32)
    t = [ident, synth.REALISTIC_THETAS[0], synth.REALISTIC_THETAS[1]]
    g = synth._rng(5, "thetas")
    for _ in range(9):
        t.append((ident + g.normal(0, 0.15, (3, 3))).astype(np.float32))
    ns = ident.copy(); ns[2] = [0.9, 0.0, 0.0]
    t.append(ns)
    far = ident.copy(); far[0, 2] = 5.0
    t.append(far)
    z = ident.copy(); z[2] = [0.0, 0.0, 0.0]   # Z == 0 everywhere -> scale 1 branch
    t.append(z)
    return torch.from_numpy(np.stack(t)).reshape(-1, 1, 3, 3)


@pytest.mark.parametrize("mode", ["nearest", "bilinear"])
@pytest.mark.parametrize("size", [(640, 360), (1280, 720), (97, 61)])
@pytest.mark.parametrize("shared", [True, False])
def test_warp_vs_oracle(E, mode, size, shared):
    w, h = size
    theta = _thetas()
    B = theta.shape[0]
    if (w, h) == (97, 61):
        ids = synth._rng(3, "tmpl").integers(0, 4, (61 + 3, 97 + 5))
        tmpl = torch.from_numpy(ids.astype(np.float32) / 4.0)[None, None].repeat(B, 1, 1, 1)
    else:
        tmpl = synth.load_court_template(f"ncaa_nc4_{w}x{h}", 4, B)
    if not shared:
        tmpl = tmpl.clone()
        tmpl[1::2] = torch.flip(tmpl[1::2], dims=[3])
    want = warp_ref.homography_warp(theta, tmpl, h, w, mode)
    of, oi = E.homography_warp(theta.cuda(), tmpl.cuda(), h, w, mode == "nearest", scale=4.0,
                               want_f32=True, want_i32=True, shared_template=shared)
    torch.cuda.synchronize()
    if mode == "nearest":
        assert torch.equal(of.cpu(), want), f"{(of.cpu() != want).sum().item()} pixels differ"
        assert torch.equal(oi.cpu(), (want * 4).to(torch.int32))
    else:
        assert _maxerr(of.cpu(), want) < 1e-5
        assert (oi.cpu() - (want * 4).to(torch.int32)).abs().max().item() <= 1


def test_poi_vs_oracle(E):
    theta = _thetas()[:12]
    poi = synth.load_court_poi("pitch", 12)
    want = torch_ref.transform_poi(theta, poi)
    got = E.poi_project(theta.cuda(), poi.cuda())
    torch.cuda.synchronize()
    assert _maxerr(got.cpu(), want) < 1e-4
    ok = np.isfinite(want.numpy()).all(-1) & (np.abs(want.numpy()) < 4).all(-1)
    px_w = np.rint(want.numpy()[ok] * np.array([640, 360]))
    px_g = np.rint(got.cpu().numpy()[ok] * np.array([640, 360]))
    # pixel coordinates int(round(p*W)) (predict.py:383): exact except on .5 ties within fp32 noise
    frac = np.abs(want.numpy()[ok] * np.array([640, 360]) % 1 - 0.5)
    assert np.array_equal(px_w[frac > 1e-3], px_g[frac > 1e-3])


@pytest.mark.parametrize("resize", [False, True])
def test_consistency_ce_vs_oracle(E, resize):
    g = synth._rng(9, "ce")
    B, nc, H, W = 3, 4, 90, 112
    logits = torch.from_numpy(g.normal(0, 3, (B, nc, H, W)).astype(np.float32))
    hm, wm = (45, 56) if resize else (H, W)
    mask = torch.from_numpy(g.integers(0, nc, (B, hm, wm)).astype(np.int32))
    m = mask.float()
    if resize:
        m = torch.nn.functional.interpolate(m.unsqueeze(1), size=(H, W), mode="nearest").squeeze(1)
    want = torch.nn.functional.cross_entropy(logits, m.long(), reduction="none").mean(dim=(1, 2))
    got = E.consistency_ce(logits.cuda(), mask.cuda())
    torch.cuda.synchronize()
    assert _maxerr(got.cpu(), want) < 1e-5


@pytest.mark.parametrize("size", [(640, 360), (1280, 720), (97, 61), (320, 45)])
@pytest.mark.parametrize("shared", [True, False])
def test_warp_consistency_fused_kernel(E, size, shared):
    """Round 5: nearest warp + consistency CE fused (sfh_warp_consistency_fwd).  The mask is bit-identical to the
    oracle's warp (and to sfh_homography_warp_fwd's), the score equals torch's cross_entropy on that mask and the separate CE
    kernels' result; a second launch gives the same bits (the partial sums are added in a fixed order); thetas with Z = 0, out-of-range taps and widths that are not a multiple of 64 included."""
    w, h = size
    theta = _thetas()
    B = theta.shape[0]
    if (w, h) in ((640, 360), (1280, 720)):
        tmpl = synth.load_court_template(f"ncaa_nc4_{w}x{h}", 4, B)
    else:
        ids = synth._rng(3, "tmpl").integers(0, 4, (h + 3, w + 5))
        tmpl = torch.from_numpy(ids.astype(np.float32) / 4.0)[None, None].repeat(B, 1, 1, 1)
    if not shared:
        tmpl = tmpl.clone()
        tmpl[1::2] = torch.flip(tmpl[1::2], dims=[3])
    g = synth._rng(11, f"wce{w}x{h}")
    logits = torch.from_numpy(g.normal(0, 3, (B, 4, h, w)).astype(np.float32))
    want_mask = (warp_ref.homography_warp(theta, tmpl, h, w, "nearest") * 4).to(torch.int32)
    want = torch.nn.functional.cross_entropy(logits, want_mask.long(), reduction="none").mean(dim=(1, 2))
    lg, th, tm = logits.cuda(), theta.cuda(), tmpl.cuda()
    wm, score = E.warp_consistency(th, tm, lg, 4.0, shared_template=shared)
    _, wm2 = E.homography_warp(th, tm, h, w, True, scale=4.0, want_f32=False, want_i32=True, shared_template=shared)
    sep = E.consistency_ce(lg, wm2)
    wm_b, score_b = E.warp_consistency(th, tm, lg, 4.0, shared_template=shared)        # deterministic: the same bits again
    torch.cuda.synchronize()
    assert torch.equal(wm.cpu(), want_mask) and torch.equal(wm, wm2)
    assert torch.equal(wm_b, wm) and torch.equal(score_b, score)
    assert _maxerr(score.cpu(), want) < 1e-5, _maxerr(score.cpu(), want)
    assert _maxerr(score, sep) < 1e-5


@pytest.mark.parametrize("size", [(1280, 720), (194, 122), (640, 90)])
def test_warp_consistency_fused_kernel_with_a_warp_twice_the_logits_size(E, size):
    """predict.py's default geometry (predict.py:151-155): the warp is twice the logits' size, the reference scores through
    F.interpolate(mask, mode='nearest') - logit pixel (y, x) against mask pixel (2y, 2x).  The fused kernel warps every pixel and
    lets the lanes at even x of the even rows score: mask bit-identical to the oracle's warp, score equal to torch's
    cross_entropy through the nearest-resized mask and to the separate kernels'."""
    w, h = size
    theta = _thetas()
    B = theta.shape[0]
    if (w, h) == (1280, 720):
        tmpl = synth.load_court_template("ncaa_nc4_1280x720", 4, B)
    else:
        ids = synth._rng(3, "tmpl2").integers(0, 4, (h + 3, w + 5))
        tmpl = torch.from_numpy(ids.astype(np.float32) / 4.0)[None, None].repeat(B, 1, 1, 1)
    g = synth._rng(12, f"wce2_{w}x{h}")
    logits = torch.from_numpy(g.normal(0, 3, (B, 4, h // 2, w // 2)).astype(np.float32))
    want_mask = (warp_ref.homography_warp(theta, tmpl, h, w, "nearest") * 4).to(torch.int32)
    m = torch.nn.functional.interpolate(want_mask.float().unsqueeze(1), size=(h // 2, w // 2), mode="nearest").squeeze(1).long()
    assert torch.equal(m, want_mask[:, ::2, ::2].long())
    want = torch.nn.functional.cross_entropy(logits, m, reduction="none").mean(dim=(1, 2))
    lg, th, tm = logits.cuda(), theta.cuda(), tmpl.cuda()
    wm, score = E.warp_consistency(th, tm, lg, 4.0, shared_template=True, warp_hw=(h, w))
    _, wm2 = E.homography_warp(th, tm, h, w, True, scale=4.0, want_f32=False, want_i32=True, shared_template=True)
    sep = E.consistency_ce(lg, wm2)
    torch.cuda.synchronize()
    assert torch.equal(wm.cpu(), want_mask) and torch.equal(wm, wm2)
    assert _maxerr(score.cpu(), want) < 1e-5 and _maxerr(score, sep) < 1e-5
    with pytest.raises(ValueError):
        E.warp_consistency(th, tm, lg, 4.0, shared_template=True, warp_hw=(h // 2 + 1, w // 2))


@pytest.mark.parametrize("wh", [(112, 90), (640, 360), (160, 96)])
def test_single_kernel_up_block_gives_the_two_launch_bits(E, wh):
    """Round 5 (csrc/conv_upfused.hip; the engine's default for levels 3 and 4, SFH_UP_SINGLE): the first conv of a fused Up block as ONE kernel -
    the composed 2x2 conv over the low-resolution tensor and the skip-half 3x3 conv accumulate into the same registers, a wave per
    output-parity class - against the two-launch form (fp32 partial + acc_init): same products in the same order per output =>
    identical logits and theta, at every level, incl. the level whose skip tensor is one row larger than twice the low-resolution
    one (90 -> 45 -> 22: F.pad with diff 1)."""
    net, sd, court, poi = _model(wh, warp_with_nearest=True)
    net.load_state_dict(sd)
    net.cuda().eval()
    x = synth.smooth_frames(2, wh[1], wh[0], seed=19).cuda()
    outs = {}
    for single in ((), (4,), (1, 2, 3, 4)):
        net.invalidate_engines()
        un, _ = net._get_engines()
        un.up_single = set(single)
        with torch.no_grad():
            outs[single] = net.predict(x, consistency=False)
    for single in ((4,), (1, 2, 3, 4)):
        assert torch.equal(outs[single]["logits"], outs[()]["logits"]), single
        assert torch.equal(outs[single]["theta"], outs[()]["theta"]), single
    calls = []
    real = E.run_upfused
    E.run_upfused = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
    try:
        with torch.no_grad():
            net.predict(x, consistency=False)
    finally:
        E.run_upfused = real
    assert len(calls) == 4


def test_predict_uses_the_fused_warp_consistency_kernel_and_matches_the_separate_kernels(E):
    """predict(consistency=True) with a nearest warp of the logits' size takes the fused kernel; `fuse_warp_ce = False` (or a
    bilinear warp, or a warp of another size) the two separate ones: same mask bits, scores within 1e-5, same keys."""
    net, sd, court, poi = _model((112, 90), warp_with_nearest=True)
    net.load_state_dict(sd)
    net.cuda().eval()
    x = synth.smooth_frames(2, 90, 112, seed=19).cuda()
    calls = []
    real = E.warp_consistency
    E.warp_consistency = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
    try:
        with torch.no_grad():
            fused = net.predict(x, consistency=True, project_poi=True)
            assert len(calls) == 1
            net.fuse_warp_ce = False
            sep = net.predict(x, consistency=True, project_poi=True)
            assert len(calls) == 1
            noc = net.predict(x, consistency=False)
            assert len(calls) == 1 and "consist_score" not in noc
    finally:
        E.warp_consistency = real
    assert sorted(fused) == sorted(sep)
    assert torch.equal(fused["warp_mask"], sep["warp_mask"]) and torch.equal(fused["theta"], sep["theta"])
    assert torch.equal(noc["warp_mask"], sep["warp_mask"])
    assert _maxerr(fused["consist_score"], sep["consist_score"]) < 1e-5
    want = torch_ref.predict(x.cpu(), sd, court, poi, warp_size=(112, 90), unet_size=(112, 90), target_size=(112, 90))
    assert _maxerr(fused["consist_score"].cpu(), want["consist_score"]) < 2e-4


# ---------------------------------------------------------------- whole model
def _model(court_wh=(640, 360), B=2, **kw):
    from sfh_amd.reconstructor import Reconstructor
    w, h = court_wh
    court = synth.load_court_template("ncaa_nc4_640x360", 4, B)
    if (w, h) != (640, 360):
        court = court[:, :, :h, :w].contiguous()
    poi = synth.load_court_poi("pitch", B)
    net = Reconstructor(court.cuda(), poi.cuda(), target_size=(w, h), unet_size=(w, h), warp_size=(w, h), **kw)
    sd = synth.synth_state_dict(net.state_dict(), 19)
    return net, sd, court, poi


@pytest.mark.parametrize("precision", ["f16x3", "bf16x6", "fp32"])
def test_whole_net_small_golden(E, golden_blocks, precision):
    g = golden_blocks
    net, sd, court, poi = _model((112, 90), warp_with_nearest=True)
    net.precision = precision
    net.load_state_dict(sd)
    net.cuda().eval()
    x = synth.smooth_frames(2, 90, 112, seed=19)
    with torch.no_grad():
        out = net.predict(x.cuda(), consistency=True, project_poi=True)
    torch.cuda.synchronize()
    assert _maxerr(out["logits"].cpu(), g["net_90x112.logits"]) < 2e-4
    assert _maxerr(out["theta"].cpu(), g["net_90x112.theta"]) < 1e-4
    want = torch_ref.predict(x, sd, court, poi, warp_size=(112, 90), unet_size=(112, 90), target_size=(112, 90),
                             project_poi=True)
    assert out["warp_mask"].dtype == torch.int32 and out["theta"].shape == (2, 1, 3, 3)
    assert _maxerr(out["consist_score"].cpu(), want["consist_score"]) < 1e-3
    assert _maxerr(out["poi"].cpu(), want["poi"]) < 1e-4
    # warp of the GPU theta through the oracle must equal the GPU warp bit for bit
    wm = (warp_ref.homography_warp(out["theta"].cpu(), court, 90, 112, "nearest") * 4).to(torch.int32)
    assert torch.equal(out["warp_mask"].cpu(), wm)
    # argmax: exact wherever the oracle's top-2 margin exceeds the measured logit error
    lg = torch.from_numpy(g["net_90x112.logits"])
    top2 = torch.topk(lg, 2, dim=1).values
    safe = (top2[:, 0] - top2[:, 1]) > 1e-3
    assert torch.equal(out["logits"].cpu().argmax(1)[safe], lg.argmax(1)[safe])


@pytest.mark.parametrize("precision", ["f16x3", "bf16x6", "fp32"])
def test_full_640x360_golden(E, golden_full, precision):
    """BASELINE config C2 shape (B=2 of the 16): every output against the committed vector."""
    g = golden_full
    net, _, court, poi = _model((640, 360), warp_with_nearest=True)
    net.precision = precision
    sd = synth.synth_state_dict(net.state_dict(), 0)
    net.load_state_dict(sd)
    net.cuda().eval()
    x = synth.frames_to_float(synth.synth_frames_u8(2, 360, 640, seed=0))
    with torch.no_grad():
        out = net.predict(x.cuda(), consistency=True, project_poi=True)
    torch.cuda.synchronize()
    dtheta = _maxerr(out["theta"].cpu(), g["theta"])
    assert dtheta < 1e-4, dtheta
    assert _maxerr(out["logits"].cpu()[:, :, ::8, ::8], g["logits_sub"]) < 5e-4
    assert _maxerr(out["consist_score"].cpu(), g["consist"]) < 2e-3
    assert _maxerr(out["poi"].cpu(), g["poi"]) < 1e-4
    am = np.unpackbits(g["argmax_2bit"], axis=-1).reshape(2, 360, 640, 2)
    am = am[..., 0] * 2 + am[..., 1]
    margin = g["margin_f16"].astype(np.float32)
    mine = out["logits"].cpu().argmax(1).numpy()
    safe = margin > 2e-3
    assert np.array_equal(mine[safe], am[safe])
    assert (mine != am).mean() < 1e-4
    # warp mask: integer-exact wherever a 1e-4 change of theta cannot move the sample across a
    # template edge; overall mismatch must be tiny
    wm = out["warp_mask"].cpu().numpy()
    assert (wm != g["warp_mask"]).mean() < 2e-3


def test_forward_eval_and_batch_chunking(E):
    """forward() (bilinear warp + POI, models/reconstructor.py:160-194) against the oracle, and the
    sub-batch path used when an activation tensor would exceed the 4 GiB descriptor range."""
    net, sd, court, poi = _model((112, 90), B=4)
    net.load_state_dict(sd)
    net.cuda().eval()
    x = synth.smooth_frames(4, 90, 112, seed=23)
    with torch.no_grad():
        out = net.forward(x.cuda())
        want = torch_ref.forward(x, sd, court, poi, warp_size=(112, 90), unet_size=(112, 90), target_size=(112, 90))
        assert set(out) == set(want) == {"logits", "theta", "poi", "warp_mask"}
        assert out["warp_mask"].dtype == torch.float32
        assert _maxerr(out["theta"].cpu(), want["theta"]) < 1e-4
        assert _maxerr(out["poi"].cpu(), want["poi"]) < 1e-4
        wm = warp_ref.homography_warp(out["theta"].cpu(), court, 90, 112, "bilinear")
        assert _maxerr(out["warp_mask"].cpu(), wm) < 1e-5
        whole = net.predict(x.cuda(), consistency=True, project_poi=True)
        net._max_frames = lambda _x: 3            # force two sub-batches (3 + 1 frames)
        parts = net.predict(x.cuda(), consistency=True, project_poi=True)
    for k in whole:
        assert parts[k].shape == whole[k].shape and torch.equal(parts[k], whole[k]), k


def test_bilinear_up_variant_golden(E, golden_blocks):
    """Up(bilinear=True) (unet/unet_parts.py:48-50): 2x align_corners upsampling + DoubleConv
    with mid_channels, against the reference-class golden vector."""
    from sfh_amd import _lib
    g = golden_blocks
    m, _ = _mods_to_cuda(modules.Up(128, 64, True), 15)
    x1 = torch.from_numpy(g["upbl_128_64.x1"])
    x2 = torch.from_numpy(g["upbl_128_64.x2"])
    lib = _lib.load()
    x1n = _nhwc(x1)
    upb = torch.empty((1, 20, 18, 64), device="cuda")
    _lib.check(lib.sfh_upsample2x_bilinear_nhwc(E._ptr(x1n), E._ptr(upb), 1, 10, 9, 64, E._stream()), "upsample2x")
    torch.cuda.synchronize()
    want_up = torch.nn.functional.interpolate(x1, scale_factor=2, mode="bilinear", align_corners=True)
    assert _maxerr(_nchw(upb), want_up) < 1e-6
    y = _run_double_conv(E, m.conv, _nhwc(x2), 1, 21, 19, 64, src1=upb, c1=64, pad1=(0, 0))
    assert _maxerr(_nchw(y), g["upbl_128_64.y"]) < 5e-5


@pytest.mark.parametrize("precision", ["f16x3", "bf16x6", "fp32"])
def test_bilinear_unet_and_resize_paths(E, precision):
    """unet_bilinear=True end to end, plus input bilinear resize / logits nearest resize /
    warp_size != target_size (K12) against the oracle."""
    from sfh_amd.reconstructor import Reconstructor
    B = 2
    court = synth.load_court_template("ncaa_nc4_640x360", 4, B)[:, :, :120, :160].contiguous()
    poi = synth.load_court_poi("pitch", B)
    kw = dict(target_size=(128, 96), unet_size=(112, 80), warp_size=(160, 120), warp_with_nearest=True,
              unet_bilinear=True)
    net = Reconstructor(court.cuda(), poi.cuda(), **kw)
    net.precision = precision
    sd = synth.synth_state_dict(net.state_dict(), 29)
    net.load_state_dict(sd)
    net.cuda().eval()
    x = synth.smooth_frames(B, 96, 128, seed=29)
    with torch.no_grad():
        out = net.predict(x.cuda(), consistency=True, project_poi=True)
        # oracle with the bilinear Up wiring
        xr = torch.nn.functional.interpolate(x, size=(80, 112), mode="bilinear", align_corners=False)
        logits, _, _ = torch_ref.forward_unet(xr, sd, (112, 80), (112, 80), bilinear=True)
        logits = torch.nn.functional.interpolate(logits, size=(96, 128), mode="nearest")
        theta = torch_ref.resnet_stn(torch.cat((logits, x), 1), sd)
    assert out["logits"].shape == (B, 4, 96, 128) and out["warp_mask"].shape == (B, 120, 160)
    assert _maxerr(out["logits"].cpu(), logits) < 3e-4
    assert _maxerr(out["theta"].cpu(), theta) < 1e-4
    wm = (warp_ref.homography_warp(out["theta"].cpu(), court, 120, 160, "nearest") * 4)
    assert torch.equal(out["warp_mask"].cpu(), wm.to(torch.int32))
    m = torch.nn.functional.interpolate(wm.unsqueeze(1), size=(96, 128), mode="nearest").squeeze(1)
    ce = torch.nn.functional.cross_entropy(out["logits"].cpu(), m.long(), reduction="none").mean(dim=(1, 2))
    assert _maxerr(out["consist_score"].cpu(), ce) < 1e-4


def test_predict_resnet50_variant(E):
    """resnet_name='resnet50' (Bottleneck blocks, models/resnet.py:85-140) end to end."""
    from sfh_amd.reconstructor import Reconstructor
    B = 2
    court = synth.load_court_template("ncaa_nc4_640x360", 4, B)[:, :, :90, :112].contiguous()
    poi = synth.load_court_poi("pitch", B)
    net = Reconstructor(court.cuda(), poi.cuda(), target_size=(112, 90), unet_size=(112, 90), warp_size=(112, 90),
                        warp_with_nearest=True, resnet_name="resnet50")
    sd = synth.synth_state_dict(net.state_dict(), 31)
    net.load_state_dict(sd)
    net.cuda().eval()
    x = synth.smooth_frames(B, 90, 112, seed=31)
    with torch.no_grad():
        out = net.predict(x.cuda(), consistency=False)
        logits, _, _ = torch_ref.forward_unet(x, sd, (112, 90), (112, 90))
        theta = torch_ref.resnet_stn(torch.cat((logits, x), 1), sd)
    assert _maxerr(out["theta"].cpu(), theta) < 1e-4
    wm = (warp_ref.homography_warp(out["theta"].cpu(), court, 90, 112, "nearest") * 4)
    assert torch.equal(out["warp_mask"].cpu(), wm.to(torch.int32))


def _rescaled_checkpoint(sd, factor, pairs):
    """The same function with larger intermediate activations: BatchNorm affine of layer a times `factor`, the conv
    that reads it divided by `factor` (ReLU and max-pool are positively homogeneous)."""
    sd2 = {k: v.clone() for k, v in sd.items()}
    for bn, convs in pairs:
        sd2[bn + ".weight"] *= factor
        sd2[bn + ".bias"] *= factor
        for c in convs:
            if isinstance(c, tuple):    # (key, slice of input channels)
                sd2[c[0]][:, c[1]] /= factor
            else:
                sd2[c] /= factor
    return sd2


def test_f16x3_frame_beyond_the_fp16_range_lowers_the_frame_exponent(E):
    """Round 4: the input frame is an H2 tensor too (the first layer runs on the fp16 matrix cores from sfh_frame_to_h2's
    two-plane copy).  Frames scaled by 2^17 (values up to 131072 against the +-16376 of the default exponent) with the first
    conv's weights divided by it - the same function: the range word of "frame" reports it, its exponent goes down, the
    pass repeats from the first launch and stays on the two-plane path; results equal the unscaled run (itself checked against
    the CPU restatement elsewhere) to rounding; the next batch repeats nothing."""
    from sfh_amd.reconstructor import Reconstructor
    B, H, W = 2, 48, 64
    court = synth.load_court_template("ncaa_nc4_640x360", 4, B)[:, :, :H, :W].contiguous()
    poi = synth.load_court_poi("pitch", B)
    net = Reconstructor(court.cuda(), poi.cuda(), target_size=(W, H), unet_size=(W, H), warp_size=(W, H),
                        warp_with_nearest=True, resnet_input="mask")      # (the STN input holds no frame channels)
    sd = synth.synth_state_dict(net.state_dict(), 44)
    x = synth.smooth_frames(B, H, W, seed=44)
    net.load_state_dict(sd)
    net.cuda().eval()
    net.precision = "f16x3"
    with torch.no_grad():
        base = net.predict(x.cuda(), consistency=False)
    # (these smooth frames peak below 1.0 = below 4.0 in stored units: the two-sided guard of round 5 may have raised "frame")
    assert net.range_rescales == 0 and net._h2_ranges.exp("frame") >= 2
    f = 2.0 ** 17
    sd2 = dict(sd)
    sd2["inc.double_conv.0.weight"] = sd["inc.double_conv.0.weight"] / f
    net.load_state_dict(sd2)
    xs = (x * f).contiguous()
    with torch.no_grad(), warnings.catch_warnings():
        warnings.simplefilter("error")
        got = net.predict(xs.cuda(), consistency=False)
    n1 = net.range_rescales
    assert net.range_fallbacks == 0 and n1 >= 1 and net._h2_ranges.exp("frame") <= 2 - 4
    # (the unscaled run is the yardstick: the scaling by a power of two changes no product, only the frame's exponent)
    assert _maxerr(got["logits"].cpu(), base["logits"].cpu()) < 5e-4 and _maxerr(got["theta"].cpu(), base["theta"].cpu()) < 1e-4
    with torch.no_grad():
        again = net.predict(xs.cuda(), consistency=False)
    assert net.range_rescales == n1 and torch.equal(again["logits"], got["logits"])
    assert net.h2_headroom()["frame"] >= 1.0


def test_f16x3_range_guard_rescales_and_resumes(E):
    """A checkpoint whose activations leave the fp16 range of the default H2 exponent (here: BatchNorms scaled by
    2^16 / 2^8 and the convs that read them divided by it - the same function) must neither give saturated results
    nor leave the two-plane path: the kernels record the magnitudes, the saturated tensors get a smaller exponent
    and the pass resumes at the first of them; afterwards the model runs such batches with no extra work."""
    from sfh_amd.reconstructor import Reconstructor
    B, H, W = 2, 48, 64
    court = synth.load_court_template("ncaa_nc4_640x360", 4, B)[:, :, :H, :W].contiguous()
    poi = synth.load_court_poi("pitch", B)
    net = Reconstructor(court.cuda(), poi.cuda(), target_size=(W, H), unet_size=(W, H), warp_size=(W, H),
                        warp_with_nearest=True)
    sd = synth.synth_state_dict(net.state_dict(), 43)
    x = synth.smooth_frames(B, H, W, seed=43).cuda()
    net.load_state_dict(sd)
    net.cuda().eval()
    net.precision = "f16x3"
    with torch.no_grad():
        base = net.predict(x, consistency=False)
    assert net.range_fallbacks == 0 and net.range_rescales == 0     # the ordinary checkpoint stays inside the range
    sd2 = _rescaled_checkpoint(sd, 65536.0, [("inc.double_conv.1", ["inc.double_conv.3.weight"])])
    sd2 = _rescaled_checkpoint(sd2, 256.0, [
        # inc.out feeds down1 (through the pool) and, as the skip, the first 64 input channels of up4's conv
        ("inc.double_conv.4", ["down1.maxpool_conv.1.double_conv.0.weight", ("up4.conv.double_conv.0.weight", slice(0, 64))]),
        ("down3.maxpool_conv.1.double_conv.1", ["down3.maxpool_conv.1.double_conv.3.weight"]),
        ("up2.conv.double_conv.1", ["up2.conv.double_conv.3.weight"]),
        ("resnet_reg.layer2.1.bn1", ["resnet_reg.layer2.1.conv2.weight"])])
    net.load_state_dict(sd2)
    with torch.no_grad(), warnings.catch_warnings():
        warnings.simplefilter("error")
        got = net.predict(x, consistency=False)
    n1 = net.range_rescales
    assert net.range_fallbacks == 0 and n1 >= 1
    xc = x.cpu()
    want = torch_ref.predict(xc, sd2, court, poi, warp_size=(W, H), unet_size=(W, H), target_size=(W, H), consistency=False)
    assert _maxerr(got["theta"].cpu(), want["theta"]) < 1e-4
    assert _maxerr(got["logits"].cpu(), want["logits"]) < 5e-4
    assert _maxerr(got["logits"].cpu(), base["logits"].cpu()) < 2e-3    # same function up to rounding
    ex = net._h2_ranges.exps
    # the scaled tensors got smaller exponents (how much smaller depends on the headroom they had)
    assert ex["inc.mid"] <= 2 - 6 and all(e < 2 for e in ex.values())
    assert len(ex) <= 8, ex       # (a rescaled tensor's consumer may follow it down; the rest of the net keeps exponent 2)
    assert min(net.h2_headroom().values()) >= 1.0
    # sticky: the next batches - predict(), forward() and forward_unet() alike - repeat nothing and give the same bits
    with torch.no_grad():
        again = net.predict(x, consistency=False)
        lg, _, _ = net.forward_unet(x)
        fw = net(x)
    assert net.range_rescales == n1 and net.range_fallbacks == 0
    assert torch.equal(again["logits"], got["logits"]) and torch.equal(again["theta"], got["theta"])
    assert torch.equal(lg, got["logits"]) and torch.equal(fw["theta"], got["theta"])
    # new weights keep the exponents (same model): loading the checkpoint again costs no second calibration
    net.load_state_dict(sd2)
    with torch.no_grad():
        third = net.predict(x, consistency=False)
    assert net.range_rescales == n1 and torch.equal(third["theta"], got["theta"])
    # a caller that pipelines batches switches the guard off and asks afterwards
    net2 = Reconstructor(court.cuda(), poi.cuda(), target_size=(W, H), unet_size=(W, H), warp_size=(W, H), warp_with_nearest=True)
    net2.load_state_dict(sd2)
    net2.cuda().eval()
    net2.range_guard = False
    with torch.no_grad():
        net2.predict(x, consistency=False)
    assert net2.range_overflowed() is True and net2.range_rescales == 0


@pytest.mark.parametrize("factor", [2.0 ** 8, 2.0 ** 12])
def test_f16x3_stays_on_the_two_plane_path_with_scaled_logits_and_gammas(E, factor):
    """Larger frames, every level's first BatchNorm scaled by 2^8 / 2^12 and the logit head by 2^8 (the head for real: the
    STN then sees 256x the logits, a different function - compared with the CPU restatement of that checkpoint)."""
    from sfh_amd.reconstructor import Reconstructor
    B, H, W = 2, 90, 112
    court = synth.load_court_template("ncaa_nc4_640x360", 4, B)[:, :, :H, :W].contiguous()
    poi = synth.load_court_poi("pitch", B)
    net = Reconstructor(court.cuda(), poi.cuda(), target_size=(W, H), unet_size=(W, H), warp_size=(W, H), warp_with_nearest=True)
    sd = synth.synth_state_dict(net.state_dict(), 61)
    pairs = [(f"down{i}.maxpool_conv.1.double_conv.1", [f"down{i}.maxpool_conv.1.double_conv.3.weight"]) for i in (1, 2, 3, 4)]
    pairs += [(f"up{i}.conv.double_conv.1", [f"up{i}.conv.double_conv.3.weight"]) for i in (1, 2, 3, 4)]
    pairs += [("resnet_reg.layer1.0.bn1", ["resnet_reg.layer1.0.conv2.weight"]), ("resnet_reg.layer4.2.bn1", ["resnet_reg.layer4.2.conv2.weight"])]
    sd2 = _rescaled_checkpoint(sd, factor, pairs)
    head = min(factor, 256.0)      # (beyond 2^8 the STN's theta leaves every sensible range and its inverse for the POI is ill-conditioned)
    sd2["outc.conv.weight"] *= head
    sd2["outc.conv.bias"] *= head
    net.load_state_dict(sd2)
    net.cuda().eval()
    x = synth.smooth_frames(B, H, W, seed=61)
    with torch.no_grad():
        got = net.predict(x.cuda(), consistency=True, project_poi=True)
        want = torch_ref.predict(x, sd2, court, poi, warp_size=(W, H), unet_size=(W, H), target_size=(W, H), project_poi=True)
    assert net.range_fallbacks == 0 and (net.range_rescales >= 1 or factor < 1000)
    assert _maxerr(got["theta"].cpu(), want["theta"]) < 1e-4
    assert _maxerr(got["logits"].cpu(), want["logits"]) < 5e-4 * head
    assert _maxerr(got["poi"].cpu(), want["poi"]) < 5e-4      # (theta is 256x as sensitive to the logits as usual here)
    wm = (warp_ref.homography_warp(got["theta"].cpu(), court, H, W, "nearest") * 4).to(torch.int32)
    assert torch.equal(got["warp_mask"].cpu(), wm)
    n1 = net.range_rescales
    with torch.no_grad():
        net.predict(x.cuda(), consistency=True, project_poi=True)
    assert net.range_rescales == n1 and net.range_fallbacks == 0


@pytest.mark.parametrize("factor", [2.0 ** -8, 2.0 ** -14])
def test_f16x3_quiet_layers_get_larger_exponents(E, factor):
    """Round 5: the range guard is two-sided.  Ten BatchNorms scaled DOWN by 2^8 / 2^14 and the convs that read them scaled
    up by it (the same function): at the default exponent those tensors peak at 2^-6 .. 2^-12 of the fp16 range, their low
    planes are subnormal and a K = 576 .. 9216 dot product inherits 1e-5 .. 1e-4 of relative error - nothing saturates,
    so the one-sided guard of rounds 3-4 never noticed.  Now the kernels' range words show the quiet tensors, their
    exponents go UP (peak -> [2^12, 2^13)), the pass resumes from the first of them, and the result meets the usual
    bounds against the CPU restatement of that checkpoint.  The run without the raise is measured beside it."""
    from sfh_amd.reconstructor import Reconstructor
    B, H, W = 2, 90, 112
    court = synth.load_court_template("ncaa_nc4_640x360", 4, B)[:, :, :H, :W].contiguous()
    poi = synth.load_court_poi("pitch", B)
    net = Reconstructor(court.cuda(), poi.cuda(), target_size=(W, H), unet_size=(W, H), warp_size=(W, H), warp_with_nearest=True)
    sd = synth.synth_state_dict(net.state_dict(), 61)
    pairs = [(f"down{i}.maxpool_conv.1.double_conv.1", [f"down{i}.maxpool_conv.1.double_conv.3.weight"]) for i in (1, 2, 3, 4)]
    pairs += [(f"up{i}.conv.double_conv.1", [f"up{i}.conv.double_conv.3.weight"]) for i in (1, 2, 3, 4)]
    pairs += [("resnet_reg.layer1.0.bn1", ["resnet_reg.layer1.0.conv2.weight"]), ("resnet_reg.layer4.2.bn1", ["resnet_reg.layer4.2.conv2.weight"])]
    sd2 = _rescaled_checkpoint(sd, factor, pairs)
    net.load_state_dict(sd2)
    net.cuda().eval()
    x = synth.smooth_frames(B, H, W, seed=61)
    want = torch_ref.predict(x, sd2, court, poi, warp_size=(W, H), unet_size=(W, H), target_size=(W, H), project_poi=True)
    # (a) the one-sided guard of round 4: nothing saturates, nothing happens
    with torch.no_grad():
        net._get_engines()
        net._h2_ranges.RAISE_BELOW = 0.0
        one_sided = net.predict(x.cuda(), consistency=True, project_poi=True)
    assert net.range_raises == 0 and net.range_rescales == 0
    err1 = (_maxerr(one_sided["theta"].cpu(), want["theta"]), _maxerr(one_sided["logits"].cpu(), want["logits"]))
    del net._h2_ranges.RAISE_BELOW           # back to the class default
    # (b) two-sided
    with torch.no_grad(), warnings.catch_warnings():
        warnings.simplefilter("error")
        got = net.predict(x.cuda(), consistency=True, project_poi=True)
    n1 = net.range_raises
    assert n1 >= 1 and net.range_fallbacks == 0 and net.range_rescales == 0
    ex = net._h2_ranges.exps
    want_e = 13 - math.frexp(factor)[1]      # a tensor of O(1..100) / factor lands about here
    scaled = [f"down{i}.mid" for i in (1, 2, 3, 4)] + [f"up{i}.conv.mid" for i in (1, 2, 3, 4)]
    assert all(want_e - 9 <= ex.get(k, 2) <= want_e + 2 for k in scaled), {k: ex.get(k, 2) for k in scaled}
    assert all(k in ex for k in ("rn.layer1.0.t", "rn.layer4.2.t"))
    err2 = (_maxerr(got["theta"].cpu(), want["theta"]), _maxerr(got["logits"].cpu(), want["logits"]))
    assert err2[0] < 1e-4 and err2[1] < 5e-4, err2
    assert _maxerr(got["poi"].cpu(), want["poi"]) < 1e-4
    wm = (warp_ref.homography_warp(got["theta"].cpu(), court, H, W, "nearest") * 4).to(torch.int32)
    assert torch.equal(got["warp_mask"].cpu(), wm)
    assert min(net.h2_headroom().values()) >= 1.0
    # every raised tensor now peaks in [2^12, 2^13) in stored units
    bits = net._h2_ranges.read()
    for k in scaled:
        u = E._bits_to_float(bits[k])
        assert 2.0 ** 12 <= u < 2.0 ** 13, (k, u)
    # sticky: the next batch repeats nothing and gives the same bits; the pipelined entry point as well
    with torch.no_grad():
        again = net.predict(x.cuda(), consistency=True, project_poi=True)
        piped = net.predict_async(x.cuda(), consistency=True, project_poi=True).result()
    assert net.range_raises == n1 and all(torch.equal(again[k], got[k]) and torch.equal(piped[k], got[k]) for k in got)
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    rec = {"case": "quiet_layers", "factor_log2": math.log2(factor), "one_sided_dtheta": err1[0], "one_sided_dlogits": err1[1],
           "two_sided_dtheta": err2[0], "two_sided_dlogits": err2[1], "raises": n1, "exps": {k: ex.get(k, 2) for k in scaled}}
    print(json.dumps(rec))
    if os.path.isdir(out):
        with open(os.path.join(out, "r06_parity_full_size.jsonl"), "a") as f:
            f.write(json.dumps(rec) + "\n")


def test_f16x3_quiet_tensor_in_a_pipelined_batch_is_raised_and_recomputed(E):
    """predict_async(): the range check in result() sees a quiet tensor, drains, raises the exponent and recomputes the
    batches in flight - their outputs equal predict()'s of a model that calibrated synchronously."""
    from sfh_amd.reconstructor import Reconstructor
    B, H, W = 2, 48, 64
    court = synth.load_court_template("ncaa_nc4_640x360", 4, B)[:, :, :H, :W].contiguous()
    poi = synth.load_court_poi("pitch", B)
    nets = []
    sd2 = None
    for _ in range(2):
        net = Reconstructor(court.cuda(), poi.cuda(), target_size=(W, H), unet_size=(W, H), warp_size=(W, H), warp_with_nearest=True)
        if sd2 is None:
            sd = synth.synth_state_dict(net.state_dict(), 71)
            sd2 = _rescaled_checkpoint(sd, 2.0 ** -12, [("down2.maxpool_conv.1.double_conv.1", ["down2.maxpool_conv.1.double_conv.3.weight"])])
        net.load_state_dict(sd2)
        nets.append(net.cuda().eval())
    ref, pipe = nets
    xs = [synth.smooth_frames(B, H, W, seed=100 + k).cuda() for k in range(3)]
    with torch.no_grad():
        want = [ref.predict(x, consistency=True, project_poi=True) for x in xs]
        want[0] = ref.predict(xs[0], consistency=True, project_poi=True)      # (with the exponents the three batches settled on)
        hs = [pipe.predict_async(xs[0], consistency=True, project_poi=True), pipe.predict_async(xs[1], consistency=True, project_poi=True)]
        got = [hs[0].result()]
        hs.append(pipe.predict_async(xs[2], consistency=True, project_poi=True))
        got += [hs[1].result(), hs[2].result()]
    assert pipe.range_raises >= 1 and ref.range_raises >= 1 and pipe.range_fallbacks == 0
    assert pipe._h2_ranges.exps == ref._h2_ranges.exps
    for g, w in zip(got, want):
        for key in w:
            assert torch.equal(g[key], w[key]), key


def test_predict_async_pipeline_gives_predict_s_bits(E):
    """predict_async(): the ResNet-STN / warp / CE / POI of batch k run on a side stream under the UNet of batch k + 1.
    Five different batches through the pipeline (two in flight) give exactly predict()'s outputs; a batch that
    saturates a tensor is noticed in its result(), the pipeline drains, the exponent goes down and the batches in flight
    are recomputed - still equal to predict(); a synchronous predict() between pipelined batches is safe."""
    from sfh_amd.reconstructor import Reconstructor
    B, H, W = 2, 48, 64
    court = synth.load_court_template("ncaa_nc4_640x360", 4, B)[:, :, :H, :W].contiguous()
    poi = synth.load_court_poi("pitch", B)
    sd = None
    nets = []
    for _ in range(2):
        net = Reconstructor(court.cuda(), poi.cuda(), target_size=(W, H), unet_size=(W, H), warp_size=(W, H), warp_with_nearest=True)
        sd = sd or synth.synth_state_dict(net.state_dict(), 71)
        net.load_state_dict(sd)
        nets.append(net.cuda().eval())
    ref, pipe = nets
    xs = [synth.smooth_frames(B, H, W, seed=100 + k).cuda() for k in range(5)]
    with torch.no_grad():
        want = [ref.predict(x, consistency=True, project_poi=True) for x in xs]
        got, prev = [], None
        for k, x in enumerate(xs):
            h = pipe.predict_async(x, consistency=True, project_poi=True)
            if prev is not None:
                got.append(prev.result())
            if k == 2:      # a synchronous call in the middle of the stream of batches
                mid = pipe.predict(xs[0], consistency=True, project_poi=True)
                assert torch.equal(mid["theta"], want[0]["theta"])
            prev = h
        got.append(prev.result())
    torch.cuda.synchronize()
    for g, w in zip(got, want):
        assert sorted(g) == sorted(w)
        for key in w:
            assert torch.equal(g[key], w[key]), key
    assert pipe.range_rescales == 0
    # a checkpoint that saturates inc.mid at the default exponent: noticed in result(), recomputed, same answer as predict()
    sd2 = _rescaled_checkpoint(sd, 65536.0, [("inc.double_conv.1", ["inc.double_conv.3.weight"])])
    for n in nets:
        n.load_state_dict(sd2)
    with torch.no_grad():
        want = [ref.predict(x, consistency=True, project_poi=True) for x in xs[:3]]
        hs = [pipe.predict_async(xs[0], consistency=True, project_poi=True), pipe.predict_async(xs[1], consistency=True, project_poi=True)]
        got = [hs[0].result()]
        hs.append(pipe.predict_async(xs[2], consistency=True, project_poi=True))
        got += [hs[1].result(), hs[2].result()]
    assert pipe.range_rescales >= 1 and pipe.range_fallbacks == 0
    for g, w in zip(got, want):
        for key in w:
            assert torch.equal(g[key], w[key]), key


def test_forward_and_forward_unet_between_pipelined_batches(E):
    """Round 4 (review finding): forward() and forward_unet() write the STN-input buffer and the ResNet workspace that a
    predict_async() batch in flight may still be reading on the side stream - they now wait for those reads like predict()
    does.  Interleaved with pipelined batches they give their own sequential results, and the batches theirs; and a
    synchronous call that finds a SATURATED tensor (it zeroes the shared range words) invalidates the batches in flight, whose
    result() then recomputes them instead of handing out clamped outputs."""
    from sfh_amd.reconstructor import Reconstructor
    B, H, W = 2, 48, 64
    court = synth.load_court_template("ncaa_nc4_640x360", 4, B)[:, :, :H, :W].contiguous()
    poi = synth.load_court_poi("pitch", B)
    nets = []
    sd = None
    for _ in range(2):
        net = Reconstructor(court.cuda(), poi.cuda(), target_size=(W, H), unet_size=(W, H), warp_size=(W, H), warp_with_nearest=True)
        sd = sd or synth.synth_state_dict(net.state_dict(), 72)
        net.load_state_dict(sd)
        nets.append(net.cuda().eval())
    ref, pipe = nets
    xs = [synth.smooth_frames(B, H, W, seed=200 + k).cuda() for k in range(4)]
    with torch.no_grad():
        want_p = [ref.predict(x, consistency=True, project_poi=True) for x in xs]
        want_f = ref(xs[3])
        want_u = ref.forward_unet(xs[2])[0]
        h0 = pipe.predict_async(xs[0], consistency=True, project_poi=True)
        fw = pipe(xs[3])                                  # forward() while batch 0's ResNet may still run
        h1 = pipe.predict_async(xs[1], consistency=True, project_poi=True)
        lg = pipe.forward_unet(xs[2])[0]
        r0, r1 = h0.result(), h1.result()
    torch.cuda.synchronize()
    for got, want in ((r0, want_p[0]), (r1, want_p[1]), (fw, want_f)):
        assert sorted(got) == sorted(want)
        for key in want:
            assert torch.equal(got[key], want[key]), key
    assert torch.equal(lg, want_u)
    # a saturating checkpoint: the synchronous predict() in the middle lowers the exponent and zeroes the words; the batch
    # in flight was computed with the old exponent (clamped values) - its result() must not return that
    sd2 = _rescaled_checkpoint(sd, 65536.0, [("inc.double_conv.1", ["inc.double_conv.3.weight"])])
    for n in nets:
        n.load_state_dict(sd2)
    with torch.no_grad():
        want = [ref.predict(x, consistency=True, project_poi=True) for x in xs[:2]]
        h0 = pipe.predict_async(xs[0], consistency=True, project_poi=True)       # saturates inc.mid, not checked yet
        mid = pipe.predict(xs[1], consistency=True, project_poi=True)            # finds the word, rescales, resets the words
        r0 = h0.result()
    assert pipe.range_rescales >= 1 and pipe.range_fallbacks == 0
    for got, w in ((mid, want[1]), (r0, want[0])):
        for key in w:
            assert torch.equal(got[key], w[key]), key


@pytest.mark.parametrize("B,size", [(1, (50, 70)), (3, (33, 47)), (1, (16, 16)), (5, (64, 48))])
def test_ragged_batches_and_sizes_vs_oracle(E, B, size):
    """Batch sizes 1 / 3 / 5, odd and minimal frame sizes (16x16 is the smallest frame four 2x2 poolings allow),
    default arithmetic, against the CPU restatement; and a batch of zero frames gives empty tensors like torch."""
    from sfh_amd.reconstructor import Reconstructor
    H, W = size
    court = synth.load_court_template("ncaa_nc4_640x360", 4, B)[:, :, :H, :W].contiguous()
    poi = synth.load_court_poi("pitch", B)
    net = Reconstructor(court.cuda(), poi.cuda(), target_size=(W, H), unet_size=(W, H), warp_size=(W, H),
                        warp_with_nearest=True)
    sd = synth.synth_state_dict(net.state_dict(), 53)
    net.load_state_dict(sd)
    net.cuda().eval()
    x = synth.smooth_frames(B, H, W, seed=53)
    with torch.no_grad():
        out = net.predict(x.cuda(), consistency=True, project_poi=True)
        want = torch_ref.predict(x, sd, court, poi, warp_size=(W, H), unet_size=(W, H), target_size=(W, H), project_poi=True)
    assert _maxerr(out["theta"].cpu(), want["theta"]) < 1e-4
    assert _maxerr(out["logits"].cpu(), want["logits"]) < 5e-4
    assert _maxerr(out["poi"].cpu(), want["poi"]) < 1e-4
    wm = (warp_ref.homography_warp(out["theta"].cpu(), court, H, W, "nearest") * 4).to(torch.int32)
    assert torch.equal(out["warp_mask"].cpu(), wm)
    with torch.no_grad():
        e = net.predict(x[:0].cuda(), consistency=True, project_poi=True)
        f = net(x[:0].cuda())
    assert tuple(e["logits"].shape) == (0, 4, H, W) and tuple(e["theta"].shape) == (0, 1, 3, 3)
    assert tuple(e["warp_mask"].shape) == (0, H, W) and e["warp_mask"].dtype == torch.int32
    assert tuple(e["consist_score"].shape) == (0,) and tuple(e["poi"].shape) == (0, poi.shape[1], 2)
    assert tuple(f["logits"].shape) == (0, 4, H, W) and f["warp_mask"].dtype == torch.float32


@pytest.mark.parametrize("precision", ["f16x3", "bf16x6", "fp32"])
def test_non_finite_frame_gives_non_finite_outputs_like_torch(E, precision):
    """torch propagates a NaN through conv / BatchNorm / ReLU / max-pool, so the reference answers a frame that
    holds one with NaN logits and a NaN theta; the HIP path must not launder it into finite numbers (ReLU and the
    pooling maxima keep NaNs; the H2 conversions flag them and the batch is repeated in bf16x6).  The other frame of
    the batch is untouched."""
    from sfh_amd.reconstructor import Reconstructor
    B, H, W = 2, 48, 64
    court = synth.load_court_template("ncaa_nc4_640x360", 4, B)[:, :, :H, :W].contiguous()
    poi = synth.load_court_poi("pitch", B)
    net = Reconstructor(court.cuda(), poi.cuda(), target_size=(W, H), unet_size=(W, H), warp_size=(W, H),
                        warp_with_nearest=True)
    sd = synth.synth_state_dict(net.state_dict(), 47)
    net.load_state_dict(sd)
    net.cuda().eval()
    net.precision = precision
    x = synth.smooth_frames(B, H, W, seed=47)
    with torch.no_grad():
        clean = net.predict(x.cuda(), consistency=False)
    xn = x.clone()
    xn[1, 2, 20, 30] = float("nan")
    with torch.no_grad(), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        got = net.predict(xn.cuda(), consistency=False)
        want = torch_ref.predict(xn, sd, court, poi, warp_size=(W, H), unet_size=(W, H), target_size=(W, H))
    assert torch.isnan(want["theta"][1]).all() and torch.isnan(got["theta"][1].cpu()).all()
    assert torch.isnan(got["logits"][1]).any()
    # where the reference's logits are NaN ours are too (the receptive field of the bad pixel)
    assert bool((torch.isnan(got["logits"][1].cpu()) | ~torch.isnan(want["logits"][1])).all())
    assert not torch.isnan(got["theta"][0]).any() and not torch.isnan(got["logits"][0]).any()
    assert _maxerr(got["theta"][0].cpu(), clean["theta"][0].cpu()) < 1e-5
    if precision == "f16x3":
        assert net.range_fallbacks == 1


@pytest.mark.parametrize("precision", ["f16x3", "bf16x6"])
@pytest.mark.parametrize("size", [(96, 128), (90, 112), (54, 72)])
def test_fused_up_block_vs_unfused_and_oracle(E, monkeypatch, size, precision):
    """Up blocks without F.pad run as skip-half conv + composed 2x2 quadrant conv over the low-resolution
    tensor (ConvTranspose2d folded into the consumer conv); with SFH_FUSE_UP=0 as ConvTranspose2d + conv over
    the concatenation.  Both against the oracle, and against each other, incl. the image borders."""
    from sfh_amd.reconstructor import Reconstructor
    # 96x128: no level needs F.pad; 90x112: 5->10 vs 11 and 22->44 vs 45 (one padded row); 54x72: 27 rows, 9 cols
    B, (H, W) = 2, size
    court = synth.load_court_template("ncaa_nc4_640x360", 4, B)[:, :, :H, :W].contiguous()
    poi = synth.load_court_poi("pitch", B)
    x = synth.smooth_frames(B, H, W, seed=37)
    outs = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("SFH_FUSE_UP", flag)
        net = Reconstructor(court.cuda(), poi.cuda(), target_size=(W, H), unet_size=(W, H), warp_size=(W, H),
                            warp_with_nearest=True)
        sd = synth.synth_state_dict(net.state_dict(), 37)
        net.load_state_dict(sd)
        net.cuda().eval()
        net.precision = precision
        with torch.no_grad():
            outs[flag] = net.predict(x.cuda(), consistency=False)
        un, _ = net._get_engines()
        assert ("up4.fused" in un.L) == (flag == "1")
    with torch.no_grad():
        logits, _, _ = torch_ref.forward_unet(x, sd, (W, H), (W, H))
    for flag in ("1", "0"):
        assert _maxerr(outs[flag]["logits"].cpu(), logits) < 3e-4, flag
    d = (outs["1"]["logits"] - outs["0"]["logits"]).abs()
    assert d.max().item() < 1e-4
    # borders are where the transposed conv's bias reaches the conv through fewer taps
    assert max(d[:, :, 0].max().item(), d[:, :, -1].max().item(), d[:, :, :, 0].max().item(), d[:, :, :, -1].max().item()) < 1e-4
    assert _maxerr(outs["1"]["theta"].cpu(), outs["0"]["theta"].cpu()) < 1e-5


@pytest.mark.parametrize("precision", ["f16x3", "bf16x6"])
@pytest.mark.parametrize("size", [(90, 112), (48, 64)])
def test_fused_outconv_head_vs_outconv_kernel(E, monkeypatch, size, precision):
    """OutConv + cat((logits, x)) in the epilogue of the last 3x3 conv against the separate OutConv kernel."""
    from sfh_amd.reconstructor import Reconstructor
    B, (H, W) = 2, size
    court = synth.load_court_template("ncaa_nc4_640x360", 4, B)[:, :, :H, :W].contiguous()
    poi = synth.load_court_poi("pitch", B)
    x = synth.smooth_frames(B, H, W, seed=41)
    res = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("SFH_FUSE_HEAD", flag)
        net = Reconstructor(court.cuda(), poi.cuda(), target_size=(W, H), unet_size=(W, H), warp_size=(W, H))
        sd = synth.synth_state_dict(net.state_dict(), 41)
        net.load_state_dict(sd)
        net.cuda().eval()
        net.precision = precision
        un, _ = net._get_engines()
        with torch.no_grad():
            r = un.run(x.cuda(), want_stn_in=True)
            torch.cuda.synchronize()
        assert ("y4" in r) == (flag == "0")          # the 64-channel tensor is not stored when the head is fused
        res[flag] = (r["logits"].clone(), r["stn_in"].clone())
    assert _maxerr(res["1"][0].cpu(), res["0"][0].cpu()) < 2e-5
    assert _maxerr(res["1"][1].cpu(), res["0"][1].cpu()) < 2e-5
    assert torch.equal(res["1"][1][..., 4:7], res["0"][1][..., 4:7]) and float(res["1"][1][..., 7].abs().max()) == 0.0
    with torch.no_grad():
        logits, _, _ = torch_ref.forward_unet(x, sd, (W, H), (W, H))
    assert _maxerr(res["1"][0].cpu(), logits) < 3e-4


@pytest.mark.parametrize("fmt", ["s3", "h2"])
@pytest.mark.parametrize("shape", [(3, 45, 83), (1, 16, 16), (2, 90, 112), (2, 23, 70)])
@pytest.mark.parametrize("cin", [7, 5, 8])
def test_stem_kernel_vs_torch(E, shape, cin, fmt):
    """7x7 stride-2 pad-3 conv + BatchNorm(eval) + ReLU (models/resnet.py:172,241-243) on the tap-packed
    split-bf16 kernel: odd sizes, partial tiles, fewer than 8 real channels."""
    B, H, W = shape
    g = torch.Generator().manual_seed(B * 1000 + H + cin)
    conv = torch.nn.Conv2d(cin, 64, 7, stride=2, padding=3, bias=False)
    bn = torch.nn.BatchNorm2d(64)
    with torch.no_grad():
        conv.weight.normal_(0, 0.1, generator=g)
        bn.weight.uniform_(0.5, 1.5, generator=g); bn.bias.uniform_(-0.3, 0.3, generator=g)
        bn.running_mean.uniform_(-0.2, 0.2, generator=g); bn.running_var.uniform_(0.5, 1.5, generator=g)
    bn.eval()
    x = torch.randn(B, cin, H, W, generator=g)
    with torch.no_grad():
        want = torch.relu(bn(conv(x.double().float()))).double()
        ref64 = torch.relu(torch.nn.functional.batch_norm(
            torch.nn.functional.conv2d(x.double(), conv.weight.double(), None, 2, 3), bn.running_mean.double(),
            bn.running_var.double(), bn.weight.double(), bn.bias.double(), False, 0.0, bn.eps))
    conv.cuda(); bn.cuda()
    st = E.StemConv(conv, bn, cin, fmt=fmt)
    xin = E.nchw_to_nhwc(x.cuda(), 8)
    ho, wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    out = torch.empty((B, ho, wo, 64), device="cuda")
    st.run(xin, B, H, W, out)
    torch.cuda.synchronize()
    got = out.permute(0, 3, 1, 2).cpu().double()
    assert (got - ref64).abs().max().item() < 3e-5 * max(1.0, ref64.abs().max().item())
    assert (got - ref64).abs().max().item() < 4 * (want - ref64).abs().max().item() + 1e-6   # fp32-grade accuracy


def test_u8_frame_preprocessing_matches_dataset(E):
    fr = synth.synth_frames_u8(3, 45, 80, seed=7)
    want = torch.from_numpy((fr.transpose(0, 3, 1, 2) / 255)).type(torch.FloatTensor)   # utils/dataset.py:154-159
    got = E.frames_u8_to_input(torch.from_numpy(fr).cuda())
    torch.cuda.synchronize()
    assert torch.equal(got.cpu(), want)
    assert torch.equal(synth.frames_to_float(fr), want)


def test_u8_frame_area2_downscale(E):
    """1280x720-style frames -> half size with OpenCV's 2x2 INTER_AREA rule, then /255 (utils/dataset.py:310-330)."""
    fr = synth.synth_frames_u8(2, 44, 72, seed=9)
    f = fr.astype(np.int32)
    area = ((f[:, 0::2, 0::2] + f[:, 0::2, 1::2] + f[:, 1::2, 0::2] + f[:, 1::2, 1::2] + 2) >> 2).astype(np.uint8)
    want = torch.from_numpy((area.transpose(0, 3, 1, 2) / 255)).type(torch.FloatTensor)
    got = E.frames_u8_to_input(torch.from_numpy(fr).cuda(), target_size=(36, 22))
    torch.cuda.synchronize()
    assert torch.equal(got.cpu(), want)
    with pytest.raises(NotImplementedError):      # frames narrower than the target: the reference switches to INTER_LINEAR
        E.frames_u8_to_input(torch.from_numpy(fr).cuda(), target_size=(80, 50))
    # unequal integer factors (3 x 2) and factors beyond 16: OpenCV's fast path with a kx x ky block
    from oracle import post_ref
    for (tw, th) in ((24, 22), (36, 11), (4, 2)):
        kx, ky = 72 // tw, 44 // th
        area = np.stack([post_ref.resize_area_int(f_, kx, ky) for f_ in fr])
        want = torch.from_numpy((area.transpose(0, 3, 1, 2) / 255)).type(torch.FloatTensor)
        assert torch.equal(E.frames_u8_to_input(torch.from_numpy(fr).cuda(), target_size=(tw, th)).cpu(), want), (tw, th)


@pytest.mark.parametrize("src,dst", [((1080, 1920), (576, 1024)), ((900, 1600), (360, 640)), ((50, 77), (21, 30)), ((45, 64), (44, 63)),
                                     ((720, 1280), (480, 854))])
def test_u8_frame_non_integer_area_downscale(E, src, dst):
    """Round 5: any downscale, not only integer factors (1920x1080 -> 1024x576, 1600x900 -> 640x360, ...): OpenCV's generic
    INTER_AREA path (computeResizeAreaTab + resizeArea_), then /255 (utils/dataset.py:310-330), bit for bit against the numpy
    restatement oracle/post_ref.resize_area; the per-axis tables of the C-ABI's host helper against the oracle's own."""
    import ctypes
    from oracle import post_ref
    from sfh_amd import _lib
    (hs, ws), (hd, wd) = src, dst
    B = 2
    fr = synth.synth_frames_u8(B, hs, ws, seed=23)
    fr[0, :8, :8] = 255
    fr[0, :8, 8:16] = 0
    area = np.stack([post_ref.resize_area(f, (wd, hd)) for f in fr])
    want = torch.from_numpy((area.transpose(0, 3, 1, 2) / 255)).type(torch.FloatTensor)
    got = E.frames_u8_to_input(torch.from_numpy(fr).cuda(), target_size=(wd, hd))
    torch.cuda.synchronize()
    assert tuple(got.shape) == (B, 3, hd, wd)
    assert torch.equal(got.cpu(), want)
    # an area average stays within the range of its sources and keeps a constant image constant
    flat = np.full((1, hs, ws, 3), 137, np.uint8)
    assert torch.equal(E.frames_u8_to_input(torch.from_numpy(flat).cuda(), target_size=(wd, hd)).cpu(),
                       torch.full((1, 3, hd, wd), 137 / 255, dtype=torch.float32))
    lib = _lib.load()
    for ss, ds in ((ws, wd), (hs, hd)):
        cap = 2 * ds + ss
        ofs, si, al = np.zeros(ds + 1, np.int32), np.zeros(cap, np.int32), np.zeros(cap, np.float32)
        n = lib.sfh_resize_area_tab(ss, ds, ofs.ctypes.data_as(ctypes.c_void_p), si.ctypes.data_as(ctypes.c_void_p),
                                    al.ctypes.data_as(ctypes.c_void_p), cap)
        tab = post_ref._area_tab(ss, ds)
        assert n == len(tab)
        assert [(d, int(si[k]), float(al[k])) for d in range(ds) for k in range(ofs[d], ofs[d + 1])] == [(d, s_, float(a)) for d, s_, a in tab]


@pytest.mark.parametrize("k,hw", [(3, (360, 640)), (3, (15, 22)), (4, (11, 18)), (5, (7, 9))])
def test_u8_frame_integer_area_downscale(E, k, hw):
    """1920x1080 -> 640x360 (k = 3) and other integer factors: OpenCV's INTER_AREA block average, then /255
    (utils/dataset.py:310-330), against the numpy restatement oracle/post_ref.resize_area_int; every possible block
    sum of the 3x3 case is also checked against exact rational rounding where no tie is near."""
    from oracle import post_ref
    h, w = hw
    B = 2
    fr = synth.synth_frames_u8(B, h * k, w * k, seed=11 + k)
    fr[0, :k, :k] = 255                      # saturating block
    fr[0, :k, k:2 * k] = 0
    area = np.stack([post_ref.resize_area_int(f, k) for f in fr])
    want = torch.from_numpy((area.transpose(0, 3, 1, 2) / 255)).type(torch.FloatTensor)
    got = E.frames_u8_to_input(torch.from_numpy(fr).cuda(), target_size=(w, h))
    torch.cuda.synchronize()
    assert tuple(got.shape) == (B, 3, h, w)
    assert torch.equal(got.cpu(), want)
    assert float(got[0, :, 0, 0].min()) == 1.0 and float(got[0, :, 0, 1].max()) == 0.0
    # the restatement itself: mean of the block rounded to nearest (no block sum / k^2 of these sizes hits .5 exactly
    # unless k is even)
    if k % 2:
        exact = np.floor(fr.reshape(B, h, k, w, k, 3).astype(np.int64).sum(axis=(2, 4)) / (k * k) + 0.5).astype(np.uint8)
        assert np.array_equal(area, exact)


def test_model_api_errors(E):
    from sfh_amd.reconstructor import Reconstructor
    net, sd, court, poi = _model((112, 90))
    net.load_state_dict(sd)
    with pytest.raises(RuntimeError):
        net.eval().predict(torch.zeros(1, 3, 90, 112))          # CPU model: no fallback
    net.cuda()
    with pytest.raises(NotImplementedError):
        net.train().predict(torch.zeros(1, 3, 90, 112).cuda())   # training mode not on the HIP path
    net.eval()
    with pytest.raises(ValueError):
        net.predict(torch.zeros(3, 3, 90, 112).cuda())           # batch > template batch (2)
    with pytest.raises(NotImplementedError):
        Reconstructor(court, poi, resnet_input="bogus")
    bad = dict(sd); bad.pop("outc.conv.bias")
    with pytest.raises(RuntimeError):
        net.load_state_dict(bad)                                  # strict key check like the reference


def _spawned_predict(net, x, q):
    net.court_img, net.court_poi = net.court_img.cuda(), net.court_poi.cuda()
    with torch.no_grad():
        out = net.to("cuda").eval().predict(x.cuda(), consistency=True)
    q.put({k: v.cpu().numpy() for k, v in out.items() if k in ("theta", "consist_score", "warp_mask")})


def test_predict_in_spawned_worker_process(E):
    """predict.py runs the model in a spawned worker per device (predict.py:130,252): the pickled model
    (no kernel-side state) gives the same result there as in this process."""
    import torch.multiprocessing as mp
    net, sd, court, poi = _model((112, 90))
    net.load_state_dict(sd)
    x = synth.smooth_frames(2, 90, 112, seed=3)
    with torch.no_grad():
        here = net.cuda().eval().predict(x.cuda(), consistency=True)
    net.cpu()
    net.court_img, net.court_poi = court.cpu(), poi.cpu()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_spawned_predict, args=(net, x, q))
    p.start()
    there = q.get(timeout=300)
    p.join(timeout=60)
    assert p.exitcode == 0
    assert np.array_equal(there["theta"], here["theta"].cpu().numpy())
    assert np.array_equal(there["warp_mask"], here["warp_mask"].cpu().numpy())
    assert np.allclose(there["consist_score"], here["consist_score"].cpu().numpy(), atol=1e-6)
