"""Round-6 GPU tests of the host logic around the kernels (each cites the review item it answers)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from sfh_amd import synth  # noqa: E402


def _net(B, w=112, h=90, seed=23, **kw):
    from sfh_amd.reconstructor import Reconstructor
    court = synth.load_court_template("ncaa_nc4_640x360", 4, B)[:, :, :h, :w].contiguous()
    poi = synth.load_court_poi("pitch", B)
    kw.setdefault("warp_with_nearest", True)
    net = Reconstructor(court.cuda(), poi.cuda(), target_size=(w, h), unet_size=(w, h), warp_size=(w, h), **kw)
    sd = synth.synth_state_dict(net.state_dict(), seed)
    net.load_state_dict(sd)
    return net.cuda().eval(), sd


def _close(a, b, th=1e-4, lg=5e-4):
    assert float((a["theta"] - b["theta"]).abs().max()) < th
    assert float((a["logits"] - b["logits"]).abs().max()) < lg


def test_new_weights_with_two_batches_in_flight_finish_on_the_old_engines():
    """ADVICE r05 (reconstructor.py:_on_new_weights): batches of predict_async() in flight when the weights change are
    resolved BEFORE the engines are dropped; one whose range check fails is recomputed by the old engines (packed copies of
    the old weights), never with the new weights."""
    B, w, h = 2, 112, 90
    net, sd_a = _net(B, seed=23)
    sd_b = synth.synth_state_dict(net.state_dict(), 24)
    xs = [synth.smooth_frames(B, h, w, seed=60 + k).cuda() for k in range(3)]
    with torch.no_grad():
        ref_a = [{k: v.clone() for k, v in net.predict(x, consistency=True).items()} for x in xs[:2]]
        # make the batches in flight FAIL their range check: an activation exponent far too high saturates its tensor
        rg = net._h2_ranges
        assert rg is not None and "down2.mid" in rg.slot
        rg.exps[rg.key("down2.mid")] = 30
        h1 = net.predict_async(xs[0], consistency=True)
        h2 = net.predict_async(xs[1], consistency=True)
        net.load_state_dict(sd_b)                         # weights change with both batches in flight
        out_b = net.predict(xs[2], consistency=True)      # first call with the new stamp: drains, then builds new engines
        o1, o2 = h1.result(), h2.result()
    torch.cuda.synchronize()
    assert net.range_rescales >= 1                        # the planted saturation was seen and fixed
    _close(o1, ref_a[0])
    _close(o2, ref_a[1])                                  # old weights, not sd_b's
    fresh, _ = _net(B, seed=24)
    with torch.no_grad():
        want_b = fresh.predict(xs[2], consistency=True)
    _close(out_b, want_b)
    assert float((out_b["theta"] - ref_a[0]["theta"]).abs().max()) > 1e-3      # the two checkpoints really differ


def test_outconv_backward_filter_at_1280x720_against_fp64():
    """ADVICE r05 (csrc/train.hip outconv_bwd_kernel): a workgroup covers npix / 1024 pixels, so a thread's dW / db chain
    grows with the image (about 900 terms at 1280x720 x 16); the partial sums are promoted to fp64 every 64 pixels.  dW
    and db of sfh_outconv_bwd at 1280x720 against an fp64 contraction of the same tensors (unet/unet_parts.py:71-77)."""
    import ctypes
    from sfh_amd import _lib
    lib = _lib.load()
    B, H, W, cin, nc = 4, 720, 1280, 64, 4
    g = torch.Generator(device="cuda").manual_seed(12)
    # activations with a large common offset: the case an fp32 running sum loses (sum of ~1e3 terms of magnitude 3)
    y = (torch.rand((B, H, W, cin), device="cuda", generator=g) + 2.5).contiguous()
    dl = (torch.randn((B, nc, H, W), device="cuda", generator=g) * 1e-3 + 2e-3).contiguous()
    w = torch.randn((nc, cin), device="cuda", generator=g).contiguous()
    dy = torch.empty_like(y)
    acc_w = torch.zeros((nc, cin), dtype=torch.float64, device="cuda")
    acc_b = torch.zeros((nc,), dtype=torch.float64, device="cuda")
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    _lib.check(lib.sfh_outconv_bwd(p(y), cin, p(w), p(dl), nc, B, H, W, p(dy), p(acc_w), p(acc_b), st), "outconv_bwd")
    torch.cuda.synchronize()
    want_w = torch.zeros((nc, cin), dtype=torch.float64, device="cuda")
    for b in range(B):                                   # fp64 contraction frame by frame (memory)
        want_w += torch.einsum("khw,hwc->kc", dl[b].double(), y[b].double())
    want_b = dl.double().sum(dim=(0, 2, 3))
    rel_w = float(((acc_w - want_w).abs() / want_w.abs().clamp(min=1e-30)).max())
    rel_b = float(((acc_b - want_b).abs() / want_b.abs()).max())
    # 64-term fp32 chains of products g * x: relative error a few 1e-7 per chain, averaging down over the 57,600 chains
    assert rel_w < 2e-6 and rel_b < 2e-6, (rel_w, rel_b)
    want_dy = torch.einsum("bkhw,kc->bhwc", dl, w)
    assert float((dy - want_dy).abs().max()) < 1e-6


@pytest.mark.parametrize("precision", ["f16x3", "bf16x6"])
@pytest.mark.parametrize("B", [1, 2, 8])
def test_graph_replay_gives_predict_s_bits(B, precision):
    """predict_replay() - one HIP-graph launch per batch - returns predict()'s outputs bit for bit, batch
    after batch with different frames, as fresh tensors; new weights capture anew; `net.graph_replay = True` routes predict()."""
    w, h = 160, 96
    net, sd = _net(B, w=w, h=h, seed=29)
    net.precision = precision
    xs = [synth.smooth_frames(B, h, w, seed=80 + k).cuda() for k in range(4)]
    with torch.no_grad():
        want = [{k: v.clone() for k, v in net.predict(x, consistency=True, project_poi=True).items()} for x in xs]
        got = [net.predict_replay(x, consistency=True, project_poi=True) for x in xs]     # first call captures, the rest replay
        assert len(net.__dict__["_replay"]) == 1
        for g_, w_ in zip(got, want):
            assert sorted(g_) == sorted(w_)
            for k in w_:
                assert g_[k].dtype == w_[k].dtype and torch.equal(g_[k], w_[k]), k
        # fresh tensors: a later replay does not overwrite an earlier result
        assert got[1]["theta"].data_ptr() != got[2]["theta"].data_ptr() and torch.equal(got[1]["logits"], want[1]["logits"])
        # another output set is another capture; the first one stays valid
        o2 = net.predict_replay(xs[0], consistency=False)
        assert "consist_score" not in o2 and torch.equal(o2["theta"], want[0]["theta"])
        o3 = net.predict_replay(xs[3], consistency=True, project_poi=True)
        assert torch.equal(o3["warp_mask"], want[3]["warp_mask"])
        # new weights: the captures of the old engines are dropped, the next call is right for the new checkpoint
        net.load_state_dict(synth.synth_state_dict(net.state_dict(), 30))
        fresh = net.predict(xs[0], consistency=True, project_poi=True)
        fresh = {k: v.clone() for k, v in fresh.items()}
        again = [net.predict_replay(xs[0], consistency=True, project_poi=True) for _ in range(3)]
        for a in again:
            assert all(torch.equal(a[k], fresh[k]) for k in fresh)
        assert float((fresh["theta"] - want[0]["theta"]).abs().max()) > 1e-4
        # predict() itself goes through the graph when asked to
        net.graph_replay = True
        via = net.predict(xs[0], consistency=True, project_poi=True)
        assert all(torch.equal(via[k], fresh[k]) for k in fresh)
    torch.cuda.synchronize()


def test_graph_replay_range_event_recomputes_and_recaptures():
    """a replayed batch whose activations leave the fp16 range of the captured exponents is never returned: predict()
    recomputes it (lowering the exponent), and the next call captures under the new exponents"""
    B, w, h = 2, 160, 96
    net, sd = _net(B, w=w, h=h, seed=29)
    x = synth.smooth_frames(B, h, w, seed=90).cuda()
    with torch.no_grad():
        net.predict_replay(x)
        net.predict_replay(x)
        big = x * 40000.0                                   # the frame tensor itself leaves +-16376
        want = net.predict(big)
        want = {k: v.clone() for k, v in want.items()}
        assert net.range_rescales >= 1
        # back to ordinary frames: exponents moved -> the old capture's key no longer matches
        a = net.predict_replay(x)
        b = net.predict_replay(x)
        ref = net.predict(x)
        assert all(torch.equal(a[k], ref[k]) and torch.equal(b[k], ref[k]) for k in ref)
        # and a saturating batch through the replay path itself
        c = net.predict_replay(big * 64.0)
        d = net.predict(big * 64.0)
        assert all(torch.equal(c[k], d[k]) or (torch.isnan(c[k]).any() and torch.isnan(d[k]).any()) for k in d)
    torch.cuda.synchronize()


def test_multi_absminmax_against_torch():
    """sfh_multi_absminmax (one launch for all weight exponents of an engine / a training step): max |x| and min |x| of
    tensors of awkward sizes and alignments, NaN / Inf visible as non-finite"""
    from sfh_amd import engine as E
    g = torch.Generator(device="cuda").manual_seed(3)
    base = torch.randn(3_000_000, device="cuda", generator=g)
    sizes = [1, 2, 3, 5, 255, 256, 257, 1023, 4096, 4097, 70001, 1_000_003, 9 * 512 * 512]
    ts, off = [], 0
    for k, n in enumerate(sizes):
        if off + n + 1 > base.numel():
            off = 0
        ts.append(base[off + (k % 2):off + (k % 2) + n])          # every second one starts 4 bytes off a 16-byte boundary
        off += n + 1
    got = E.absminmax(ts)
    for t, (mx, mn) in zip(ts, got):
        assert mx == float(t.abs().max()) and mn == float(t.abs().min()), t.numel()
    bad = torch.ones(1000, device="cuda")
    bad[777] = float("nan")
    inf = torch.ones(1000, device="cuda")
    inf[3] = float("-inf")
    (m1, _), (m2, _) = E.absminmax([bad, inf])
    assert m1 != m1 and m2 == float("inf")
    assert E.absminmax([torch.zeros(17, device="cuda")]) == [(0.0, 0.0)]


def test_engine_build_helpers_against_torch():
    """csrc/hostprep.hip through the C-ABI: vec_op (scale / div / mul with a tiled operand), pitched weight slices, the
    ResNet-STN input assembly for every input mode (models/reconstructor.py:174-183,214), the template replication check, fills"""
    from sfh_amd import engine as E
    g = torch.Generator(device="cuda").manual_seed(8)
    a = torch.randn(4 * 96, device="cuda", generator=g)
    b = torch.rand(96, device="cuda", generator=g) + 0.5
    assert torch.equal(E.vec_op(a, factor=0.25), a * 0.25)
    assert torch.equal(E.vec_op(a, b, "div"), a / b.repeat(4))
    assert torch.equal(E.vec_op(a, b, "mul", factor=2.0), a * b.repeat(4) * 2.0)
    one = torch.tensor([3.0], device="cuda")
    c = a.clone()
    assert E.vec_op(c, one, "mul", out=c) is c and torch.equal(c, a * 3.0)          # in place, one-element operand
    border = torch.randn(16, 4 * 96, device="cuda", generator=g)
    assert torch.equal(E.vec_op(border, b, "div"), border / b.repeat(4))
    snap = E.snapshot(a)
    assert snap.data_ptr() != a.data_ptr() and torch.equal(snap, a)
    w = torch.randn(70, 24, 3, 3, device="cuda", generator=g)
    assert torch.equal(E.slice_in_channels(w, 0, 8), w[:, :8].contiguous())
    assert torch.equal(E.slice_in_channels(w, 8, 24), w[:, 8:].contiguous())
    B, H, W = 2, 9, 13
    lg = torch.randn(B, 4, H, W, device="cuda", generator=g)
    fr = torch.randn(B, 3, H, W, device="cuda", generator=g)
    uv = torch.randn(B, 2, H, W, device="cuda", generator=g)
    for srcs, cs in (((lg, fr, None), 8), ((None, fr, None), 4), ((lg, None, None), 4), ((lg, fr, uv), 12), ((lg, fr, uv), 16)):
        want = torch.cat([t for t in srcs if t is not None], 1).permute(0, 2, 3, 1)
        got = E.stn_input_assemble(*srcs, cs)
        assert tuple(got.shape) == (B, H, W, cs) and torch.equal(got[..., :want.shape[3]], want)
        assert float(got[..., want.shape[3]:].abs().sum()) == 0.0
    tmpl = torch.rand(1, 1, 36, 64, device="cuda", generator=g).repeat(5, 1, 1, 1)
    assert E.rows_all_equal(tmpl)
    tmpl[3, 0, 35, 63] += 1.0
    assert not E.rows_all_equal(tmpl)
    assert E.rows_all_equal(tmpl[:1])
    z = E.filled((3, 5), torch.float32, "cuda", 1.5)
    zi = E.filled((7,), torch.int32, "cuda", -3)
    assert torch.equal(z, torch.full((3, 5), 1.5, device="cuda")) and torch.equal(zi, torch.full((7,), -3, dtype=torch.int32, device="cuda"))


def test_pipeline_without_split_k_is_opt_in_and_fp32_close():
    """`net.pipeline_splitk = False`: predict_async() runs the ResNet-STN launches unsplit on the side stream (one frame per call:
    +10 % frames/s, profiles/r06_batch_sweep.txt) - another fp32 summation order, so theta moves in its last bits; the logits
    (UNet) stay bit-identical.  The default keeps predict()'s bits (tests/test_gpu_parity.py, smoke())."""
    B, w, h = 1, 320, 192
    net, _ = _net(B, w=w, h=h, seed=33)
    x = synth.smooth_frames(B, h, w, seed=95).cuda()
    with torch.no_grad():
        ref = {k: v.clone() for k, v in net.predict(x, consistency=True).items()}
        same = net.predict_async(x, consistency=True).result()
        assert all(torch.equal(same[k], ref[k]) for k in ref)                  # default: the same bits
        net.pipeline_splitk = False
        got = net.predict_async(x, consistency=True).result()
    torch.cuda.synchronize()
    assert torch.equal(got["logits"], ref["logits"])
    d = float((got["theta"] - ref["theta"]).abs().max())
    assert 0.0 < d < 1e-5, d               # differs (split-K was in use at this size) and stays at fp32 rounding level
    assert float((got["warp_mask"] != ref["warp_mask"]).float().mean()) < 1e-3
