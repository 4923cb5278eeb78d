"""BASELINE.json configs at their STATED sizes on the GPU, against vectors produced by the reference's own
classes (oracle/make_fixtures.py --configs c2,c5,c3; the Kornia leg - warp / POI - is the pinned-order
restatement oracle/warp_ref.py, see its header for the pin status):

* C2: 640x360, batch 16, NCAA template: theta / poi / consistency / logits, arg-max mask with the number
  of differing pixels and the margin they sit at, exact warp for the GPU's own theta, rounded POI pixels.
* C5: 1280x720, batch 16 through the real sub-batch path, 4-class pitch template, 33-point POI: all 16 frames
  against the golden vector.
* C3: one training forward + backward at 640x360 (B=2 of the 16): losses and, for every parameter, the
  gradient against an fp64 run of the reference classes - bounded by a multiple of the error the
  reference's own fp32 run has against fp64; and the forward pass + losses + BatchNorm running statistics at the
  stated batch of 16 (batch-statistics BatchNorm depends on it) against the reference classes' fp32 run.

Measured figures are appended to gpurun_out/r06_parity_full_size.jsonl when that directory exists (they are quoted
in DESIGN.md).
"""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import train_ref, warp_ref  # noqa: E402
from sfh_amd import synth  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")
torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))


FLIP_MARGIN = 2e-4                               # golden top-2 margin below which an arg-max pixel may differ
FLIP_CAP = {(640, 360): 16, (1280, 720): 48}     # ... and how many of the batch's pixels may (3.69 M / 14.7 M)
# |sum over an 8x8 block of (logit - golden)|.  The accumulation-order error of neighbouring pixels is correlated (same
# weights, similar inputs), so block sums move by up to 64 x 7e-5 = 4.3e-3 in EVERY mode (measured: f16x3 3.0e-3, exact
# bf16x6 operands 4.3e-3); a seam error of 1e-3 along one row of a block adds 8e-3.  (64 x the per-pixel bound = 3.2e-2.)
BLOCKSUM_TOL = 8e-3


def _record(tag, **kw):
    out = os.path.join(os.path.dirname(HERE), "gpurun_out")
    print(tag, json.dumps({k: v for k, v in kw.items() if k != "table"}))
    if os.path.isdir(out):
        with open(os.path.join(out, "r06_parity_full_size.jsonl"), "a") as f:
            f.write(json.dumps(dict(case=tag, **kw)) + "\n")


def _unpack2(packed, shape):
    b = np.unpackbits(packed, axis=-1).reshape(shape + (2,))
    return (b[..., 0] * 2 + b[..., 1]).astype(np.uint8)


def _net(template, wh, B, precision, seed=0, nearest=True):
    from sfh_amd.reconstructor import Reconstructor
    court = synth.load_court_template(template, 4, B)
    poi = synth.load_court_poi("pitch", B)
    net = Reconstructor(court.cuda(), poi.cuda(), target_size=wh, unet_size=wh, warp_size=wh,
                        warp_with_nearest=nearest)
    net.precision = precision
    sd = synth.synth_state_dict(net.state_dict(), seed)
    net.load_state_dict(sd)
    return net.cuda().eval(), sd, court, poi


def _coverage_errors(logits, g, n):
    """(max |d 8x8 block sum|, max |d| on the stored tile-edge rows, ... columns) of logits (CPU, NCHW) against the
    golden's coverage vectors (oracle/make_fixtures.py:_coverage_vectors): every logit of every frame is in a block sum."""
    _, C, H, W = logits.shape
    bs = logits[:n].double().reshape(n, C, H // 8, 8, W // 8, 8).sum(dim=(3, 5)).float()
    dblock = float((bs - torch.from_numpy(g["logits_blocksum8"])).abs().max())
    sel = logits[torch.from_numpy(g["line_frames"]).long()]
    drow = float((sel[:, :, torch.from_numpy(g["line_rows"]).long(), :] - torch.from_numpy(g["logits_rows"])).abs().max())
    dcol = float((sel[:, :, :, torch.from_numpy(g["line_cols"]).long()] - torch.from_numpy(g["logits_cols"])).abs().max())
    return dblock, drow, dcol


def _check_predict(tag, out, g, court, wh, nframes_golden):
    """Every output of predict() for the first `nframes_golden` frames against the golden vector, the exact
    warp check for all frames."""
    W, H = wh
    n = nframes_golden
    B = out["theta"].shape[0]
    theta = out["theta"].cpu()
    logits = out["logits"].cpu()
    dtheta = float((theta[:n] - torch.from_numpy(g["theta"])).abs().max())
    dlog = float((logits[:n, :, 4::16, 4::16] - torch.from_numpy(g["logits_sub"])).abs().max())
    dcons = float((out["consist_score"].cpu()[:n] - torch.from_numpy(g["consist"])).abs().max())
    dpoi = float((out["poi"].cpu()[:n] - torch.from_numpy(g["poi"])).abs().max())
    assert dtheta < 1e-4, dtheta                  # north_star: homography within 1e-4 abs
    assert dlog < 5e-4, dlog
    assert dpoi < 1e-4, dpoi

    # ---- arg-max mask: which pixels differ, and at what top-2 margin of the reference logits
    am_ref = _unpack2(g["argmax_2bit"], (n, H, W))
    am = logits[:n].argmax(1).numpy().astype(np.uint8)
    diff = np.argwhere(am != am_ref)
    margin = np.full((n, H * W), np.inf, np.float32)       # only margins below 1e-2 are stored
    margin[g["low_margin_frame"], g["low_margin_pixel"]] = g["low_margin_value"]
    margin = margin.reshape(n, H, W)
    dmarg = margin[diff[:, 0], diff[:, 1], diff[:, 2]] if len(diff) else np.zeros(0, np.float32)
    # An ABSOLUTE rule (it does not scale with this run's own error): a pixel may differ from the golden arg-max only
    # where the golden top-2 logits are closer than FLIP_MARGIN, and only FLIP_CAP[size] pixels of the batch may.
    safe = FLIP_MARGIN
    assert (dmarg < safe).all(), (len(diff), float(dmarg.max()), safe)
    assert len(diff) <= FLIP_CAP[(W, H)], (len(diff), FLIP_CAP[(W, H)])
    below = int((g["low_margin_value"] < safe).sum())
    # the reference labels with argmax(softmax(logits)) (utils/postprocess.py:10-11), the HIP path with argmax(logits): they
    # can differ only where the top-2 margin is inside oracle.torch_ref.SOFTMAX_TIE_MARGIN (fp32 softmax ties; DESIGN.md
    # section 5).  Golden pixels inside that band - each would also have to survive the 1e-4-level summation-order error:
    from oracle.torch_ref import SOFTMAX_TIE_MARGIN
    in_tie_band = int((g["low_margin_value"] < SOFTMAX_TIE_MARGIN).sum())

    # ---- every logit enters a compared quantity: 8x8 block sums of all channels of all frames (tile seams, frame
    # borders, the rows behind the odd 45 -> 22 pooling), and whole rows / columns at tile edges point by point
    dblock, drow, dcol = _coverage_errors(logits, g, n)
    assert dblock < BLOCKSUM_TOL, dblock          # a row of a block off by 1e-3 (a seam error) trips this
    assert drow < 5e-4 and dcol < 5e-4, (drow, dcol)

    # ---- nearest warp: the oracle's warp of the GPU's OWN theta must equal the GPU mask on every pixel
    wm = out["warp_mask"].cpu()
    want = (warp_ref.homography_warp(theta, court[:B], H, W, "nearest") * 4).to(torch.int32)
    nexact = int((wm != want).sum())
    assert nexact == 0, nexact
    wm_ref = _unpack2(g["warp_mask_2bit"], (n, H, W))
    warp_vs_golden = float((wm[:n].numpy() != wm_ref).mean())    # moves only with the 1e-7-level theta difference
    assert warp_vs_golden < 2e-3
    assert dcons < 2e-3 + 20.0 * warp_vs_golden, (dcons, warp_vs_golden)

    # ---- POI pixel coordinates int(round(p * W)) (predict.py:383): exact away from rounding ties
    p_ref, p = g["poi"], out["poi"].cpu().numpy()[:n]
    scale = np.array([W, H], np.float32)
    pix_ref, pix = np.rint(p_ref * scale), np.rint(p * scale)
    frac = np.abs((p_ref * scale) - np.floor(p_ref * scale) - 0.5)
    tie_band = 4.0 * dpoi * max(W, H) + 1e-6
    off_tie = frac > tie_band
    assert np.array_equal(pix[off_tie], pix_ref[off_tie])
    _record(tag, frames=B, frames_vs_golden=n, max_abs_dtheta=dtheta, max_abs_dlogits_sub=dlog,
            max_abs_dconsist=dcons, max_abs_dpoi=dpoi, argmax_pixels=int(n * H * W), argmax_differ=int(len(diff)),
            argmax_differ_max_margin=float(dmarg.max()) if len(diff) else 0.0, safe_margin=safe,
            max_abs_dblocksum8=dblock, max_abs_dlogits_tile_edge_rows=drow, max_abs_dlogits_tile_edge_cols=dcol,
            pixels_below_safe_margin=below, pixels_inside_softmax_tie_margin=in_tie_band, margin_hist=g["margin_hist"].sum(0).tolist(),
            margin_bins=g["margin_bins"].tolist(), warp_mismatch_vs_oracle_of_gpu_theta=nexact,
            warp_mismatch_frac_vs_golden_theta=warp_vs_golden, poi_points=int(off_tie.size),
            poi_points_in_tie_band=int((~off_tie).sum()), poi_pixels_differ=int((pix != pix_ref).sum()))


@pytest.mark.parametrize("precision", ["f16x3", "bf16x6", "fp32"])
def test_c2_640x360_batch16_golden(precision):
    g = np.load(os.path.join(GOLD, "c2_640x360_b16.npz"))
    net, _, court, _ = _net("ncaa_nc4_640x360", (640, 360), 16, precision)
    x = synth.frames_to_float(synth.synth_frames_u8(16, 360, 640, seed=0))
    with torch.no_grad():
        out = net.predict(x.cuda(), consistency=True, project_poi=True)
    torch.cuda.synchronize()
    assert out["warp_mask"].dtype == torch.int32 and tuple(out["warp_mask"].shape) == (16, 360, 640)
    _check_predict(f"C2 640x360 B=16 {precision}", out, g, court, (640, 360), 16)


@pytest.mark.parametrize("precision", ["f16x3", "bf16x6"])
def test_c2d_predict_py_default_geometry_unet640x360_warp1280x720(precision):
    """predict.py's DEFAULT geometry at full size (round 5): `court_size` / `warp_size` are raised to out_size = 1280x720
    while the UNet stays at 640x360 (predict.py:151-155), so `warp_mask` is (16, 720, 1280) and the consistency CE reads
    it through a nearest resize to the logits' size (models/reconstructor.py:226-240).  Golden from the reference's classes
    (oracle/make_fixtures.py --configs c2d; logits / theta are those of the C2 vector): theta, the 1280x720 warp mask,
    the consistency score through the resized mask, POI; all of C2's logit checks as well."""
    from oracle import torch_ref
    from sfh_amd.reconstructor import Reconstructor
    g2 = np.load(os.path.join(GOLD, "c2_640x360_b16.npz"))
    g = np.load(os.path.join(GOLD, "c2d_unet640x360_warp1280x720_b16.npz"))
    B, W, H, WW, WH = 16, 640, 360, 1280, 720
    court = synth.load_court_template("ncaa_nc4_1280x720", 4, B)
    poi = synth.load_court_poi("pitch", B)
    net = Reconstructor(court.cuda(), poi.cuda(), target_size=(W, H), unet_size=(W, H), warp_size=(WW, WH), warp_with_nearest=True)
    net.precision = precision
    net.load_state_dict(synth.synth_state_dict(net.state_dict(), 0))
    net.cuda().eval()
    x = synth.frames_to_float(synth.synth_frames_u8(B, H, W, seed=0))
    with torch.no_grad():
        out = net.predict(x.cuda(), consistency=True, project_poi=True)
        piped = net.predict_async(x.cuda(), consistency=True, project_poi=True).result()
    torch.cuda.synchronize()
    assert out["warp_mask"].dtype == torch.int32 and tuple(out["warp_mask"].shape) == (B, WH, WW)
    assert tuple(out["logits"].shape) == (B, 4, H, W) and tuple(out["consist_score"].shape) == (B,)
    assert all(torch.equal(piped[k], out[k]) for k in out)
    theta, logits = out["theta"].cpu(), out["logits"].cpu()
    dtheta = float((theta - torch.from_numpy(g["theta"])).abs().max())
    dpoi = float((out["poi"].cpu() - torch.from_numpy(g["poi"])).abs().max())
    dblock, drow, dcol = _coverage_errors(logits, g2, B)
    assert dtheta < 1e-4 and dpoi < 1e-4 and dblock < BLOCKSUM_TOL and drow < 5e-4 and dcol < 5e-4
    # the 1280x720 nearest warp of the GPU's OWN theta: exact on every pixel; against the golden theta's mask: moves only
    # with the 1e-7-level theta difference
    wm = out["warp_mask"].cpu()
    want = (warp_ref.homography_warp(theta, court, WH, WW, "nearest") * 4).to(torch.int32)
    nexact = int((wm != want).sum())
    assert nexact == 0, nexact
    wm_ref = _unpack2(g["warp_mask_2bit"], (B, WH, WW))
    warp_vs_golden = float((wm.numpy() != wm_ref).mean())
    assert warp_vs_golden < 2e-3
    # consistency through the nearest-resized mask: against the golden, and exactly the reference's formula on the GPU's
    # own logits and mask (F.interpolate nearest + cross_entropy on the CPU)
    dcons = float((out["consist_score"].cpu() - torch.from_numpy(g["consist"])).abs().max())
    assert dcons < 2e-3 + 20.0 * warp_vs_golden, (dcons, warp_vs_golden)
    m = torch.nn.functional.interpolate(wm.float().unsqueeze(1), size=(H, W), mode="nearest").squeeze(1).long()
    own = torch.nn.functional.cross_entropy(logits, m, reduction="none").mean(dim=(1, 2))
    down = float((out["consist_score"].cpu() - own).abs().max())
    assert down < 2e-5, down
    assert net.range_fallbacks == 0 and net.range_rescales == 0 and net.range_raises == 0
    _record(f"C2d unet 640x360 warp 1280x720 B=16 {precision}", frames=B, max_abs_dtheta=dtheta, max_abs_dpoi=dpoi,
            max_abs_dblocksum8=dblock, max_abs_dconsist=dcons, max_abs_dconsist_vs_torch_on_own_outputs=down,
            warp_mismatch_vs_oracle_of_gpu_theta=nexact, warp_mismatch_frac_vs_golden_theta=warp_vs_golden)


@pytest.mark.parametrize("precision", ["f16x3", "bf16x6"])
def test_c5_1280x720_batch16_pitch_template_poi(precision):
    """predict.py's HD configuration (predict.py:151-155,186-192): 16 frames in one call - whatever sub-batching
    the 32-bit tensor addressing needs happens inside predict()."""
    g = np.load(os.path.join(GOLD, "c5_1280x720_b16.npz"))
    net, _, court, _ = _net("pitch_v3_nc4_1280x720", (1280, 720), 16, precision)
    x = synth.frames_to_float(synth.synth_frames_u8(16, 720, 1280, seed=0))
    with torch.no_grad():
        out = net.predict(x.cuda(), consistency=True, project_poi=True)
    torch.cuda.synchronize()
    assert tuple(out["logits"].shape) == (16, 4, 720, 1280) and tuple(out["poi"].shape) == (16, 33, 2)
    _check_predict(f"C5 1280x720 B=16 {precision}", out, g, court, (1280, 720), 16)
    assert net.range_fallbacks == 0 and net.range_rescales == 0
    # frames are independent (eval-mode BatchNorm): a frame gives the same bits wherever it sits in the batch (a batch
    # of another SIZE may split the K loop of the small ResNet layers differently, engine.choose_ksplit)
    perm = torch.tensor([5, 0, 11, 3, 15, 1, 8, 2, 13, 4, 9, 6, 14, 7, 10, 12])
    with torch.no_grad():
        mixed = net.predict(x[perm].cuda(), consistency=True, project_poi=True)
    assert torch.equal(mixed["theta"], out["theta"][perm.cuda()]) and torch.equal(mixed["warp_mask"], out["warp_mask"][perm.cuda()])
    assert torch.equal(mixed["logits"], out["logits"][perm.cuda()])


@pytest.mark.parametrize("precision", ["f16x3", "bf16x6", "fp32"])
def test_theta_only_predict_640x360_b8(precision):
    """predict.py with --req_outputs theta (predict.py:169-174: no warper, no consistency, no POI; BASELINE config 1's
    workload - 8 frames of 640x360 - on the HIP path): the dict holds exactly logits + theta; theta / logits against
    the reference-class golden (first 8 of C2's 16 frames: frames are independent in eval mode) and against the CPU
    restatement's own predict(); with the reference's initialisation of the regression head (zero weight, identity
    bias: models/resnet.py:206-208) theta is the identity BIT FOR BIT whatever the conv stack computes."""
    from oracle import torch_ref
    from sfh_amd.reconstructor import Reconstructor
    g = np.load(os.path.join(GOLD, "c2_640x360_b16.npz"))
    B, W, H = 8, 640, 360
    court = synth.load_court_template("ncaa_nc4_640x360", 4, B)
    poi = synth.load_court_poi("pitch", B)
    net = Reconstructor(court.cuda(), poi.cuda(), target_size=(W, H), unet_size=(W, H), warp_size=(W, H), use_warper=False)
    net.precision = precision
    sd = synth.synth_state_dict(net.state_dict(), 0)
    net.load_state_dict(sd)
    net.cuda().eval()
    x = synth.frames_to_float(synth.synth_frames_u8(16, H, W, seed=0))[:B].contiguous()
    with torch.no_grad():
        out = net.predict(x.cuda(), consistency=False, project_poi=False)
        also = net.predict(x.cuda(), consistency=True, project_poi=False)      # no warper: consistency has nothing to score
    torch.cuda.synchronize()
    assert set(out) == {"logits", "theta"} and set(also) == {"logits", "theta"}
    assert tuple(out["theta"].shape) == (B, 1, 3, 3) and out["theta"].dtype == torch.float32
    assert tuple(out["logits"].shape) == (B, 4, H, W)
    assert torch.equal(out["theta"], also["theta"]) and torch.equal(out["logits"], also["logits"])
    theta, logits = out["theta"].cpu(), out["logits"].cpu()
    dtheta = float((theta - torch.from_numpy(g["theta"][:B])).abs().max())
    dlog = float((logits[:, :, 4::16, 4::16] - torch.from_numpy(g["logits_sub"][:B])).abs().max())
    bs = logits.double().reshape(B, 4, H // 8, 8, W // 8, 8).sum(dim=(3, 5)).float()
    dblock = float((bs - torch.from_numpy(g["logits_blocksum8"][:B])).abs().max())
    assert dtheta < 1e-4 and dlog < 5e-4 and dblock < BLOCKSUM_TOL, (dtheta, dlog, dblock)
    # the CPU restatement's own predict() on two of the frames: same keys, same values
    want = torch_ref.predict(x[:2], sd, None, None, use_warper=False, consistency=False, project_poi=False)
    assert set(want) == set(out)
    assert float((theta[:2] - want["theta"]).abs().max()) < 1e-4
    assert float((logits[:2] - want["logits"]).abs().max()) < 5e-4
    # reference initialisation of the head: theta == I exactly
    sd_id = dict(sd)
    sd_id["resnet_reg.reg.weight"] = torch.zeros_like(sd["resnet_reg.reg.weight"])
    sd_id["resnet_reg.reg.bias"] = torch.eye(3).reshape(9).clone()
    net.load_state_dict(sd_id)
    with torch.no_grad():
        ident = net.predict(x.cuda(), consistency=False, project_poi=False)["theta"].cpu()
    assert torch.equal(ident, torch.eye(3).expand(B, 1, 3, 3))
    _record(f"C1 analogue theta-only 640x360 B=8 {precision}", frames=B, max_abs_dtheta=dtheta,
            max_abs_dlogits_sub=dlog, max_abs_dblocksum8=dblock, identity_head_bit_exact=True)


def test_warp_arithmetic_selftest():
    """The warp kernel's reciprocal (hardware rcp + two FMA Newton steps) and meshgrid division (one FMA
    residual step) give the IEEE quotients on EVERY input of their domain (exhaustive sweep on the GPU)."""
    import ctypes
    from sfh_amd import _lib
    lib = _lib.load()
    bad = torch.ones(2, dtype=torch.int64, device="cuda")
    _lib.check(lib.sfh_selftest_warp_arith(ctypes.c_void_p(bad.data_ptr()),
                                           ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "selftest")
    torch.cuda.synchronize()
    assert bad.tolist() == [0, 0], bad.tolist()


@pytest.mark.parametrize("hw", [(61, 97), (45, 640), (7, 130), (360, 640)])
@pytest.mark.parametrize("nearest", [True, False])
def test_warp_writes_nothing_outside_its_output(hw, nearest):
    """Frame heights that are not a multiple of the rows a wave handles, widths that are not a multiple of 64:
    the bytes behind the last frame (and between nothing else) must stay untouched, and every frame must equal
    the oracle - a row of the last partial tile must not spill into the next frame or past the tensor."""
    import ctypes
    from sfh_amd import _lib
    lib = _lib.load()
    h, w = hw
    B = 3
    th = torch.tensor(synth.REALISTIC_THETAS)[[0, 1, 0]].reshape(B, 3, 3).contiguous()
    th[2] = torch.eye(3) + 0.03 * torch.randn(3, 3, generator=torch.Generator().manual_seed(h))
    tmpl = synth.load_court_template("ncaa_nc4_640x360", 4, 1)
    guard = 4096
    of = torch.full((B * h * w + guard,), -7.0, device="cuda")
    oi = torch.full((B * h * w + guard,), -7, dtype=torch.int32, device="cuda")
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    thc, tc = th.cuda(), tmpl.cuda()
    _lib.check(lib.sfh_homography_warp_fwd(p(thc), p(tc), 0, 360, 640, B, h, w, 0 if nearest else 1, 4.0, p(of), p(oi), st), "warp")
    torch.cuda.synchronize()
    assert float(of[B * h * w:].min()) == -7.0 and float(of[B * h * w:].max()) == -7.0
    assert int(oi[B * h * w:].min()) == -7 and int(oi[B * h * w:].max()) == -7
    want = warp_ref.homography_warp(th.reshape(B, 1, 3, 3), tmpl.expand(B, -1, -1, -1), h, w, "nearest" if nearest else "bilinear")
    got = of[:B * h * w].reshape(B, h, w).cpu()
    if nearest:
        assert torch.equal(got, want)
        assert torch.equal(oi[:B * h * w].reshape(B, h, w).cpu(), (want * 4).to(torch.int32))
    else:
        assert float((got - want).abs().max()) < 1e-6


def test_warp_huge_and_non_finite_theta_take_the_ieee_path():
    """A wave-uniform test on theta routes |t| > 2^59, inf and NaN to the IEEE-division / NaN-tolerant path of the
    warp kernel (LEVEL 0): exact against the oracle in nearest mode (non-finite coordinates sample 0)."""
    from sfh_amd import engine as E
    base = torch.tensor(synth.REALISTIC_THETAS)
    th = torch.stack([base[0] * 1e25, base[1] * -3e30, base[0].clone(), base[1].clone(), torch.eye(3)])
    th[2, 0, 1] = float("inf")
    th[3, 2, 2] = float("nan")
    th[4, 0, 0] = 2.0 ** 60          # finite, but beyond the range the fast reciprocal was verified on
    tmpl = synth.load_court_template("ncaa_nc4_640x360", 4, th.shape[0])
    want = warp_ref.homography_warp(th.reshape(-1, 1, 3, 3), tmpl, 61, 97, "nearest")
    of, oi = E.homography_warp(th.reshape(-1, 1, 3, 3).cuda(), tmpl.cuda(), 61, 97, True, scale=4.0, want_f32=True,
                               want_i32=True, shared_template=True)
    torch.cuda.synchronize()
    assert torch.equal(of.cpu(), want)
    assert torch.equal(oi.cpu(), (want * 4).to(torch.int32))
    assert float(want[:2].abs().sum()) > 0       # the scaled matrices still hit the template


def test_public_methods_direct():
    """net.forward_unet(), net.warp(), net.transform_poi() called the way the reference's callers do
    (models/reconstructor.py:109-158), not through predict()."""
    from oracle import torch_ref
    net, sd, court, poi = _net("ncaa_nc4_640x360", (640, 360), 2, "bf16x6", seed=19)
    x = synth.smooth_frames(2, 360, 640, seed=5)
    with torch.no_grad():
        logits, x_top, uv = net.forward_unet(x.cuda())
        want_l, want_top, _ = torch_ref.forward_unet(x, sd)
    assert uv is None and tuple(x_top.shape) == (2, 1024, 22, 40)
    assert float((logits.cpu() - want_l).abs().max()) < 5e-4
    assert float((x_top.cpu() - want_top).abs().max()) < 5e-4 * max(1.0, float(want_top.abs().max()))
    th = torch.tensor(synth.REALISTIC_THETAS).reshape(2, 1, 3, 3)
    with torch.no_grad():
        wm = net.warp(th.cuda(), net.court_img)           # nearest (warp_with_nearest=True), float output
        pp = net.transform_poi(th.cuda(), net.court_poi)
        pr = net.transform_poi(th.cuda(), net.court_poi, normalize=False)
    assert wm.dtype == torch.float32 and tuple(wm.shape) == (2, 360, 640)
    assert torch.equal(wm.cpu(), warp_ref.homography_warp(th, court, 360, 640, "nearest"))
    assert float((pp.cpu() - torch_ref.transform_poi(th, poi)).abs().max()) < 1e-5
    assert float((pr.cpu() - torch_ref.transform_poi(th, poi, normalize=False)).abs().max()) < 1e-5
    net.warp_with_nearest = False
    with torch.no_grad():
        wb = net.warp(th.cuda(), net.court_img)
    assert float((wb.cpu() - warp_ref.homography_warp(th, court, 360, 640, "bilinear")).abs().max()) < 1e-6


@pytest.mark.parametrize("precision", ["f16x3", "bf16x6", "fp32"])  # training precisions
def test_c3_batch16_forward_losses_and_running_stats(precision, monkeypatch):
    """BASELINE config 3 at its stated batch: 16 frames of 640x360 through TrainStep.loss_and_grads (forward with
    batch statistics, the four losses, the whole backward) against the reference classes' own train()-mode forward
    in fp32 (oracle/make_fixtures.py:make_c3_b16_golden): loss values, theta, sub-sampled logits / warp, and EVERY
    BatchNorm layer's running_mean / running_var / num_batches_tracked after the step."""
    from oracle.fixture_inputs import c3_batch
    from sfh_amd import training
    from sfh_amd.reconstructor import Reconstructor
    monkeypatch.setenv("SFH_TRAIN_PRECISION", precision)
    g = np.load(os.path.join(GOLD, "c3_fwd_640x360_b16.npz"))
    B, H, W = 16, 360, 640
    court = synth.load_court_template("ncaa_nc4_640x360", 4, B)
    poi = synth.load_court_poi("pitch", B)
    net = Reconstructor(court.cuda(), poi.cuda(), target_size=(W, H), unet_size=(W, H), warp_size=(W, H))
    net.load_state_dict(synth.synth_state_dict(net.state_dict(), 0))
    net.cuda().train()
    ts = training.TrainStep(net, lr=1e-5, weight_decay=1e-8, seg_lambda=1.0, rec_lambda=1.0, reproj_lambda=1.0,
                            consist_lambda=1.0)
    x = synth.frames_to_float(synth.synth_frames_u8(B, H, W, seed=0)).cuda()
    batch = {k: v.cuda() for k, v in c3_batch(B, H, W, poi.shape[1]).items()}
    losses = ts.loss_and_grads(x, batch).cpu().numpy()      # [seg, rec, consist, reproj]
    torch.cuda.synchronize()
    got = dict(zip(("seg", "rec", "consist", "reproj"), losses.tolist()))
    rec = {}
    for k, tol in (("seg", 2e-5), ("rec", 2e-5), ("reproj", 2e-5), ("consist", 1e-3)):
        want = float(g[f"loss.{k}"])
        rec[k] = (got[k], want)
        # trunc(warp * 4) targets of the consistency term flip with the last bit of the bilinear warp
        assert abs(got[k] - want) < tol * max(1.0, abs(want)), (k, got[k], want)
    assert ts.range_fallbacks == 0
    # the training forward's logits: sub-sample, 8x8 block sums of every frame and tile-edge lines (batch-statistics
    # BatchNorm couples the frames: the golden is the reference classes' train()-mode forward of the same 16 frames)
    lg = ts.last_outputs["logits"].detach().cpu()
    dth = float((ts.last_outputs["theta"].detach().cpu().reshape(B, 1, 3, 3) - torch.from_numpy(g["theta"])).abs().max())
    dsub = float((lg[:, :, 4::16, 4::16] - torch.from_numpy(g["logits_sub"])).abs().max())
    dblock, drow, dcol = _coverage_errors(lg, g, B)
    assert dth < 1e-4 and dsub < 5e-4 and drow < 5e-4 and dcol < 5e-4 and dblock < BLOCKSUM_TOL, (dth, dsub, drow, dcol, dblock)
    bufs = dict(net.named_buffers())
    worst = {"running_mean": 0.0, "running_var": 0.0}
    names = [str(n) for n in g["buffers"]]
    assert len(names) == 162 and all(n in bufs for n in names)
    for n in names:
        want, have = g[f"buf.{n}"], bufs[n].detach().cpu().numpy()
        if n.endswith("num_batches_tracked"):
            assert int(have) == int(want) == 101, n      # the synthetic checkpoint starts at 100
            continue
        kind = n.rsplit(".", 1)[1]
        # relative to the size of the statistic's own tensor (means of some channels sit near zero)
        err = float(np.abs(have - want).max() / max(np.abs(want).max(), 1e-6))
        worst[kind] = max(worst[kind], err)
        assert err < 2e-4, (n, err)
    _record(f"C3 forward 640x360 B=16 {precision}", losses=rec, batchnorm_layers=len(names) // 3, max_abs_dtheta=dth,
            max_abs_dlogits_sub=dsub, max_abs_dblocksum8=dblock, max_abs_dlogits_tile_edge_rows=drow,
            max_abs_dlogits_tile_edge_cols=dcol,
            max_rel_err_running_mean=worst["running_mean"], max_rel_err_running_var=worst["running_var"])


def _masked_tail_gradients(net, cap, keys):
    """fp64 gradients of the ResNet blocks named in `keys` (BasicBlocks of the last stage) and of everything behind
    them, with the ReLU decisions of the captured GPU pass imposed: block input and masks from training.CAPTURE, batch-
    statistics BatchNorm recomputed in fp64, backward from the captured theta gradient.  -> {state_dict key: gradient}"""
    import torch.nn.functional as F
    blocks = sorted({".".join(k.split(".")[1:3]) for k in keys})          # e.g. "layer4.1"
    if not blocks:
        return {}
    li = int(blocks[0][5])
    first = min(int(b.split(".")[1]) for b in blocks)
    stage = getattr(net.resnet_reg, f"layer{li}")
    assert li == 4 and first >= 1, blocks          # blocks without a downsample branch, up to the head
    leaves = {}

    def leaf(name, t):
        leaves[name] = t.detach().cpu().double().requires_grad_(True)
        return leaves[name]

    nchw = lambda t: t.detach().cpu().double().permute(0, 3, 1, 2)
    x = nchw(cap[f"layer{li}.{first}"]["in"])
    for bi in range(first, len(stage)):
        blk, c = stage[bi], cap[f"layer{li}.{bi}"]
        pre = f"resnet_reg.layer{li}.{bi}."
        m1, m2 = (nchw(c["t"]) > 0).double(), (nchw(c["out"]) > 0).double()
        t = F.conv2d(x, leaf(pre + "conv1.weight", blk.conv1.weight), padding=1)
        t = F.batch_norm(t, None, None, leaf(pre + "bn1.weight", blk.bn1.weight), leaf(pre + "bn1.bias", blk.bn1.bias), True, 0.1, blk.bn1.eps) * m1
        u = F.conv2d(t, leaf(pre + "conv2.weight", blk.conv2.weight), padding=1)
        u = F.batch_norm(u, None, None, leaf(pre + "bn2.weight", blk.bn2.weight), leaf(pre + "bn2.bias", blk.bn2.bias), True, 0.1, blk.bn2.eps)
        x = (u + x) * m2
    theta = F.linear(x.mean(dim=(2, 3)), leaf("resnet_reg.reg.weight", net.resnet_reg.reg.weight),
                     leaf("resnet_reg.reg.bias", net.resnet_reg.reg.bias))
    theta.backward(cap["dtheta"].detach().cpu().double().reshape(theta.shape))
    return {k: v.grad for k, v in leaves.items()}


@pytest.mark.parametrize("precision", ["f16x3", "bf16x6", "fp32"])  # training precisions
def test_c3_training_step_640x360_vs_fp64_reference(precision, monkeypatch):
    """BASELINE config 3 at 640x360 (2 of the 16 frames): the reference classes under train() + autograd
    produced loss values and gradients in fp32 and in fp64 (oracle/make_fixtures.py:make_c3_golden).  The HIP
    training path must reproduce the losses, and every parameter gradient must sit within 4x of the distance
    the reference's own fp32 run keeps from the fp64 gradient (per tensor, or the upper quartile of its stage where
    the tensor's own fp32 figure is smaller).  A tensor beyond 4x its OWN fp32 error must pass a second, flip-free
    check: its gradient re-derived in fp64 under the ReLU decisions of the GPU pass (_masked_tail_gradients).  All
    three train precisions."""
    from oracle.fixture_inputs import c3_batch, grad_sample_index
    from sfh_amd.reconstructor import Reconstructor
    monkeypatch.setenv("SFH_TRAIN_PRECISION", precision)
    g = np.load(os.path.join(GOLD, "c3_train_640x360_b2.npz"))
    B, H, W = 2, 360, 640
    court = synth.load_court_template("ncaa_nc4_640x360", 4, B)
    poi = synth.load_court_poi("pitch", B)
    net = Reconstructor(court.cuda(), poi.cuda(), target_size=(W, H), unet_size=(W, H), warp_size=(W, H))
    net.load_state_dict(synth.synth_state_dict(net.state_dict(), 0))
    net.cuda().train()
    x = synth.frames_to_float(synth.synth_frames_u8(B, H, W, seed=0))
    batch = {k: v.cuda() for k, v in c3_batch(B, H, W, poi.shape[1]).items()}
    from sfh_amd import training
    cap = training.CAPTURE = {}
    try:
        preds = net(x.cuda())
        loss = train_ref.losses(preds, batch)
        loss["total"].backward()
    finally:
        training.CAPTURE = None
    torch.cuda.synchronize()
    assert float((preds["theta"].detach().cpu().double() - torch.from_numpy(g["theta_f64"])).abs().max()) < 1e-4
    dl = float((preds["logits"].detach().cpu()[:, :, 4::16, 4::16].double() - torch.from_numpy(g["logits_sub_f64"])).abs().max())
    dl32 = float(np.abs(g["logits_sub_f32"].astype(np.float64) - g["logits_sub_f64"]).max())
    assert dl < 3.0 * dl32 + 2e-5, (dl, dl32)
    lrec = {}
    for k in ("seg", "rec", "reproj", "consist", "total"):
        got, w64, w32 = float(loss[k].detach()), float(g[f"loss_f64.{k}"]), float(g[f"loss_f32.{k}"])
        lrec[k] = (got, w32, w64)
        # trunc(warp * 4) targets of the consistency term flip with the last bit of the bilinear warp
        tol = 3.0 * abs(w32 - w64) + (1e-3 if k in ("consist", "total") else 2e-5) * max(1.0, abs(w64))
        assert abs(got - w64) < tol, (k, got, w32, w64)
    names = [str(n) for n in g["names"]]
    params = dict(net.named_parameters())

    def stage(k):   # tensors of one stage see the same upstream gradient noise
        p = k.split(".")
        if p[0] != "resnet_reg":
            return p[0]
        return "resnet_reg." + (p[1] if p[1].startswith("layer") or p[1] == "reg" else "stem")

    e32_stage = {}
    for i, k in enumerate(names):
        if g[f"stat.{i}"][0] >= 1e-9:
            e32_stage.setdefault(stage(k), []).append(float(g[f"stat.{i}"][2]))
    e32_stage = {k: float(np.quantile(v, 0.75)) for k, v in e32_stage.items()}
    rows, worst = [], []
    for i, k in enumerate(names):
        g64 = g[f"g64.{i}"]
        n64, e32_full, e32 = g[f"stat.{i}"]
        grad = params[k].grad
        assert grad is not None, k
        got = grad.detach().reshape(-1)[torch.from_numpy(grad_sample_index(grad.numel())).cuda()].cpu().double().numpy()
        ns = np.linalg.norm(g64)
        if ns < 1e-12 * max(1.0, n64) or n64 < 1e-9:
            # conv bias in front of BatchNorm: the exact gradient is zero
            assert np.abs(got).max() < 1e-6, (k, np.abs(got).max())
            continue
        e = np.linalg.norm(got - g64) / ns
        rows.append((k, e, e32))
        # The error of either fp32-grade run against fp64 is made of discrete events (ReLU / max-pool decisions
        # that flip with the forward rounding), so a single tensor's fp32 figure can be luckily small: the
        # yardstick is the larger of the tensor's own fp32 error and the upper-quartile fp32 error of its stage.
        if e > 4.0 * max(e32, e32_stage[stage(k)]) + 1e-5:
            worst.append((k, e, e32))
    # ---- second check for every tensor beyond 4x its OWN fp32 error (they passed above only through the yardstick of
    # their stage): all of them sit in the last ResNet stage (12x20 pixels: one flipped ReLU moves a BatchNorm
    # gradient by 1e-3).  Re-derive that stage's gradients in fp64 UNDER THE RELU DECISIONS THIS PASS TOOK: its input
    # and its ReLU masks come from the GPU pass (training.CAPTURE), the theta gradient it started from as well; what is
    # left is arithmetic, and that has to agree to 2e-4.
    offenders = [k for k, e, e32 in rows if e > 4.0 * max(e32, 1e-6)]
    masked = _masked_tail_gradients(net, cap, [k for k in offenders if k.startswith("resnet_reg.layer4.")])
    tail = {}
    for k in offenders:
        assert k in masked, f"{k}: beyond 4x its fp32 error and outside the stage the masked re-derivation covers"
        got = params[k].grad.detach().cpu().double().reshape(-1)
        want = masked[k].reshape(-1)
        tail[k] = float((got - want).norm() / want.norm())
        assert tail[k] < 2e-4, (k, tail[k])
    ratios = np.array([e / max(e32, 1e-6) for _, e, e32 in rows])
    errs = np.array([e for _, e, _ in rows])
    _record(f"C3 train 640x360 B=2 {precision}", losses=lrec, tensors=len(rows), median_err=float(np.median(errs)),
            max_err=float(errs.max()), median_ratio_to_fp32=float(np.median(ratios)), max_ratio_to_fp32=float(ratios.max()),
            over_bound=[(k, float(e), float(e32)) for k, e, e32 in worst], max_abs_dlogits=dl, fp32_max_abs_dlogits=dl32,
            beyond_4x_own_fp32_error_rederived_under_gpu_relu_masks=tail,
            table=[(k, float(e), float(e32)) for k, e, e32 in rows])
    assert not worst, worst


@pytest.mark.parametrize("precision", ["f16x3", "bf16x6"])  # training precisions
def test_c3_batch16_gradients_and_weight_update(precision, monkeypatch):
    """BASELINE config 3 at its stated batch WITH the backward pass and the optimizer step (train.py:155-237): one
    `TrainStep.step` on 16 frames of 640x360 against the reference's own classes under torch autograd in fp32 on the same
    16 frames, followed by clip_grad_value_(0.1) and one RMSprop(momentum 0.9) step (oracle/make_fixtures.py:
    make_c3_b16_grad_golden).  Every parameter gradient: relative L2 distance on a fixed sample within 6x the distance
    the reference's fp32 run keeps from fp64 (per tensor, or the upper quartile of its stage: the B=2 vector - an fp64 run
    of 16 frames does not fit the build container); the weight update w_after - w_before: equal to 3 % of the step on at
    least 99.5 % of the sampled elements whose golden gradient is not vanishing (RMSprop's first step is lr * g / (0.1 |g|
    + eps): about 10 * lr * sign(g); "not vanishing" = beyond 50x the tensor's measured gradient error, so that the sign is
    not a matter of rounding: about 6000 of the 60000 sampled elements).  The gradients of a train-mode pass carry percent-level
    differences between ANY two fp32-grade runs (ReLU / max-pool decisions that flip with the forward rounding): the
    reference's own fp32 run sits 1.5 % (median over the tensors) from its fp64 run at B=2, and so does this path."""
    from oracle.fixture_inputs import c3_batch, grad_sample_index
    from sfh_amd import training
    from sfh_amd.reconstructor import Reconstructor
    monkeypatch.setenv("SFH_TRAIN_PRECISION", precision)
    g = np.load(os.path.join(GOLD, "c3_grads_640x360_b16.npz"))
    g2 = np.load(os.path.join(GOLD, "c3_train_640x360_b2.npz"))
    B, H, W = 16, 360, 640
    court = synth.load_court_template("ncaa_nc4_640x360", 4, B)
    poi = synth.load_court_poi("pitch", B)
    net = Reconstructor(court.cuda(), poi.cuda(), target_size=(W, H), unet_size=(W, H), warp_size=(W, H))
    net.load_state_dict(synth.synth_state_dict(net.state_dict(), 0))
    net.cuda().train()
    lr = 1e-5
    ts = training.TrainStep(net, lr=lr, weight_decay=1e-8, seg_lambda=1.0, rec_lambda=1.0, reproj_lambda=1.0, consist_lambda=1.0)
    x = synth.frames_to_float(synth.synth_frames_u8(B, H, W, seed=0)).cuda()
    batch = {k: v.cuda() for k, v in c3_batch(B, H, W, poi.shape[1]).items()}
    names = [str(n) for n in g["names"]]
    params = dict(net.named_parameters())
    assert names == [str(n) for n in g2["names"]] and set(names) == set(params)
    before = {k: params[k].detach().clone() for k in names}
    losses = ts.step(x, batch).cpu().numpy()
    torch.cuda.synchronize()
    assert ts.range_fallbacks == 0
    for k, v in zip(("seg", "rec", "consist", "reproj"), losses.tolist()):
        want = float(g[f"loss.{k}"])
        assert abs(v - want) < (1e-3 if k == "consist" else 2e-5) * max(1.0, abs(want)), (k, v, want)
    grads = {ts.names(p): gr for p, gr in zip(ts.params, ts.grads)}

    def stage(k):
        p = k.split(".")
        if p[0] != "resnet_reg":
            return p[0]
        return "resnet_reg." + (p[1] if p[1].startswith("layer") or p[1] == "reg" else "stem")

    e32_stage = {}
    for i, k in enumerate(names):
        if g2[f"stat.{i}"][0] >= 1e-9:
            e32_stage.setdefault(stage(k), []).append(float(g2[f"stat.{i}"][2]))
    e32_stage = {k: float(np.quantile(v, 0.75)) for k, v in e32_stage.items()}
    rows, bad = [], []
    upd_total = upd_off = 0
    for i, k in enumerate(names):
        want = g[f"g.{i}"].astype(np.float64)
        idx = torch.from_numpy(grad_sample_index(params[k].numel())).cuda()
        got = grads[k].detach().reshape(-1)[idx].cpu().double().numpy()
        gn = float(g[f"gnorm.{i}"])
        ns = np.linalg.norm(want)
        if gn < 1e-9 or ns < 1e-12 * max(1.0, gn):
            assert np.abs(got).max() < 1e-6, (k, np.abs(got).max())      # conv bias in front of BatchNorm: exactly zero
            continue
        e = float(np.linalg.norm(got - want) / ns)
        yard = max(float(g2[f"stat.{i}"][2]), e32_stage[stage(k)])
        rows.append((k, e, yard))
        if e > 6.0 * yard + 2e-4:       # (floor: the regression head's gradient is reproduced to 1e-5 by the fp32 run itself)
            bad.append((k, e, yard))
        # the optimizer step: elements whose golden gradient is clearly non-zero move by ~ 10 * lr * sign(g)
        dw_want = g[f"dw.{i}"].astype(np.float64)
        dw_got = (params[k].detach() - before[k]).reshape(-1)[idx].cpu().double().numpy()
        sel = np.abs(want) > max(1e-5, 50.0 * e * ns / np.sqrt(len(want)))
        upd_total += int(sel.sum())
        upd_off += int((np.abs(dw_got - dw_want)[sel] > 0.03 * 10.0 * lr + 1e-7 * np.abs(before[k].reshape(-1)[idx].cpu().numpy())[sel]).sum())
    errs = np.array([e for _, e, _ in rows])
    ratios = np.array([e / max(y, 1e-6) for _, e, y in rows])
    _record(f"C3 step 640x360 B=16 {precision}", tensors=len(rows), median_err_vs_fp32_reference=float(np.median(errs)),
            max_err_vs_fp32_reference=float(errs.max()), median_ratio_to_fp32_yardstick=float(np.median(ratios)),
            max_ratio_to_fp32_yardstick=float(ratios.max()), over_bound=[(k, float(e), float(y)) for k, e, y in bad],
            update_elements_checked=upd_total, update_elements_off=upd_off)
    assert not bad, bad
    assert upd_total > 3000 and upd_off <= 0.005 * upd_total, (upd_off, upd_total)


def test_device_calibration_probe():
    """bench.py's `device_calibration` (sfh_probe_mfma_f16 through the C-ABI): a register-resident fp16 MFMA loop reports a rate
    between a fifth of and the full dense peak, and the clock the chip held inside the kernel is a plausible shader clock."""
    import importlib.util
    root = os.path.dirname(HERE)
    spec = importlib.util.spec_from_file_location("bench_for_probe", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    c = bench.device_calibration(torch.device("cuda", 0), ms_target=10.0)
    assert 500.0 < c["mfma_f16_tflops"] <= 2600.0, c
    assert 0.8 < c["in_kernel_clock_ghz"] < 2.6, c
    assert c["compute_units"] >= 64 and c["launches"] == 5
    assert c["power"] is None or c["power"]["max_w"] > 50.0
