"""Generates the committed fixtures.  Runs ONLY in the build container (it reads
/root/reference); nothing under tests/, bench.py or the package imports this file.

Two kinds of output:

1. ``sports-field-homography_amd/data/*.npy`` - input DATA derived from the reference's
   asset files (court class-id templates resized NEAREST like utils/dataset.py:51-53, the
   RGBA pitch mask converted to ids with the colour table of
   dataset_utils/preparation.py:216-239, POI coordinates normalised like
   utils/dataset.py:78-79).
2. ``tests/golden/*.npz`` - golden input/output vectors produced by the reference's OWN
   classes (``unet/unet_parts.py``, ``models/resnet.py``; imported by file path because
   the package ``__init__`` files pull in Kornia, which is not installed) on the
   deterministic synthetic weights of ``sfh_amd.synth``.  These pin ``oracle/torch_ref.py``.

Usage:  python oracle/make_fixtures.py [--full]     (--full adds the 640x360 end-to-end vector)
        python oracle/make_fixtures.py --configs c2,c2d,c5,c3,c3b16,c3b16grad   (BASELINE configs at their stated sizes)
"""
import argparse
import importlib.util
import json
import os
import sys

import numpy as np
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference"
GOLD = os.path.join(ROOT, "tests", "golden")
DATA = os.path.join(ROOT, "sports-field-homography_amd", "data")

from sfh_amd import synth  # noqa: E402
from oracle import torch_ref  # noqa: E402
from oracle.fixture_inputs import c3_batch, grad_sample_index  # noqa: E402


def _load(name, rel):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def make_data():
    from PIL import Image
    os.makedirs(DATA, exist_ok=True)
    im = Image.open(os.path.join(REF, "assets/mask_ncaa_v4_nc4_m_onehot.png"))
    for (w, h) in [(640, 360), (1280, 720)]:
        ids = np.array(im.resize((w, h), resample=Image.NEAREST)).astype(np.uint8)
        np.save(os.path.join(DATA, f"court_ids_ncaa_nc4_{w}x{h}.npy"), ids)
    rgba = np.array(Image.open(os.path.join(REF, "assets/pitch_mask_v3_nc4_hd.png")))
    rgb = rgba[..., :3]
    ids = np.zeros(rgb.shape[:2], np.uint8)
    # dataset_utils/preparation.py:218-221 (colours given there in BGR order)
    ids[(rgb == (0, 255, 0)).all(-1)] = 1
    ids[(rgb == (0, 0, 255)).all(-1)] = 2   # BGR (255,0,0)
    ids[(rgb == (255, 0, 0)).all(-1)] = 3   # BGR (0,0,255)
    np.save(os.path.join(DATA, "court_ids_pitch_v3_nc4_1280x720.npy"), ids)
    small = np.array(Image.fromarray(ids).resize((640, 360), resample=Image.NEAREST))
    np.save(os.path.join(DATA, "court_ids_pitch_v3_nc4_640x360.npy"), small)
    for tag, fn in [("pitch", "template_pitch_points.json"), ("ncaa", "template_ncaa_v4_points.json")]:
        d = json.load(open(os.path.join(REF, "assets", fn)))
        assert d["ranges"][0] == 1.0 and d["ranges"][1] == 1.0
        pts = np.array([[(p["coords"][0] - 0.5) * 2, (p["coords"][1] - 0.5) * 2] for p in d["points"]])
        np.save(os.path.join(DATA, f"court_poi_{tag}.npy"), pts.astype(np.float32))
        print(tag, "poi", pts.shape)


class _RefNet(nn.Module):
    """The reference's UNet + ResNetSTN sub-modules under the attribute names of
    models/reconstructor.py:66-97 (Reconstructor itself cannot be imported: Kornia)."""

    def __init__(self, up_mod, rn_mod, mask_classes=4, bilinear=False, uv=False, resnet="resnet34"):
        super().__init__()
        f = 2 if bilinear else 1
        self.inc = up_mod.DoubleConv(3, 64)
        self.down1 = up_mod.Down(64, 128)
        self.down2 = up_mod.Down(128, 256)
        self.down3 = up_mod.Down(256, 512)
        self.down4 = up_mod.Down(512, 1024 // f)
        self.up1 = up_mod.Up(1024, 512 // f, bilinear)
        self.up2 = up_mod.Up(512, 256 // f, bilinear)
        self.up3 = up_mod.Up(256, 128 // f, bilinear)
        self.up4 = up_mod.Up(128, 64, bilinear)
        self.outc = up_mod.OutConv(64, mask_classes)
        if uv:
            self.outuv = up_mod.OutConv(64, 2)
        self.resnet_reg = rn_mod.resnet_stn(resnet, None, mask_classes + 3 + (2 if uv else 0))

    def unet(self, x):  # wiring of models/reconstructor.py:138-148
        x1 = self.inc(x)
        x2 = self.down1(x1)
        x3 = self.down2(x2)
        x4 = self.down3(x3)
        xt = self.down4(x4)
        y = self.up1(xt, x4)
        y = self.up2(y, x3)
        y = self.up3(y, x2)
        y = self.up4(y, x1)
        return self.outc(y), xt, y


def _rand(shape, seed, name, lo=0.0, hi=1.0):
    g = synth._rng(seed, name)
    return torch.from_numpy(g.uniform(lo, hi, shape).astype(np.float32))


def _loaded(mod, seed):
    sd = synth.synth_state_dict(mod.state_dict(), seed)
    mod.load_state_dict(sd, strict=True)
    return mod.eval()


def make_block_goldens(up_mod, rn_mod):
    out = {}
    with torch.no_grad():
        # key layouts of the three UNet variants + resnet18/34/50
        layouts = {}
        for tag, kw in [("default", {}), ("bilinear", {"bilinear": True}), ("uv", {"uv": True}),
                        ("resnet18", {"resnet": "resnet18"}), ("resnet50", {"resnet": "resnet50"})]:
            net = _RefNet(up_mod, rn_mod, **kw)
            layouts[tag] = [[k, list(v.shape), str(v.dtype)] for k, v in net.state_dict().items()]
        json.dump(layouts, open(os.path.join(GOLD, "state_dict_layouts.json"), "w"))
        print("default layout keys:", len(layouts["default"]))

        # channel counts are multiples of 64 so the same vectors also drive the MFMA kernels
        m = _loaded(up_mod.DoubleConv(3, 64), 11)
        x = _rand((2, 3, 20, 24), 11, "x")
        out["dc_3_64.x"], out["dc_3_64.y"] = x.numpy(), m(x).numpy()

        m = _loaded(up_mod.DoubleConv(64, 128, 64), 12)
        x = _rand((1, 64, 17, 23), 12, "x", -1, 1)
        out["dc_64_128_m64.x"], out["dc_64_128_m64.y"] = x.numpy(), m(x).numpy()

        m = _loaded(up_mod.Down(64, 128), 13)
        x = _rand((1, 64, 21, 18), 13, "x", -1, 1)
        out["down_64_128.x"], out["down_64_128.y"] = x.numpy(), m(x).numpy()

        m = _loaded(up_mod.Up(128, 64, False), 14)
        x1 = _rand((1, 128, 10, 9), 14, "x1", -1, 1)
        x2 = _rand((1, 64, 21, 19), 14, "x2", -1, 1)
        out["up_128_64.x1"], out["up_128_64.x2"], out["up_128_64.y"] = x1.numpy(), x2.numpy(), m(x1, x2).numpy()

        m = _loaded(up_mod.Up(128, 64, True), 15)
        x1 = _rand((1, 64, 10, 9), 15, "x1", -1, 1)
        x2 = _rand((1, 64, 21, 19), 15, "x2", -1, 1)
        out["upbl_128_64.x1"], out["upbl_128_64.x2"], out["upbl_128_64.y"] = x1.numpy(), x2.numpy(), m(x1, x2).numpy()

        m = _loaded(up_mod.OutConv(64, 4), 16)
        x = _rand((1, 64, 9, 13), 16, "x", -1, 1)
        out["outc_64_4.x"], out["outc_64_4.y"] = x.numpy(), m(x).numpy()

        m = _loaded(rn_mod.resnet_stn("resnet34", None, 7), 17)
        x = _rand((2, 7, 72, 128), 17, "x", -1, 1)
        out["resnet34_7.x"], out["resnet34_7.theta"] = x.numpy(), m(x).numpy()
        m = _loaded(rn_mod.resnet_stn("resnet18", None, 7), 18)
        out["resnet18_7.theta"] = m(x).numpy()
        # Bottleneck depths: resnet50 through the reference factory; the wide variant through
        # _make_resnet (its named constructor does not accept in_channels, models/resnet.py:339-355)
        m = _loaded(rn_mod.resnet_stn("resnet50", None, 7), 19)
        out["resnet50_7.theta"] = m(x).numpy()
        m = _loaded(rn_mod._make_resnet(rn_mod.Bottleneck, [3, 4, 6, 3], None, 7, width_per_group=128), 20)
        out["wide_resnet50_2_7.theta"] = m(x).numpy()

        # whole UNet + STN with the real channel plan on a small odd-sized frame
        net = _loaded(_RefNet(up_mod, rn_mod), 19)
        x = synth.smooth_frames(2, 90, 112, seed=19)
        logits, xt, y = net.unet(x)
        theta = net.resnet_reg(torch.cat((logits, x), 1))
        out["net_90x112.logits"] = logits.numpy()
        out["net_90x112.xtop_mean"] = xt.mean(dim=(2, 3)).numpy()
        out["net_90x112.theta"] = theta.numpy()
    np.savez_compressed(os.path.join(GOLD, "blocks.npz"), **out)
    print("blocks.npz:", {k: v.shape for k, v in out.items()})


def make_train_goldens(up_mod, rn_mod):
    """Training-mode vectors from the reference's own classes: ``module.train()``, forward, then
    ``out.backward(dy)`` with torch autograd - outputs, input/parameter gradients and the running
    statistics after the step (unet/unet_parts.py:7-68, models/resnet.py:36-82)."""
    out = {}

    def run(tag, mod, seed, inputs, only=None):
        m = _loaded(mod, seed).train()
        xs = [t.clone().requires_grad_(True) for t in inputs]
        y = m(*xs)
        dy = _rand(tuple(y.shape), seed, "dy", -1, 1)
        y.backward(dy)
        out[tag + ".y"], out[tag + ".dy"] = y.detach().numpy(), dy.numpy()
        for i, t in enumerate(xs):
            out[f"{tag}.x{i}"], out[f"{tag}.dx{i}"] = inputs[i].numpy(), t.grad.numpy()
        for k, p in m.named_parameters():
            if only is None or k in only:
                out[f"{tag}.grad.{k}"] = p.grad.numpy()
        for k, b in m.named_buffers():
            if k.endswith("running_var") and (only is None or k.replace("running_var", "weight") in only):
                out[f"{tag}.buf.{k}"] = b.numpy().copy()

    run("t_dc_64_128", up_mod.DoubleConv(64, 128), 31, [_rand((2, 64, 13, 18), 31, "x", -1, 1)])
    run("t_down_64_128", up_mod.Down(64, 128), 32, [_rand((2, 64, 14, 19), 32, "x", -1, 1)])
    run("t_up_128_64", up_mod.Up(128, 64, False), 33,
        [_rand((2, 128, 6, 9), 33, "x1", -1, 1), _rand((2, 64, 13, 19), 33, "x2", -1, 1)])
    blk = rn_mod.BasicBlock(64, 128, 2, torch.nn.Sequential(rn_mod.conv1x1(64, 128, 2), torch.nn.BatchNorm2d(128)))
    run("t_basic_64_128_s2", blk, 34, [_rand((2, 64, 13, 18), 34, "x", -1, 1)])
    np.savez_compressed(os.path.join(GOLD, "train_blocks.npz"), **out)
    print("train_blocks.npz:", len(out), "arrays,", sum(v.nbytes for v in out.values()) // 1024, "KiB raw")


def make_full_golden(up_mod, rn_mod):
    """640x360 end-to-end vector: reference classes for UNet/ResNet + oracle warp/CE/POI."""
    torch.set_num_threads(os.cpu_count())
    B = 2
    with torch.no_grad():
        net = _loaded(_RefNet(up_mod, rn_mod), 0)
        x = synth.frames_to_float(synth.synth_frames_u8(B, 360, 640, seed=0))
        logits, _, _ = net.unet(x)
        theta = net.resnet_reg(torch.cat((logits, x), 1))
        court = synth.load_court_template("ncaa_nc4_640x360", 4, B)
        poi = synth.load_court_poi("pitch", B)
        wm = torch_ref.warp(theta, court, (640, 360), nearest=True) * 4
        ce = torch.nn.functional.cross_entropy(logits, wm.long(), reduction="none").mean(dim=(1, 2))
        p = torch_ref.transform_poi(theta, poi)
        am = torch.argmax(logits, 1).to(torch.uint8).numpy()
        top2 = torch.topk(logits, 2, dim=1).values
        margin = (top2[:, 0] - top2[:, 1]).numpy()
    packed = np.packbits(np.unpackbits(am[..., None], axis=-1)[..., 6:].reshape(B, -1), axis=-1)
    np.savez_compressed(
        os.path.join(GOLD, "full_640x360.npz"),
        theta=theta.numpy(), consist=ce.numpy(), poi=p.numpy(),
        argmax_2bit=packed, margin_f16=margin.astype(np.float16),
        logits_sub=logits[:, :, ::8, ::8].numpy(), warp_mask=wm.numpy().astype(np.uint8),
    )
    print("full golden: theta", theta.numpy().reshape(B, 9), "consist", ce.numpy())


def _pack2(am):
    """(B,H,W) uint8 class ids 0..3 -> (B, H*W/4) uint8, four ids per byte (MSB first)."""
    B = am.shape[0]
    return np.packbits(np.unpackbits(am[..., None], axis=-1)[..., 6:].reshape(B, -1), axis=-1)


LINE_FRAMES = (0, 7, 15)                       # frames whose seam lines are stored point by point
LINE_ROWS = (0, 1, 7, 8, 15, 16, 31, 32, -2, -1)   # first / last rows of 8-, 16- and 32-row workgroup tiles, frame borders
LINE_COLS = (0, 1, 7, 8, 15, 16, 31, 32, -2, -1)


def _coverage_vectors(logits):
    """Quantities every logit enters (round 4; the 4::16 sub-sample touches no tile-edge or frame-border pixel):
    the sum of every 8x8 block of every channel of every frame (fp64 -> fp32), and, for three frames, complete rows /
    columns at the first and last lines of the 8 / 16 / 32-pixel workgroup tiles and at the frame borders."""
    B, C, H, W = logits.shape
    assert H % 8 == 0 and W % 8 == 0
    bs = logits.double().reshape(B, C, H // 8, 8, W // 8, 8).sum(dim=(3, 5)).float().numpy()
    fr = [f for f in LINE_FRAMES if f < B]
    rows = [r % H for r in LINE_ROWS]
    cols = [c % W for c in LINE_COLS]
    sel = logits[fr]
    return dict(logits_blocksum8=bs, line_frames=np.array(fr, np.int32), line_rows=np.array(rows, np.int32),
                line_cols=np.array(cols, np.int32), logits_rows=sel[:, :, rows, :].numpy().copy(),
                logits_cols=sel[:, :, :, cols].numpy().copy())


def _save_checked(path, out):
    """np.savez_compressed, after checking that every array the committed file already holds is reproduced
    bit for bit (the vectors are deterministic; a difference means the generator or its inputs moved)."""
    if os.path.exists(path):
        old = np.load(path)
        for k in old.files:
            assert k in out and np.array_equal(old[k], out[k]), f"{os.path.basename(path)}: '{k}' is not reproduced"
    np.savez_compressed(path, **out)


MARGIN_BINS = np.array([0, 1e-5, 2e-5, 5e-5, 1e-4, 2e-4, 5e-4, 1e-3, 2e-3, 5e-3, 1e-2, 1e-1, 1, 1e3], np.float64)
LOW_MARGIN = 1e-2   # pixels whose top-2 logit margin is below this are listed individually


def _predict_golden(up_mod, rn_mod, x, court, poi, wh, chunk, warp_wh=None):
    """Reference classes for UNet / ResNetSTN (eval mode, frames are independent), oracle for the Kornia
    leg: theta, consistency score, POI, arg-max mask, top-2 margins, sub-sampled logits.  warp_wh: (W, H) of the warp
    when it differs from the UNet's (predict.py:151-155): the consistency CE then reads the warp mask through
    F.interpolate(..., mode='nearest') as models/reconstructor.py:230-234 does."""
    W, H = wh
    B = x.shape[0]
    net = _loaded(_RefNet(up_mod, rn_mod), 0)
    logits = torch.empty((B, 4, H, W))
    theta = torch.empty((B, 1, 3, 3))
    for i in range(0, B, chunk):
        lg, _, _ = net.unet(x[i:i + chunk])
        logits[i:i + chunk] = lg
        theta[i:i + chunk] = net.resnet_reg(torch.cat((lg, x[i:i + chunk]), 1))
        print("  frames", i, "...", i + chunk, flush=True)
    wm = torch_ref.warp(theta, court, warp_wh or (W, H), nearest=True) * 4
    wm_ce = wm
    if tuple(wm.shape[1:3]) != (H, W):
        wm_ce = torch.nn.functional.interpolate(wm.unsqueeze(1), size=(H, W), mode="nearest").squeeze(1)
    ce = torch.nn.functional.cross_entropy(logits, wm_ce.long(), reduction="none").mean(dim=(1, 2))
    p = torch_ref.transform_poi(theta, poi)
    am = torch.argmax(logits, 1).to(torch.uint8).numpy()
    top2 = torch.topk(logits, 2, dim=1).values
    margin = (top2[:, 0] - top2[:, 1]).numpy().reshape(B, -1)
    low = np.nonzero(margin < LOW_MARGIN)
    return dict(theta=theta.numpy(), consist=ce.numpy(), poi=p.numpy(), argmax_2bit=_pack2(am),
                low_margin_frame=low[0].astype(np.int16), low_margin_pixel=low[1].astype(np.int32),
                low_margin_value=margin[low].astype(np.float32),
                margin_hist=np.stack([np.histogram(margin[b], MARGIN_BINS)[0] for b in range(B)]).astype(np.int32),
                margin_bins=MARGIN_BINS, logits_sub=logits[:, :, 4::16, 4::16].numpy().copy(),
                warp_mask_2bit=_pack2(wm.numpy().astype(np.uint8)), **_coverage_vectors(logits))


def make_c2_golden(up_mod, rn_mod):
    """BASELINE config 2 at its stated size: 16 frames of 640x360, NCAA template, seed-0 weights and frames
    (= bench.py's model; the first two frames are those of full_640x360.npz)."""
    torch.set_num_threads(os.cpu_count())
    B = 16
    with torch.no_grad():
        x = synth.frames_to_float(synth.synth_frames_u8(B, 360, 640, seed=0))
        court = synth.load_court_template("ncaa_nc4_640x360", 4, B)
        poi = synth.load_court_poi("pitch", B)
        out = _predict_golden(up_mod, rn_mod, x, court, poi, (640, 360), 4)
    _save_checked(os.path.join(GOLD, "c2_640x360_b16.npz"), out)
    print("c2 golden: consist", out["consist"], "margin hist", out["margin_hist"].sum(0))


def make_c2d_golden(up_mod, rn_mod):
    """predict.py's DEFAULT geometry at full size (round 5): the UNet stays at 640x360 while court_size / warp_size are
    raised to out_size = 1280x720 (predict.py:151-155), so the warp mask is 1280x720 and the consistency CE reads it
    through a nearest resize (models/reconstructor.py:226-240).  16 frames, seed-0 weights and frames (the frames,
    logits and theta of the C2 vector), NCAA template resized NEAREST to 1280x720 (utils/dataset.py:51-53), 33-point POI."""
    torch.set_num_threads(os.cpu_count())
    B = 16
    with torch.no_grad():
        x = synth.frames_to_float(synth.synth_frames_u8(B, 360, 640, seed=0))
        court = synth.load_court_template("ncaa_nc4_1280x720", 4, B)
        poi = synth.load_court_poi("pitch", B)
        out = _predict_golden(up_mod, rn_mod, x, court, poi, (640, 360), 4, warp_wh=(1280, 720))
    c2 = np.load(os.path.join(GOLD, "c2_640x360_b16.npz"))
    assert np.array_equal(c2["theta"], out["theta"]) and np.array_equal(c2["logits_blocksum8"], out["logits_blocksum8"])
    keep = ("theta", "consist", "poi", "warp_mask_2bit")      # the logit vectors are those of c2_640x360_b16.npz
    _save_checked(os.path.join(GOLD, "c2d_unet640x360_warp1280x720_b16.npz"), {k: out[k] for k in keep})
    print("c2d golden: consist", out["consist"])


def make_c5_golden(up_mod, rn_mod):
    """BASELINE config 5 at its stated batch: 16 frames of 1280x720, 4-class pitch template
    (pitch_mask_v3_nc4_hd), 33-point POI (predict.py:151-155,186-192).  About ten minutes on 8 host cores."""
    torch.set_num_threads(os.cpu_count())
    B = 16
    with torch.no_grad():
        x = synth.frames_to_float(synth.synth_frames_u8(B, 720, 1280, seed=0))
        court = synth.load_court_template("pitch_v3_nc4_1280x720", 4, B)
        poi = synth.load_court_poi("pitch", B)
        out = _predict_golden(up_mod, rn_mod, x, court, poi, (1280, 720), 1)
    _save_checked(os.path.join(GOLD, "c5_1280x720_b16.npz"), out)
    print("c5 golden: theta", out["theta"].reshape(B, 9)[:2], "consist", out["consist"])


def make_c3_b16_golden(up_mod, rn_mod):
    """BASELINE config 3 at its stated batch (train.py:155-225: batch 16, batch-statistics BatchNorm depends on
    it): the reference's own classes under ``net.train()``, FORWARD ONLY (fp32, no_grad - the backward of 16
    frames does not fit this container), losses of train.py:181-224 through oracle/train_ref.losses.  Stored:
    the loss values, theta, sub-sampled logits, and every BatchNorm layer's running_mean / running_var /
    num_batches_tracked AFTER the step (momentum 0.1, unbiased variance)."""
    from oracle import train_ref
    torch.set_num_threads(os.cpu_count())
    B, H, W = 16, 360, 640
    x = synth.frames_to_float(synth.synth_frames_u8(B, H, W, seed=0))
    court = synth.load_court_template("ncaa_nc4_640x360", 4, B)
    poi = synth.load_court_poi("pitch", B)
    batch = c3_batch(B, H, W, poi.shape[1])
    net = _loaded(_RefNet(up_mod, rn_mod), 0).train()
    with torch.no_grad():
        logits, _, _ = net.unet(x)
        theta = net.resnet_reg(torch.cat((logits, x), 1))
        preds = {"logits": logits, "theta": theta, "poi": torch_ref.transform_poi(theta, poi),
                 "warp_mask": torch_ref.warp(theta, court, (W, H), nearest=False)}
        ls = train_ref.losses(preds, batch)
    out = {"theta": theta.numpy(), "logits_sub": logits[:, :, 4::16, 4::16].numpy().copy(),
           "poi": preds["poi"].numpy(), "warp_mask_sub": preds["warp_mask"][:, 4::16, 4::16].numpy().copy()}
    for k, v in ls.items():
        out[f"loss.{k}"] = np.float64(float(v))
    names = []
    for k, b in net.named_buffers():
        if k.endswith(("running_mean", "running_var", "num_batches_tracked")):
            names.append(k)
            out[f"buf.{k}"] = b.numpy().copy()
    out["buffers"] = np.array(names)
    out.update(_coverage_vectors(logits))
    _save_checked(os.path.join(GOLD, "c3_fwd_640x360_b16.npz"), out)
    print("c3 b16 golden:", {k: float(v) for k, v in ls.items()}, len(names), "BatchNorm buffers")


def make_c3_b16_grad_golden(up_mod, rn_mod):
    """BASELINE config 3 at its stated batch, WITH the backward pass (round 4): the reference's own classes under
    ``net.train()`` + torch autograd in fp32 on all 16 frames (about 31 GB of host memory, eight minutes on 8 cores; an fp64
    run of this size does not fit the container - the fp32-vs-fp64 yardstick stays the B=2 vector), losses of
    train.py:181-224, then one optimizer step as train.py:88,234-237 does it: ``clip_grad_value_(0.1)`` and
    ``RMSprop(lr=1e-5, weight_decay=1e-8, momentum=0.9)``.  Stored per parameter: a fixed sample of the gradient, its L2 norm,
    and the same sample of the weight update (w_after - w_before)."""
    from oracle import train_ref
    torch.set_num_threads(os.cpu_count())
    B, H, W = 16, 360, 640
    x = synth.frames_to_float(synth.synth_frames_u8(B, H, W, seed=0))
    court = synth.load_court_template("ncaa_nc4_640x360", 4, B)
    poi = synth.load_court_poi("pitch", B)
    batch = c3_batch(B, H, W, poi.shape[1])
    net = _loaded(_RefNet(up_mod, rn_mod), 0).train()
    opt = torch.optim.RMSprop(net.parameters(), lr=1e-5, weight_decay=1e-8, momentum=0.9)
    logits, _, _ = net.unet(x)
    theta = net.resnet_reg(torch.cat((logits, x), 1))
    preds = {"logits": logits, "theta": theta, "poi": torch_ref.transform_poi(theta, poi),
             "warp_mask": torch_ref.warp(theta, court, (W, H), nearest=False)}
    ls = train_ref.losses(preds, batch)
    print("losses", {k: float(v) for k, v in ls.items()}, flush=True)
    opt.zero_grad()
    ls["total"].backward()
    out = {f"loss.{k}": np.float64(float(v.detach())) for k, v in ls.items()}
    names = [k for k, _ in net.named_parameters()]
    out["names"] = np.array(names)
    before = {k: p.detach().clone() for k, p in net.named_parameters()}
    for i, (k, p) in enumerate(net.named_parameters()):
        g = p.grad.detach().reshape(-1)
        idx = torch.from_numpy(grad_sample_index(g.numel()))
        out[f"g.{i}"] = g[idx].numpy().copy()
        out[f"gnorm.{i}"] = np.float64(float(g.double().norm()))
    torch.nn.utils.clip_grad_value_(net.parameters(), 0.1)
    opt.step()
    for i, (k, p) in enumerate(net.named_parameters()):
        d = (p.detach() - before[k]).reshape(-1)
        idx = torch.from_numpy(grad_sample_index(d.numel()))
        out[f"dw.{i}"] = d[idx].numpy().copy()
    np.savez_compressed(os.path.join(GOLD, "c3_grads_640x360_b16.npz"), **out)
    print("c3 b16 gradient golden:", len(names), "parameters,", sum(v.nbytes for v in out.values()) // 1024, "KiB raw")


def make_c3_golden(up_mod, rn_mod):
    """BASELINE config 3 at 640x360 (B=2 of the 16): the reference's own classes under ``net.train()``
    (batch-statistics BatchNorm) + autograd, losses of train.py:181-224 (oracle/train_ref.losses), once in
    fp32 and once in fp64.  Stored: loss values, theta, and for every parameter a fixed sample of the fp64
    gradient plus the fp32 run's relative error against it (the yardstick for the GPU's error)."""
    from oracle import train_ref
    torch.set_num_threads(os.cpu_count())
    B, H, W = 2, 360, 640
    x = synth.frames_to_float(synth.synth_frames_u8(B, H, W, seed=0))
    court = synth.load_court_template("ncaa_nc4_640x360", 4, B)
    poi = synth.load_court_poi("pitch", B)
    batch = c3_batch(B, H, W, poi.shape[1])
    res = {}
    for tag, dt in (("f32", torch.float32), ("f64", torch.float64)):
        net = _loaded(_RefNet(up_mod, rn_mod), 0).to(dt).train()
        xx = x.to(dt)
        logits, _, _ = net.unet(xx)
        theta = net.resnet_reg(torch.cat((logits, xx), 1))
        preds = {"logits": logits, "theta": theta,
                 "poi": torch_ref.transform_poi(theta, poi.to(dt)),
                 "warp_mask": torch_ref.warp(theta, court.to(dt), (W, H), nearest=False)}
        bt = {k: (v.to(dt) if v.is_floating_point() else v) for k, v in batch.items()}
        ls = train_ref.losses(preds, bt)
        ls["total"].backward()
        res[tag] = ({k: float(v.detach()) for k, v in ls.items()}, theta.detach().double().numpy(),
                    {k: p.grad.detach().double().numpy().ravel() for k, p in net.named_parameters()},
                    logits.detach()[:, :, 4::16, 4::16].double().numpy())
        print(tag, res[tag][0], flush=True)
    out = {"theta_f32": res["f32"][1], "theta_f64": res["f64"][1],
           "logits_sub_f32": res["f32"][3].astype(np.float32), "logits_sub_f64": res["f64"][3]}
    for k in res["f32"][0]:
        out[f"loss_f32.{k}"] = np.float64(res["f32"][0][k])
        out[f"loss_f64.{k}"] = np.float64(res["f64"][0][k])
    names = list(res["f64"][2])
    out["names"] = np.array(names)
    for i, k in enumerate(names):
        g64, g32 = res["f64"][2][k], res["f32"][2][k]
        idx = grad_sample_index(g64.size)
        n64 = np.linalg.norm(g64)
        out[f"g64.{i}"] = g64[idx]
        out[f"stat.{i}"] = np.array([n64, np.linalg.norm(g32 - g64) / max(n64, 1e-300),
                                     np.linalg.norm(g32[idx] - g64[idx]) / max(np.linalg.norm(g64[idx]), 1e-300)])
    np.savez_compressed(os.path.join(GOLD, "c3_train_640x360_b2.npz"), **out)
    st = np.array([out[f"stat.{i}"] for i in range(len(names))])
    print("c3 golden: fp32-vs-fp64 per-tensor relative error: median %.2e  p90 %.2e  max %.2e" %
          (np.median(st[:, 1]), np.quantile(st[:, 1], 0.9), st[:, 1].max()))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--full", action="store_true")
    ap.add_argument("--data-only", action="store_true")
    ap.add_argument("--train-only", action="store_true", help="only tests/golden/train_blocks.npz")
    ap.add_argument("--configs", default="", help="comma list of c2,c5,c3,c3b16: only the full-size BASELINE-config vectors")
    a = ap.parse_args()
    os.makedirs(GOLD, exist_ok=True)
    make_data()
    if not a.data_only:
        up_mod = _load("ref_unet_parts", "unet/unet_parts.py")
        rn_mod = _load("ref_resnet", "models/resnet.py")
        if a.configs:
            for c in a.configs.split(","):
                {"c2": make_c2_golden, "c2d": make_c2d_golden, "c5": make_c5_golden, "c3": make_c3_golden, "c3b16": make_c3_b16_golden,
                 "c3b16grad": make_c3_b16_grad_golden}[c](up_mod, rn_mod)
            raise SystemExit(0)
        if a.train_only:
            make_train_goldens(up_mod, rn_mod)
            raise SystemExit(0)
        make_block_goldens(up_mod, rn_mod)
        make_train_goldens(up_mod, rn_mod)
        if a.full:
            make_full_golden(up_mod, rn_mod)
