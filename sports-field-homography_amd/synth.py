"""Deterministic synthetic inputs for tests and benchmarks.

Everything here is a pure function of (seed, name, shape): a counter-based Philox
stream keyed per tensor, independent of torch's RNG and of the order of generation,
so the CPU oracle, the golden-fixture generator and the GPU runs see identical bits.

* weights: He-style normal conv weights (activations stay O(1) through the stack),
  BatchNorm running_mean ~ U(-0.1, 0.1), running_var ~ U(0.5, 1.5), gamma ~ U(0.5, 1.5),
  beta ~ U(-0.1, 0.1); ``resnet_reg.reg.weight`` ~ N(0, 1e-4) on top of the identity bias
  so that theta != I (the reference's random init gives exactly the identity for every
  input: models/resnet.py:206-208).
* frames: uint8 RGB uniform {0..255}, converted like the reference's dataset
  (utils/dataset.py:154-159): /255 -> float32, HWC -> CHW.
"""
import os
import zlib

import numpy as np
import torch

_DATA_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")


def _rng(seed, name):
    key = (int(seed) & 0xFFFFFFFF) | ((zlib.crc32(name.encode()) & 0xFFFFFFFF) << 32)
    return np.random.Generator(np.random.Philox(key=key))


def _fill(seed, name, shape):
    g = _rng(seed, name)
    shape = tuple(shape)
    leaf = name.rsplit(".", 1)[-1]
    if leaf == "num_batches_tracked":
        return np.asarray(100, dtype=np.int64)
    if leaf == "running_mean":
        return g.uniform(-0.1, 0.1, shape).astype(np.float32)
    if leaf == "running_var":
        return g.uniform(0.5, 1.5, shape).astype(np.float32)
    if len(shape) == 1:
        if leaf == "weight":  # BatchNorm gamma
            if ("resnet_reg" in name or name.startswith("layer")) and ".bn3." in name:
                return g.uniform(0.1, 0.3, shape).astype(np.float32)  # Bottleneck's last BN
            if "resnet_reg" in name and ".bn2." in name:
                # small residual-branch gain keeps the 16 un-normalised residual adds bounded
                return g.uniform(0.1, 0.3, shape).astype(np.float32)
            return g.uniform(0.5, 1.5, shape).astype(np.float32)
        if name.endswith("reg.bias"):
            eye = np.array([1, 0, 0, 0, 1, 0, 0, 0, 1], np.float32)
            return (eye + g.normal(0.0, 0.02, shape)).astype(np.float32)
        return g.uniform(-0.1, 0.1, shape).astype(np.float32)  # conv / BN bias
    if len(shape) == 2:  # reg.weight (9, 512)
        return g.normal(0.0, 1e-4, shape).astype(np.float32)
    if len(shape) == 4:
        if ".up." in name or name.endswith("up.weight"):  # ConvTranspose2d (Cin, Cout, 2, 2)
            fan_in = shape[0]
        else:  # Conv2d (Cout, Cin, kh, kw)
            fan_in = shape[1] * shape[2] * shape[3]
        return g.normal(0.0, np.sqrt(2.0 / fan_in), shape).astype(np.float32)
    raise ValueError(f"no synthetic rule for {name} {shape}")


def _conv_bn_pairs(names):
    """[(conv weight key, conv bias key or None, BatchNorm prefix)] for every conv that a BatchNorm follows, found by
    the checkpoint's own naming (unet/unet_parts.py:14-21: double_conv.{0,3} -> .{1,4}; models/resnet.py: convN -> bnN,
    conv0 -> bn1, downsample.0 -> downsample.1)"""
    have = set(names)
    pairs = []
    for k in names:
        if not k.endswith(".weight"):
            continue
        stem = k[:-len(".weight")]
        head, _, leaf = stem.rpartition(".")
        cand = None
        if leaf.isdigit():                       # Sequential index: BatchNorm is the next module
            cand = f"{head}.{int(leaf) + 1}"
        elif leaf == "conv0":
            cand = f"{head}.bn1"
        elif leaf.startswith("conv") and leaf[4:].isdigit():
            cand = f"{head}.bn{leaf[4:]}"
        if cand and f"{cand}.running_var" in have:
            pairs.append((k, f"{stem}.bias" if f"{stem}.bias" in have else None, cand))
    return pairs


def trained_like_state_dict(template_sd, seed=0, scaled_layer_exp=None, return_info=False, head_sigma=0.02):
    """A second weight family with the statistics of a TRAINED checkpoint, where `synth_state_dict` has those of a fresh
    one: the fp16 per-tensor-exponent arithmetic is exactly what such statistics stress.

    * every BatchNorm: running_var log-uniform over 1e-3 .. 1e3 (six decades across the channels of one layer), gamma ~
      N(0, 1) - so negative - with 5 % of the channels EXACTLY zero (dead), beta ~ N(0, 0.3);
    * every conv in front of one: Gaussian bulk + 1 % outliers at 30 - 50 sigma with random signs (heavy tail), then
      each output channel rescaled so that the conv's output has about the variance its BatchNorm recorded - what
      training makes true (running_var IS the conv output's variance; without this coupling the activations would
      grow by E[gamma^2 / var]^(1/2) = 8.5x per layer, which no checkpoint that was ever evaluated does) - and
      running_mean ~ N(0, 0.3 sqrt(var));
    * ONE UNet layer (picked by the seed) with its conv weights and bias scaled by 2^scaled_layer_exp (default: +12 for
      even seeds, -12 for odd ones) and its BatchNorm statistics scaled to match: computed through a tensor 4096x larger /
      smaller than its neighbours (+12: the same function; -12: the variances fall to 6e-11 .. 6e-5, BatchNorm's eps = 1e-5
      takes over and most of the layer's channels go quiet - a collapsed layer, as checkpoints have them);
    * residual-branch BatchNorms of the ResNet (bn2 / bn3 / downsample) keep a gain of N(0, 0.3) so that the 16
      un-normalised residual adds stay bounded;
    * the regression head (`resnet_reg.reg.weight`) ~ N(0, head_sigma = 0.02) instead of `synth_state_dict`'s N(0, 1e-4): theta
      then depends on the 512 pooled features at full strength (entries move by O(0.1 .. 1) between frames, like the trained
      homographies of utils/mapping_example.py), so an error in the ResNet-STN's features shows in theta instead of being
      scaled away by a near-zero head.

    unet/unet_parts.py:14-21 (BatchNorm after every conv), models/resnet.py:64-82."""
    names = list(template_sd.keys())
    out = {n: torch.from_numpy(np.ascontiguousarray(_fill(seed, n, tuple(t.shape)))).to(t.dtype).reshape(t.shape)
           for n, t in template_sd.items()}
    pairs = _conv_bn_pairs(names)
    unet_pairs = [p for p in pairs if not p[0].startswith("resnet_reg")]
    pick = unet_pairs[_rng(seed, "scaled-layer").integers(0, len(unet_pairs))][0] if unet_pairs else None
    if scaled_layer_exp is None:
        scaled_layer_exp = 12 if seed % 2 == 0 else -12
    for wk, bk, bn in pairs:
        g = _rng(seed, "trained:" + wk)
        w = out[wk].numpy().astype(np.float64)
        cout = w.shape[0]
        sigma = w.std()
        hot = g.random(w.shape) < 0.01
        w = np.where(hot, g.uniform(30.0, 50.0, w.shape) * sigma * g.choice([-1.0, 1.0], w.shape), w)
        var = np.exp(g.uniform(np.log(1e-3), np.log(1e3), cout))
        # conv output variance for inputs of power 1/2 per channel (ReLU of a unit Gaussian; frames: 1/3): sum of w^2 / 2
        pred = 0.5 * (w.reshape(cout, -1) ** 2).sum(1)
        w *= np.sqrt(var / np.maximum(pred, 1e-30)).reshape((cout,) + (1,) * (w.ndim - 1))
        mean = g.normal(0.0, 0.3, cout) * np.sqrt(var)
        residual = wk.startswith("resnet_reg") and (".bn2" in bn or ".bn3" in bn or "downsample" in bn)
        gamma = g.normal(0.0, 0.3 if residual else 1.0, cout)
        gamma[g.random(cout) < 0.05] = 0.0
        beta = g.normal(0.0, 0.3, cout)
        bias = g.normal(0.0, 0.1, cout) * np.sqrt(var) if bk else None
        if wk == pick:
            f = 2.0 ** scaled_layer_exp
            w, var, mean = w * f, var * f * f, mean * f
            bias = bias * f if bias is not None else None
        out[wk] = torch.from_numpy(w.astype(np.float32))
        if bk:
            out[bk] = torch.from_numpy(bias.astype(np.float32))
        out[bn + ".running_var"] = torch.from_numpy(var.astype(np.float32))
        out[bn + ".running_mean"] = torch.from_numpy(mean.astype(np.float32))
        out[bn + ".weight"] = torch.from_numpy(gamma.astype(np.float32))
        out[bn + ".bias"] = torch.from_numpy(beta.astype(np.float32))
    for k in names:
        if k.endswith("reg.weight") and out[k].dim() == 2:
            out[k] = torch.from_numpy(_rng(seed, "trained:" + k).normal(0.0, head_sigma, tuple(out[k].shape)).astype(np.float32))
    if return_info:
        return out, {"scaled_layer": pick, "scaled_layer_exp": scaled_layer_exp, "conv_bn_pairs": len(pairs)}
    return out


def synth_state_dict(template_sd, seed=0):
    """Return a state_dict with the keys/shapes of ``template_sd`` filled deterministically."""
    out = {}
    for name, t in template_sd.items():
        arr = _fill(seed, name, tuple(t.shape))
        out[name] = torch.from_numpy(np.ascontiguousarray(arr)).to(t.dtype).reshape(t.shape)
    return out


def synth_frames_u8(batch, height, width, seed=0):
    """uint8 HWC frames, shape (B, H, W, 3); batch k of a run uses seed k."""
    g = _rng(seed, f"frames{height}x{width}")
    return g.integers(0, 256, size=(batch, height, width, 3), dtype=np.uint8)


def frames_to_float(frames_u8):
    """uint8 (B,H,W,3) -> float32 (B,3,H,W) in [0,1]; reference: utils/dataset.py:154-159."""
    x = torch.from_numpy(np.ascontiguousarray(frames_u8)).to(torch.float32) / 255.0
    return x.permute(0, 3, 1, 2).contiguous()


def smooth_frames(batch, height, width, seed=0):
    """Low-frequency synthetic frames (float32 CHW in [0,1]) - gives logits with spatial
    structure, used by parity tests next to the white-noise frames of the benchmark."""
    g = _rng(seed, f"smooth{height}x{width}")
    yy, xx = np.meshgrid(np.linspace(0, 1, height, dtype=np.float32),
                         np.linspace(0, 1, width, dtype=np.float32), indexing="ij")
    out = np.empty((batch, 3, height, width), np.float32)
    for b in range(batch):
        for c in range(3):
            acc = np.zeros((height, width), np.float32)
            for _ in range(4):
                fx, fy, ph = g.uniform(0.5, 6.0), g.uniform(0.5, 6.0), g.uniform(0, 6.28)
                acc += np.sin(6.2831853 * (fx * xx + fy * yy) + ph).astype(np.float32)
            out[b, c] = 0.5 + 0.125 * acc
    out += g.uniform(-0.03, 0.03, out.shape).astype(np.float32)
    return torch.from_numpy(np.clip(out, 0.0, 1.0))


def load_court_template(name="ncaa_nc4_640x360", num_classes=4, batch_size=1):
    """Class-id court template as float32 (B,1,H,W) valued k/num_classes.

    Same tensor convention as the reference's ``open_court_template``
    (utils/dataset.py:47-61).  The id images under ``data/`` were derived once from the
    reference's assets by ``oracle/make_fixtures.py`` (NEAREST resize, utils/dataset.py:51-53).
    """
    ids = np.load(os.path.join(_DATA_DIR, f"court_ids_{name}.npy"))
    t = torch.from_numpy(ids.astype(np.float32) / float(num_classes))
    return t[None, None].repeat(batch_size, 1, 1, 1)


def load_court_poi(name="pitch", batch_size=1):
    """Court points of interest, float32 (B,N,2) in [-1,1]; reference convention:
    ``open_court_poi`` (utils/dataset.py:63-96)."""
    pts = np.load(os.path.join(_DATA_DIR, f"court_poi_{name}.npy")).astype(np.float32)
    return torch.from_numpy(pts)[None].repeat(batch_size, 1, 1)


# Two homographies predicted by the reference authors' trained model, printed as
# numeric literals in utils/mapping_example.py:12-22 and :48-58 (frame -> court,
# normalised coordinates).  Used as realistic warp test inputs.
REALISTIC_THETAS = np.array(
    [
        [[8.030766487121582, -0.22687992453575134, 9.891857147216797],
         [3.553352117538452, 25.72734260559082, -0.09768841415643692],
         [0.1463453769683838, 5.179210662841797, 16.56546974182129]],
        [[5.78266048, -0.43701401, 8.0031395],
         [3.63819695, 15.77359295, -0.46604609],
         [0.14406031, 3.68673325, 13.25017166]],
    ],
    dtype=np.float32,
)
