"""Training-mode forward and backward of the hot path on the HIP kernels.

What ``net.train(); preds = net(imgs); loss.backward()`` does in the reference (train.py:151,170,
233): batch-statistics BatchNorm (running stats updated with momentum 0.1), activations kept for
the backward pass, and gradients for every parameter.  The losses, ``clip_grad_value_`` and the
optimizer stay with the caller (train.py:181-237), exactly as in the reference.

Activations are kept in fp32 NHWC.  Forward, backward-data and 3x3 backward-filter convolutions run on the
split-operand kernels (``sfh_conv_s3_fwd`` / ``sfh_conv_wgrad_s3``); the split copies of activations and
pre-activation gradients are written by the BatchNorm kernels in the same pass.  ``SFH_TRAIN_PRECISION``:
``f16x3`` (default) - two fp16 planes per operand, three MFMA products; the backward pass is carried at a power of
two chosen from the largest output gradient of the step, so that the gradients (~1e-7 for losses that are means over
B*H*W pixels) fit fp16's exponent range, and a step whose activations or gradients still leave it is repeated with
bf16x6 from the same BatchNorm statistics (TrainStep; under the autograd node a gradient overflow raises
FP16RangeError); ``bf16x6`` - three bf16 planes, six products; ``fp32`` - fp32 MFMA kernels throughout.  The first
layer, the stem and the other backward-filter shapes (``sfh_conv_wgrad``) are fp32 MFMA in every mode.  A tape of
closures records the backward of each layer; gradients of activations are keyed by tensor identity and accumulated
with ``sfh_slice_add``.  PyTorch provides memory, streams and the autograd hook (``torch.autograd.Function``) only.
"""
import ctypes
import os

import torch

from . import _lib
from . import engine as E
from .engine import PackedConv, _ptr, _stream

BN_MOMENTUM = 0.1  # nn.BatchNorm2d default, unchanged by the reference

# Test hook: set to a dict and the next training pass leaves references to the ResNet blocks' activations
# ("layer4.1": {"in", "t", "out"} NHWC tensors) and the theta gradient it starts its backward pass from ("dtheta").
# tests/test_gpu_configs.py uses it to re-derive the gradients of the last ResNet stage in fp64 under the ReLU decisions
# this pass took.  None (the default): nothing is kept.
CAPTURE = None


def _train_fmt():
    """split format of the training convs: "h2" (f16x3, default), "s3" (bf16x6) or None (fp32 MFMA)"""
    p = os.environ.get("SFH_TRAIN_PRECISION", "f16x3")
    if p not in E.PRECISIONS:
        raise ValueError(f"SFH_TRAIN_PRECISION={p!r}: expected one of {sorted(E.PRECISIONS)}")
    return E.PRECISIONS[p]


def _empty(shape, like, dtype=torch.float32):
    return torch.empty(shape, dtype=dtype, device=like.device)


def _zeros(shape, like, dtype=torch.float32):
    return torch.zeros(shape, dtype=dtype, device=like.device)


class FP16RangeError(RuntimeError):
    """an activation or gradient of a training step did not fit the two-plane fp16 (H2) format"""


class _MultiCopy:
    """dst_i (contiguous) = src_i (up to 4-D, any strides, same number of elements) [* scale] for a fixed list of
    destinations in ONE launch of sfh_multi_copy (torch issues one copy per tensor: 344 of the ~1580 launches of a
    training step were the gradient assembly and the BatchNorm snapshot).  4-byte element types; int64 tensors go as
    pairs of words (contiguous only)."""
    CHUNK = 65536

    def __init__(self, dsts):
        import numpy as np
        self.dsts = list(dsts)
        self.dev = self.dsts[0].device if self.dsts else None
        self.words = [d.numel() * (d.element_size() // 4) for d in self.dsts]
        if any(d.element_size() not in (4, 8) or not d.is_contiguous() for d in self.dsts):
            raise ValueError("_MultiCopy: destinations must be contiguous tensors of 4- or 8-byte elements")
        chunks = [(i, min(self.CHUNK, n - off), off) for i, n in enumerate(self.words) for off in range(0, n, self.CHUNK)]
        ch = np.zeros(len(chunks), dtype=np.dtype([("t", "<i4"), ("c", "<i4"), ("o", "<i8")]))
        for j, c in enumerate(chunks):
            ch[j] = c
        self.nchunks = len(chunks)
        self.chunks = torch.from_numpy(ch.view(np.uint8).reshape(-1).copy()).to(self.dev) if chunks else None
        self._tab = np.zeros((len(self.dsts), 8), dtype=np.int64)
        self._key = None
        self._dev_tab = None

    def run(self, srcs, scale=1.0, scale_dev=None):
        """scale_dev: a one-element float32 device tensor holding the factor (sfh_multi_copy_dscale) instead of `scale`"""
        if not self.dsts:
            return
        tab = self._tab
        key = []
        for i, (d, s) in enumerate(zip(self.dsts, srcs)):
            if s.numel() != d.numel() or s.element_size() != d.element_size() or s.device != d.device:
                raise ValueError("_MultiCopy: source / destination mismatch")
            if d.element_size() == 8 or s.dim() > 4:
                if not s.is_contiguous():
                    raise ValueError("_MultiCopy: 8-byte or >4-D sources must be contiguous")
                shape, strides = (1, 1, 1, self.words[i]), (0, 0, 0, 1)
            else:
                pad = 4 - s.dim()
                shape = (1,) * pad + tuple(s.shape)
                strides = (0,) * pad + tuple(s.stride())
            tab[i, 0], tab[i, 1] = d.data_ptr(), s.data_ptr()
            tab[i, 2] = shape[1] | (shape[2] << 32)
            tab[i, 3] = shape[3]
            tab[i, 4:8] = strides
            key.append((d.data_ptr(), s.data_ptr()))
        if self._dev_tab is None or key != self._key:      # static pairs (the BatchNorm snapshot): uploaded once
            self._dev_tab = torch.from_numpy(tab.view("uint8").reshape(-1).copy()).to(self.dev)
            self._key = key
        if scale_dev is not None:
            _lib.check(_lib.load().sfh_multi_copy_dscale(_ptr(self._dev_tab), _ptr(self.chunks), self.nchunks, _ptr(scale_dev),
                                                         _stream()), "multi_copy_dscale")
            return
        _lib.check(_lib.load().sfh_multi_copy(_ptr(self._dev_tab), _ptr(self.chunks), self.nchunks, float(scale), _stream()),
                   "multi_copy")


class _BNSnapshot:
    """Copies of a model's buffers (BatchNorm running statistics), so that a training step that has to be repeated
    in another precision starts from the same statistics: one multi-tensor copy per save / restore.  The buffer objects
    are looked up again at every save(): net.to() / .float() / load_state_dict(assign=True) REPLACE them, and a restore
    into tensors the model no longer holds would leave the statistics advanced twice.  `generation` counts the
    saves: a consumer (the autograd node's backward) can tell whether the copy is still the one it made."""

    def __init__(self, net):
        self.net = net
        self.bufs, self.snap = [], []
        self._save = self._restore = None
        self.generation = 0

    def save(self):
        bufs = [b for b in self.net.buffers() if b.is_cuda]
        if len(bufs) != len(self.bufs) or any(a is not b or a.shape != c.shape or a.dtype != c.dtype
                                              for a, b, c in zip(bufs, self.bufs, self.snap)):
            self.bufs = bufs
            self.snap = [torch.empty_like(b, memory_format=torch.contiguous_format) for b in bufs]
            self._save, self._restore = _MultiCopy(self.snap), None
        if self.bufs:
            self._save.run(self.bufs)
        self.generation += 1
        return self.generation

    def restore(self):
        if self.bufs:
            if self._restore is None or any(a is not b for a, b in zip(self._restore.dsts, self.bufs)):
                self._restore = _MultiCopy(self.bufs)
            self._restore.run(self.snap)


_RANGE_WARNED = []


def _warn_range_fallback():
    if not _RANGE_WARNED:
        _RANGE_WARNED.append(1)
        import warnings
        warnings.warn("sfh_amd training: a value left the fp16 range of SFH_TRAIN_PRECISION=f16x3; the step was repeated "
                      "with bf16x6 (set SFH_TRAIN_PRECISION=bf16x6 to avoid the double work)")


class Tape:
    """Backward closures in forward order + gradients of activations by tensor identity."""

    def __init__(self, fmt="env"):
        self.ops = []
        self.grads = {}
        self.param_grads = {}
        self.lib = _lib.load()
        self.fmt = _train_fmt() if fmt == "env" else fmt            # "s3" | "h2" | None
        self.use_s3 = self.fmt is not None
        self.fmt_code = {"s3": _lib.FMT_S3, "h2": _lib.FMT_H2}.get(self.fmt, _lib.FMT_F32)
        self._s3 = {}
        self.no_f32 = set()   # ids of activation handles without fp32 storage (conv_bn_act(f32_out=False))
        # BatchNorm + ReLU layers whose output has ONE consumer (the next conv): id(y) -> {z, mi, bn}; that conv's
        # backward-data launch leaves the layer's backward sums in entry["table"] (sfh_conv_desc.bwd_z)
        self.single_consumer = {}
        # id(y) -> fp64 [sum g | sum g * xhat] of a BatchNorm + ReLU layer whose total gradient is complete and whose sums
        # were taken by the pass that completed it (pool_backward_fused); a later add_grad to y is an ordering bug
        self.bwd_sums = {}
        # id(y) -> {z, mi, bn} of every BatchNorm + ReLU layer without a residual (consumers that complete y's gradient use it to
        # take the backward sums on the way: out_conv)
        self.bn_layers = {}
        # f16x3 (H2 copies of activations and gradients): fp16's exponent range has to hold them.
        #   overflow - device word the kernels raise when a value does not fit (checked at the end of the backward pass);
        #   gscale   - power of two all gradients are carried with (the losses are means over B*H*W pixels, their
        #              gradients ~1e-7: far below fp16's range); chosen by run_backward from the largest seed gradient
        #              (one read-back per step) and divided out of the parameter gradients at its end;
        #   wexp     - per conv weight: exponent of its H2 planes, from ONE batched max over the parameters per step
        self.overflow = None
        self.gscale = 1.0
        self.wexp = {}
        self.order = E.LaunchOrder()
        self._arena = {}      # dtype -> [zero-filled buffer, next free element]

    # Zero-initialised accumulators (fp64 BatchNorm / reduction sums, fp32 split-K weight gradients): a training
    # step needs ~240 of them; carving them out of two buffers zeroed by ONE fill each replaces that many
    # fill launches.  Chunk sizes: 1 Mi doubles, 64 Mi floats (the weight gradients of the default model are
    # 52 M floats); a request that does not fit opens another chunk.
    _ARENA_CHUNK = {torch.float64: 1 << 20, torch.float32: 1 << 26}

    def zeros(self, shape, like, dtype=torch.float32):
        n = 1
        for d in shape:
            n *= int(d)
        cur = self._arena.get(dtype)
        if cur is None or cur[1] + n > cur[0].numel():
            cur = self._arena[dtype] = [torch.zeros(max(n, self._ARENA_CHUNK[dtype]), dtype=dtype, device=like.device), 0]
        out = cur[0][cur[1]:cur[1] + n].view(shape)
        cur[1] += (n + 63) & ~63          # 256- / 512-byte aligned slices
        return out

    def s3(self, t):
        """split copy (S3 or H2, the tape's format) of an NHWC activation (converted once, kept while the tape lives)"""
        v = self._s3.get(id(t))
        if v is None:
            v = self._s3[id(t)] = (t, E.f32_to_split(t, self.fmt, self.overflow))
        return v[1]

    def f32(self, t):
        """fp32 NHWC values of an activation of this pass, also when only its split copy was written (a handle)"""
        return E.s3_to_f32(self._s3[id(t)][1]) if id(t) in self.no_f32 else t

    def prepare_h2(self, net, n_pixels):
        """H2 mode: gradient scale for n_pixels = B*H*W and the weight exponents of all conv weights (one sync)."""
        if self.fmt != "h2":
            return
        import math
        dev = next(net.parameters()).device
        # two words: [0] raised by the kernels; [1] = a copy of [0] taken behind the forward pass (mark_forward_done), so that
        # the one read-back at the end of the step can tell a forward overflow from a backward one
        self.overflow = E.filled((2,), torch.int32, dev)
        ws = [p for p in net.parameters() if p.dim() == 4]
        # one launch over all conv weights + one read-back (sfh_multi_absminmax)
        for w, (m, _) in zip(ws, E.absminmax([w.detach() for w in ws])):
            if not math.isfinite(m):
                raise ValueError("a conv weight holds non-finite values")
            self.wexp[id(w)] = (14 - math.frexp(m)[1]) if m > 0 else 0

    def mark_forward_done(self):
        """H2: remember (on the device, no synchronisation) whether anything overflowed up to here"""
        if self.overflow is not None and self.overflow.numel() > 1:
            _lib.check(self.lib.sfh_copy2d_words(_ptr(self.overflow), 1, ctypes.c_void_p(self.overflow.data_ptr() + 4), 1, 1, 1,
                                                 _stream()), "copy2d_words")

    def wexp_of(self, param):
        return self.wexp.get(id(param)) if self.fmt == "h2" else None

    def push(self, fn):
        self.ops.append(fn)

    def add_grad(self, t, g):
        """grad[t] += g (g has t's shape; ownership of g passes to the tape)."""
        if id(t) in self.bwd_sums:
            raise RuntimeError("Tape.add_grad: a gradient arrived after the layer's BatchNorm sums were taken")
        cur = self.grads.get(id(t))
        if cur is None:
            self.grads[id(t)] = g
            return
        ent = self.single_consumer.get(id(t))
        if ent is not None:
            ent.pop("table", None)   # a second gradient arrives: the sums of the first are not the layer's
        B, H, W, C = g.shape
        _lib.check(self.lib.sfh_slice_add(_ptr(g), H, W, C, 0, 0, 0, _ptr(cur), B, H, W, C, 1, _stream()), "slice_add")

    def pop_grad(self, t):
        return self.grads.pop(id(t), None)

    def peek_grad(self, t):
        return self.grads.get(id(t))

    def backward(self):
        for fn in reversed(self.ops):
            fn()
        self.ops = []


# --------------------------------------------------------------------------------------- layers
STATS_IN_EPILOGUE = os.environ.get("SFH_TRAIN_STATS_EPILOGUE", "1") != "0"
BWD_SUMS_IN_EPILOGUE = os.environ.get("SFH_TRAIN_BWD_SUMS_EPILOGUE", "1") != "0"
# ConvTranspose2d backward: bias gradient + space-to-depth + split copy in one pass (sfh_s2d_split_colsum)
S2D_FUSED = os.environ.get("SFH_TRAIN_S2D_FUSED", "1") != "0"
S2D_ROWS = 32
# encoder skip tensors: BatchNorm + ReLU + MaxPool2d(2) in one forward pass (split copies only), max-pool backward +
# the BatchNorm backward sums in one backward pass (sfh_bn_apply_pool / sfh_pool2_bwd_bn_reduce)
POOL_FUSED = os.environ.get("SFH_TRAIN_POOL_FUSED", "1") != "0"
# first layer: BatchNorm backward applied inside the backward-filter kernel (sfh_conv_wgrad_c4_bn)
C4_BN_FUSED = os.environ.get("SFH_TRAIN_C4_BN_FUSED", "1") != "0"
# layers whose only consumer is an Up block's ConvTranspose2d: split copy only, backward sums from that conv's backward-data launch
UP_SUMS_FUSED = os.environ.get("SFH_TRAIN_UP_SUMS_FUSED", "1") != "0"
# the last DoubleConv's BatchNorm backward sums from the OutConv backward pass (sfh_outconv_bwd_bn)
OUTCONV_SUMS_FUSED = os.environ.get("SFH_TRAIN_OUTCONV_SUMS_FUSED", "1") != "0"
# TrainStep: the backward pass's power-of-two gradient scale chosen on the device (sfh_grad_scale), no read-back mid-step
DEVICE_GRAD_SCALE = os.environ.get("SFH_TRAIN_DEVICE_GRAD_SCALE", "1") != "0"
# BatchNorm forward: the sum of a conv epilogue's table and the finalize step in one launch (sfh_bn_finalize_partials)
FINALIZE_FUSED = os.environ.get("SFH_TRAIN_FINALIZE_FUSED", "1") != "0"
STATS_ROWS = 2048   # most rows of the table a conv epilogue adds its per-wave BatchNorm sums into (sfh_conv_desc.stats_partial)


def _bn_forward(lib, z, bn, relu, residual, tape, want_s3=True, want_f32=True, stats=None, pool=False):
    """Batch-statistics BatchNorm (+residual) (+ReLU); updates the running stats in place.  In split-operand mode
    (and want_s3) the split copy the next convolution needs is written by the same kernel.  want_f32=False (with a
    split copy): nobody reads the fp32 values - the returned tensor is a handle (shape + identity for the tape,
    no storage behind it) and the kernel writes a third less."""
    B, H, W, C = z.shape
    npix = B * H * W
    mi = _empty((2 * C,), z)
    nbt = bn.num_batches_tracked
    if nbt is not None and (nbt.dtype != torch.int64 or not nbt.is_cuda):
        raise ValueError("BatchNorm num_batches_tracked must be an int64 tensor on the GPU")
    # (the kernels also advance nn.BatchNorm2d's step counter: one launch per layer less)
    if stats is not None and FINALIZE_FUSED:
        # the conv that wrote z left the per-wave sums: no pass over z, and table sum + finalize are one launch
        _lib.check(lib.sfh_bn_finalize_partials(_ptr(stats), stats.shape[0], npix, C, float(bn.eps), BN_MOMENTUM,
                                                _ptr(bn.running_mean), _ptr(bn.running_var), _ptr(mi), _ptr(nbt), _stream()),
                   "bn_finalize_partials")
    else:
        acc = tape.zeros((2 * C,), z, torch.float64)
        if stats is not None:
            _lib.check(lib.sfh_bn_stats_partials(_ptr(stats), stats.shape[0], C, _ptr(acc), _stream()), "bn_stats_partials")
        else:
            _lib.check(lib.sfh_bn_stats(_ptr(z), npix, C, _ptr(acc), _stream()), "bn_stats")
        _lib.check(lib.sfh_bn_finalize(_ptr(acc), npix, C, float(bn.eps), BN_MOMENTUM, _ptr(bn.running_mean),
                                       _ptr(bn.running_var), _ptr(mi), _ptr(nbt), _stream()), "bn_finalize")
    if pool:
        # y and maxpool2(y) in the split format only, from one pass over z (both come back as handles without fp32 storage)
        if not (relu and residual is None and tape.use_s3 and C % 32 == 0 and H >= 2 and W >= 2):
            raise RuntimeError("_bn_forward(pool=True): needs ReLU, no residual, a split format and C % 32 == 0")
        y_s3 = E.split_empty(tape.fmt, B, H, W, C, z.device)
        p_s3 = E.split_empty(tape.fmt, B, H // 2, W // 2, C, z.device)
        _lib.check(lib.sfh_bn_apply_pool(_ptr(z), _ptr(mi), _ptr(bn.weight.detach()), _ptr(bn.bias.detach()), B, H, W, C,
                                         _ptr(y_s3), _ptr(p_s3), tape.fmt_code, _ptr(tape.overflow), _stream()),
                   "bn_apply_pool")
        y = z.new_empty((1,)).expand(z.shape)
        pl = z.new_empty((1,)).expand(B, H // 2, W // 2, C)
        for t, ts in ((y, y_s3), (pl, p_s3)):
            tape.no_f32.add(id(t))
            tape._s3[id(t)] = (t, ts)
        return y, mi, pl
    y_s3 = E.split_empty(tape.fmt, B, H, W, C, z.device) if (want_s3 and tape.use_s3 and C % 32 == 0) else None
    handle_only = y_s3 is not None and not want_f32
    y = z.new_empty((1,)).expand(z.shape) if handle_only else _empty(z.shape, z)
    _lib.check(lib.sfh_bn_apply(_ptr(z), _ptr(mi), _ptr(bn.weight.detach()), _ptr(bn.bias.detach()),
                                _ptr(residual), 1 if relu else 0, npix, C, None if handle_only else _ptr(y), _ptr(y_s3),
                                W, tape.fmt_code, _ptr(tape.overflow), _stream()), "bn_apply")
    if handle_only:
        tape.no_f32.add(id(y))
    if y_s3 is not None:
        tape._s3[id(y)] = (y, y_s3)
    return y, mi


def _bn_backward(lib, tape, dy, y, z, mi, bn, relu, want_dres, want_s3=False, want_f32=True, sums_table=None, sums=None,
                 apply=True):
    """want_dres: the layer added a residual before its ReLU (its gradient is returned as dres).
    want_f32=False (with want_s3): only the split copy of dz is written (its consumers are the split-operand
    backward-data and backward-filter kernels); the returned dz is None."""
    B, H, W, C = z.shape
    npix = B * H * W
    acc = sums if sums is not None else tape.zeros((2 * C,), z, torch.float64)
    # without a residual the ReLU decision y > 0 is a function of z alone: the kernels recompute it (same arithmetic as
    # bn_apply) instead of reading y - 8 instead of 12 bytes per element in the reduction, 16 instead of 20 in the apply
    ysign = y if (relu and want_dres) else None
    gam, bet = bn.weight.detach(), bn.bias.detach()
    if sums is not None:         # the pass that completed dy took them (pool_backward_fused)
        pass
    elif sums_table is not None:   # the backward-data launch that produced dy left the sums (conv_bn_act)
        _lib.check(lib.sfh_bn_stats_partials(_ptr(sums_table), sums_table.shape[0], C, _ptr(acc), _stream()),
                   "bn_stats_partials")
    else:
        _lib.check(lib.sfh_bn_bwd_reduce(_ptr(dy), _ptr(ysign), _ptr(z), _ptr(mi), _ptr(gam), _ptr(bet), 1 if relu else 0,
                                         npix, C, _ptr(acc), _stream()), "bn_bwd_reduce")
    if not apply:   # the caller's next kernel applies it while loading (sfh_conv_wgrad_c4_bn): sums only
        a = acc.to(torch.float32)
        return None, a[C:], a[:C], None, acc
    dres = _empty(z.shape, z) if want_dres else None
    dz_s3 = E.split_empty(tape.fmt, B, H, W, C, z.device) if (want_s3 and C % 32 == 0) else None
    dz = _empty(z.shape, z) if (want_f32 or dz_s3 is None) else None
    a = _empty((2 * C,), z)     # dbeta | dgamma as float32, written by the apply kernel
    _lib.check(lib.sfh_bn_bwd_apply(_ptr(dy), _ptr(ysign), _ptr(z), _ptr(mi), _ptr(gam), _ptr(bet), _ptr(acc),
                                    1 if relu else 0, npix, C, _ptr(dz), _ptr(dres), _ptr(dz_s3), W, tape.fmt_code,
                                    _ptr(tape.overflow), _ptr(a), _stream()), "bn_bwd_apply")
    return dz, a[C:], a[:C], dres, dz_s3   # dz, dgamma, dbeta, dresidual, S3 copy of dz


def _colsum(lib, t, C=None, cs=None):
    C = C or t.shape[-1]
    cs = cs or t.shape[-1]
    npix = t.numel() // cs
    acc = _zeros((C,), t, torch.float64)
    _lib.check(lib.sfh_colsum(_ptr(t), npix, C, cs, _ptr(acc), _stream()), "colsum")
    return acc.to(torch.float32)


def wgrad_s3_ok(ksize, stride, M, srcs):
    """the split-operand backward-filter kernel (csrc/wgrad_s3.hip) covers 3x3 and 1x1 stride-1 convs with cout % 64 == 0
    whose sources are whole split tensors (every source's channel count a multiple of 32, all of them used)"""
    return (ksize in (1, 3) and stride == 1 and M % 64 == 0
            and all(n % 32 == 0 and n == t.shape[3] for (t, n, *_r) in srcs))


def _wgrad_s3(lib, tape, dz_s3, M, srcs, B, H, W, cin_store, ksize=3):
    """raw (M, ksize^2, cin_store) on the 16-bit matrix cores; srcs as in _wgrad (fp32 NHWC tensors whose split copies
    the tape holds since the forward pass)."""
    raw = tape.zeros((M, ksize * ksize, cin_store), dz_s3)
    for (t, n, n_off, pt, pl) in srcs:
        xs = tape.s3(t)
        _lib.check(lib.sfh_conv_wgrad_s3(_ptr(dz_s3), M, _ptr(xs), t.shape[3], t.shape[1], t.shape[2], n, pt, pl,
                                         B, H, W, ksize, _ptr(raw), cin_store, n_off, tape.fmt_code, _stream()),
                   "conv_wgrad_s3")
    return raw


def _wgrad(lib, dz, srcs, B, H, W, ksize, cin_store, tape=None):
    """raw (M, k*k, cin_store) = sum_p dz[p] (x) xin[p + tap]; srcs = [(tensor, channels, n_off, pad_top, pad_left)]."""
    M = dz.shape[3]
    raw = (tape.zeros if tape is not None else _zeros)((M, ksize * ksize, cin_store), dz)
    for (t, n, n_off, pt, pl) in srcs:
        # t may be a channel slice of an NHWC tensor: the pixel stride is stride(2), not shape[3]
        _lib.check(lib.sfh_conv_wgrad(_ptr(dz), dz.shape[3], M, _ptr(t), t.stride(2), t.shape[1], t.shape[2], n,
                                      pt, pl, B, H, W, ksize, _ptr(raw), cin_store, n_off, _stream()), "conv_wgrad")
    return raw


class _Names:
    """parameter tensor -> state_dict key, for the gradient dictionary."""

    def __init__(self, module):
        self.by_id = {id(p): k for k, p in module.named_parameters()}

    def __call__(self, p):
        return self.by_id[id(p)]


def conv_bn_act(tape, names, conv, bn, srcs, B, H, W, relu=True, residual=None, need_dx=True, s3_out=True,
                f32_out=True, pool=False):
    """z = conv(cat(srcs)) + bias; y = [relu](bn_train(z) [+ residual]).
    f32_out=False: the only consumer of y is a split-operand conv (forward and backward-filter read the split
    copy): y comes back as a handle without fp32 storage (see _bn_forward).
    pool=True (split formats, ReLU, no residual): returns (y, maxpool2(y)), both as handles with split copies only,
    written by one pass; the max-pool's backward (pushed here, so that it runs right before this layer's) completes
    y's gradient and takes the BatchNorm backward sums in the same pass.

    srcs: [(tensor NHWC, channels used, pad_top, pad_left)], one or two (skip first, like torch.cat
    in unet/unet_parts.py:67).  Stride-1 3x3 / 1x1 convs, and stride-2 ones via zero-stuffing in the
    backward."""
    lib = tape.lib
    ks, stride = conv.kernel_size[0], conv.stride[0]
    w = conv.weight.detach()
    cout = w.shape[0]
    t0, c0 = srcs[0][0], srcs[0][1]
    t1, c1 = (srcs[1][0], srcs[1][1]) if len(srcs) > 1 else (None, 0)
    s3 = tape.use_s3 and c0 % 32 == 0 and c1 % 32 == 0 and c0 == t0.shape[3] and (t1 is None or c1 == t1.shape[3])
    pc = PackedConv(w, conv.bias, None, ks, c0, c1, relu=False, stride=stride, tag="train_fwd",
                    fmt=tape.fmt if s3 else None, wexp=tape.wexp_of(conv.weight), shared_unit_scale=True)
    pc.order = tape.order
    pc.overflow = tape.overflow
    ho, wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    z = _empty((B, ho, wo, cout), t0)
    # BatchNorm's batch sums ride in the conv epilogue where the kernel offers it (H2, 3x3, stride 1)
    stats = None
    if s3 and pc.stats_ok and STATS_IN_EPILOGUE:
        rows = 64
        while rows < STATS_ROWS and rows * 1024 < B * ho * wo:   # about a quarter as many rows as pixel tiles
            rows *= 2
        stats = tape.zeros((rows, 2, cout), t0, torch.float64)
    pc.run(tape.s3(t0) if s3 else t0, B, H, W, z, src1=(tape.s3(t1) if s3 else t1) if t1 is not None else None,
           pad1=(srcs[1][2], srcs[1][3]) if t1 is not None else (0, 0), stats=stats)
    if any(id(t) in tape.no_f32 for t in (t0, t1) if t is not None) and not s3:
        raise RuntimeError("conv_bn_act: a source has no fp32 storage (f32_out=False) but this conv reads fp32")
    pooled = None
    if pool:
        y, mi, pooled = _bn_forward(lib, z, bn, relu, residual, tape, stats=stats, pool=True)
    else:
        y, mi = _bn_forward(lib, z, bn, relu, residual, tape, want_s3=s3_out,   # s3_out: a conv consumes y
                            want_f32=f32_out or residual is not None, stats=stats)
    if relu and residual is None:
        tape.bn_layers[id(y)] = {"z": z, "mi": mi, "bn": bn}
    if not pool and not f32_out and residual is None and relu and BWD_SUMS_IN_EPILOGUE:
        tape.single_consumer[id(y)] = {"z": z, "mi": mi, "bn": bn}

    def backward():
        dy = tape.pop_grad(y)
        if dy is None:
            raise RuntimeError("conv_bn_act: no gradient reached this layer")
        wsrc = [(t0, min(c0 + 3 & ~3, t0.shape[3]), 0, 0, 0)]
        if t1 is not None:
            wsrc.append((t1, c1, c0, srcs[1][2], srcs[1][3]))
        wg_s3 = s3 and wgrad_s3_ok(ks, stride, cout, wsrc)
        if not wg_s3 and any(id(t) in tape.no_f32 for (t, *_r) in wsrc):
            raise RuntimeError("conv_bn_act: a source has no fp32 storage (f32_out=False) but its backward-filter reads fp32")
        # fp32 dz is read by the fp32 backward-filter kernel and by the zero-stuffing of stride-2 layers only
        ent = tape.single_consumer.pop(id(y), None)
        # the first layer (three channels stored as four, nothing upstream): its BatchNorm backward rides in the
        # backward-filter kernel's tile load - no dz tensor
        c4_bn = (C4_BN_FUSED and not need_dx and not s3 and relu and residual is None and t1 is None and ks == 3
                 and stride == 1 and c0 <= 4 and t0.shape[3] == 4 and cout % 4 == 0)
        if c4_bn:
            _, dgamma, dbeta, _, acc = _bn_backward(lib, tape, dy, y, z, mi, bn, relu, False, apply=False,
                                                    sums_table=ent.get("table") if ent is not None else None,
                                                    sums=tape.bwd_sums.pop(id(y), None))
            g = tape.param_grads
            g[names(bn.weight)], g[names(bn.bias)] = dgamma, dbeta
            if conv.bias is not None:
                g[names(conv.bias)] = tape.zeros((cout,), z)
            raw = tape.zeros((cout, 9, 4), z)
            _lib.check(lib.sfh_conv_wgrad_c4_bn(_ptr(dy), _ptr(z), _ptr(mi), _ptr(bn.weight.detach()), _ptr(bn.bias.detach()),
                                                _ptr(acc), cout, _ptr(t0), c0, B, H, W, _ptr(raw), 4, _stream()),
                       "conv_wgrad_c4_bn")
            g[names(conv.weight)] = raw.view(cout, 3, 3, 4)[..., :c0].permute(0, 3, 1, 2)
            return
        dz, dgamma, dbeta, dres, dz_s3 = _bn_backward(lib, tape, dy, y, z, mi, bn, relu, residual is not None,
                                                      want_s3=s3 and stride == 1 and (need_dx or wg_s3),
                                                      want_f32=not (s3 and stride == 1 and wg_s3),
                                                      sums_table=ent.get("table") if ent is not None else None,
                                                      sums=tape.bwd_sums.pop(id(y), None))
        g = tape.param_grads
        g[names(bn.weight)], g[names(bn.bias)] = dgamma, dbeta
        if conv.bias is not None:
            # a bias in front of a batch-statistics BatchNorm has exactly zero gradient (the batch mean
            # absorbs it); autograd's value is rounding noise around 0
            g[names(conv.bias)] = tape.zeros((cout,), z)
        if residual is not None:
            tape.add_grad(residual, dres)
        if stride == 2:  # zero-stuff dz to the input resolution: stride-1 backward from here on
            u = _empty((B, H, W, cout), dz)
            _lib.check(lib.sfh_zero_stuff2(_ptr(dz), _ptr(u), B, ho, wo, H, W, cout, _stream()), "zero_stuff2")
            dz = u
        cin_store = t0.shape[3] if t1 is None else c0 + c1
        if wg_s3:
            raw = _wgrad_s3(lib, tape, dz_s3, cout, wsrc, B, H, W, cin_store, ks)
        else:
            raw = _wgrad(lib, dz, wsrc, B, H, W, ks, cin_store, tape)
        # a strided view: the copy into the gradient buffer (TrainStep) or autograd's accumulation does the permute
        g[names(conv.weight)] = raw.view(cout, ks, ks, cin_store)[..., :c0 + c1].permute(0, 3, 1, 2)
        if not need_dx:
            return
        dz_in = ((dz_s3 if dz_s3 is not None else E.f32_to_split(dz, tape.fmt, tape.overflow)) if s3 else dz)
        if (t1 is not None and c0 % 64 == 0 and c1 % 64 == 0 and (srcs[1][2], srcs[1][3]) == (0, 0)
                and tuple(t1.shape[1:3]) == (H, W) and c0 == t0.shape[3]):
            # two sources of the same size: one backward-data conv per source writes that source's gradient directly
            # (no gradient of the concatenated tensor, no slicing passes)
            for (t, lo, hi) in ((t0, 0, c0), (t1, c0, c0 + c1)):
                bdh = PackedConv.backward_data(w[:, lo:hi].contiguous(), ks, fmt=tape.fmt if s3 else None,
                                               wexp=tape.wexp_of(conv.weight))
                bdh.order = tape.order
                dh = _empty((B, H, W, hi - lo), z)
                bdh.run(dz_in, B, H, W, dh)
                tape.add_grad(t, dh)
            return
        bd = PackedConv.backward_data(w, ks, fmt=tape.fmt if s3 else None, wexp=tape.wexp_of(conv.weight))
        bd.order = tape.order
        dx = _empty((B, H, W, bd.cout), z)
        src_ent = tape.single_consumer.get(id(t0)) if t1 is None else None
        if (src_ent is not None and s3 and bd.stats_ok and bd.cout == t0.shape[3] and tape.peek_grad(t0) is None):
            # t0 is a BatchNorm + ReLU output that only this conv consumed: its backward sums ride in this launch
            rows = 64
            while rows < STATS_ROWS and rows * 1024 < B * H * W:
                rows *= 2
            table = tape.zeros((rows, 2, bd.cout), z, torch.float64)
            sb = src_ent["bn"]
            bd.run(dz_in, B, H, W, dx, stats=table,
                   bwd=(src_ent["z"], src_ent["mi"], sb.weight.detach(), sb.bias.detach()))
            src_ent["table"] = table
        else:
            bd.run(dz_in, B, H, W, dx)
        if t1 is None:
            tape.add_grad(t0, dx)
            return
        d0 = _empty(t0.shape, z)
        _lib.check(lib.sfh_slice_add(_ptr(dx), H, W, bd.cout, 0, 0, 0, _ptr(d0), B, H, W, c0, 0, _stream()), "slice_add")
        tape.add_grad(t0, d0)
        d1 = _empty(t1.shape, z)
        _lib.check(lib.sfh_slice_add(_ptr(dx), H, W, bd.cout, c0, srcs[1][2], srcs[1][3], _ptr(d1), B, t1.shape[1],
                                     t1.shape[2], c1, 0, _stream()), "slice_add")
        tape.add_grad(t1, d1)

    tape.push(backward)
    if not pool:
        return y

    def pool_backward_fused():
        dp = tape.pop_grad(pooled)
        if dp is None:
            raise RuntimeError("conv_bn_act(pool=True): no gradient reached the pooled tensor")
        cur = tape.peek_grad(y)
        acc = tape.zeros((2 * cout,), z, torch.float64)
        fresh = cur is None
        if fresh:
            cur = _empty(z.shape, z)
        _lib.check(lib.sfh_pool2_bwd_bn_reduce(_ptr(z), _ptr(mi), _ptr(bn.weight.detach()), _ptr(bn.bias.detach()), _ptr(dp),
                                               B, ho, wo, cout, 0 if fresh else 1, _ptr(cur), _ptr(acc), _stream()),
                   "pool2_bwd_bn_reduce")
        if fresh:
            tape.add_grad(y, cur)
        tape.bwd_sums[id(y)] = acc

    tape.push(pool_backward_fused)
    return y, pooled


def maxpool2(tape, x):
    """nn.MaxPool2d(2) (unet/unet_parts.py:33)."""
    lib = tape.lib
    B, H, W, C = x.shape
    p = _empty((B, H // 2, W // 2, C), x)
    _lib.check(lib.sfh_maxpool2_fwd(_ptr(x), _ptr(p), B, H, W, C, _stream()), "maxpool2_fwd")

    def backward():
        dp = tape.pop_grad(p)
        cur = tape.peek_grad(x)
        if cur is None:
            even = H % 2 == 0 and W % 2 == 0
            cur = _empty(x.shape, x) if even else _zeros(x.shape, x)
            _lib.check(lib.sfh_maxpool2_bwd(_ptr(x), _ptr(dp), _ptr(cur), B, H, W, C, 0 if even else 1, _stream()),
                       "maxpool2_bwd")
            tape.add_grad(x, cur)
        else:
            _lib.check(lib.sfh_maxpool2_bwd(_ptr(x), _ptr(dp), _ptr(cur), B, H, W, C, 1, _stream()), "maxpool2_bwd")

    tape.push(backward)
    return p


def conv_transpose2x2(tape, names, up, x):
    """nn.ConvTranspose2d(cin, cin/2, 2, stride=2) (unet/unet_parts.py:52).  In split-operand mode the result is written
    in the split format only: its consumer is the Up block's conv, whose forward and backward-filter read the split copy
    (the returned tensor is a handle without fp32 storage, see _bn_forward)."""
    lib = tape.lib
    B, h, w, cin = x.shape
    wt = up.weight.detach()
    cout = wt.shape[1]
    s3 = tape.use_s3 and cin % 32 == 0
    pc = PackedConv(wt, up.bias, None, 1, cin, relu=False, transposed=True, tag="train_fwd",
                    fmt=tape.fmt if s3 else None, wexp=tape.wexp_of(up.weight), shared_unit_scale=True)
    pc.overflow = tape.overflow
    split_only = s3 and cout % 32 == 0 and CAPTURE is None
    if split_only:
        u = x.new_empty((1,)).expand(B, 2 * h, 2 * w, cout)
        u_s3 = E.split_empty(tape.fmt, B, 2 * h, 2 * w, cout, x.device)
        pc.run(tape.s3(x), B, h, w, u_s3)
        tape._s3[id(u)] = (u, u_s3)
        tape.no_f32.add(id(u))
    else:
        u = _empty((B, 2 * h, 2 * w, cout), x)
        pc.run(tape.s3(x) if s3 else x, B, h, w, u)

    def backward():
        du = tape.pop_grad(u)
        g = tape.param_grads
        wsrc = [(x, cin, 0, 0, 0)]
        wg_s3 = s3 and wgrad_s3_ok(1, 1, 4 * cout, wsrc)
        if wg_s3 and cout % 8 == 0 and S2D_FUSED:
            # one pass over du: bias gradient + the split copy of s (nobody reads s itself: backward-filter and
            # backward-data take the split copy)
            table = tape.zeros((S2D_ROWS, cout), x, torch.float64)     # (rows: see sfh_s2d_split_colsum)
            acc = tape.zeros((cout,), x, torch.float64)
            s_split = E.split_empty(tape.fmt, B, h, w, 4 * cout, x.device)
            _lib.check(lib.sfh_s2d_split_colsum(_ptr(du), B, h, w, cout, _ptr(s_split), tape.fmt_code, _ptr(table), S2D_ROWS,
                                                _ptr(tape.overflow), _stream()), "s2d_split_colsum")
            _lib.check(lib.sfh_bn_stats_partials(_ptr(table), S2D_ROWS, cout // 2, _ptr(acc), _stream()), "bn_stats_partials")
            g[names(up.bias)] = acc.to(torch.float32)
            s = None
        else:
            g[names(up.bias)] = _colsum(lib, du)
            s = _empty((B, h, w, 4 * cout), x)  # s[(py*2+px)*cout + co] = du[2y+py][2x+px][co]
            _lib.check(lib.sfh_space_to_depth2(_ptr(du), _ptr(s), B, 2 * h, 2 * w, cout, _stream()), "space_to_depth2")
            s_split = E.f32_to_split(s, tape.fmt, tape.overflow) if s3 else None
        if wg_s3:   # one split copy of s feeds backward-filter and backward-data
            raw = _wgrad_s3(lib, tape, s_split, 4 * cout, wsrc, B, h, w, cin, 1)
        elif id(x) in tape.no_f32:
            raise RuntimeError("conv_transpose2x2: the input has no fp32 storage but its backward-filter reads fp32")
        else:
            raw = _wgrad(lib, s, wsrc, B, h, w, 1, cin, tape)      # (4*cout, 1, cin)
        g[names(up.weight)] = raw.view(2, 2, cout, cin).permute(3, 2, 0, 1)
        bd = PackedConv.backward_data(wt, 1, transposed=True, fmt=tape.fmt if s3 else None, wexp=tape.wexp_of(up.weight))
        bd.order = tape.order
        dx = _empty((B, h, w, bd.cout), x)
        src_ent = tape.single_consumer.get(id(x))
        if src_ent is not None and s3 and bd.stats_ok and bd.cout == cin and tape.peek_grad(x) is None:
            # x is a BatchNorm + ReLU output that only this transposed conv consumed: the layer's backward sums ride in this
            # launch (as in conv_bn_act)
            rows = 64
            while rows < STATS_ROWS and rows * 1024 < B * h * w:
                rows *= 2
            table = tape.zeros((rows, 2, bd.cout), x, torch.float64)
            sb = src_ent["bn"]
            bd.run(s_split, B, h, w, dx, stats=table, bwd=(src_ent["z"], src_ent["mi"], sb.weight.detach(), sb.bias.detach()))
            src_ent["table"] = table
        else:
            bd.run(s_split if s3 else s, B, h, w, dx)
        tape.add_grad(x, dx)

    tape.push(backward)
    return u


def upsample2x(tape, x):
    """nn.Upsample(scale_factor=2, mode='bilinear', align_corners=True) (unet/unet_parts.py:49)."""
    lib = tape.lib
    B, h, w, C = x.shape
    u = _empty((B, 2 * h, 2 * w, C), x)
    _lib.check(lib.sfh_upsample2x_bilinear_nhwc(_ptr(x), _ptr(u), B, h, w, C, _stream()), "upsample2x")

    def backward():
        du = tape.pop_grad(u)
        dx = _empty(x.shape, x)
        _lib.check(lib.sfh_upsample2x_bilinear_nhwc_bwd(_ptr(du), _ptr(dx), B, h, w, C, _stream()), "upsample2x_bwd")
        tape.add_grad(x, dx)

    tape.push(backward)
    return u


def out_conv(tape, names, oc, y, B, H, W, frame_nhwc=None, stn_cs=0, sole_consumer=False):
    """OutConv (unet/unet_parts.py:74-77) -> logits NCHW (+ the STN input cat((logits, x), 1) in NHWC).
    sole_consumer: y feeds nothing but this head - when y is a BatchNorm + ReLU output, the backward pass takes that layer's
    backward sums while it writes y's gradient (sfh_outconv_bwd_bn), and reads the layer's conv output instead of y."""
    lib = tape.lib
    wt, bias = oc.conv.weight.detach(), oc.conv.bias.detach()
    nc, cin = wt.shape[0], wt.shape[1]
    logits = _empty((B, nc, H, W), y)
    stn_in = _empty((B, H, W, stn_cs), y) if frame_nhwc is not None else None
    _lib.check(lib.sfh_outconv_fwd(_ptr(y), cin, _ptr(wt), _ptr(bias), nc, B, H, W, _ptr(logits), None, _ptr(stn_in),
                                   stn_cs, _ptr(frame_nhwc), frame_nhwc.shape[3] if frame_nhwc is not None else 0,
                                   _stream()), "outconv")

    def backward(dlogits):
        """dlogits: (B,nc,H,W) contiguous - total gradient wrt the logits."""
        acc_w = tape.zeros((nc * cin,), y, torch.float64)
        acc_b = tape.zeros((nc,), y, torch.float64)
        dy = _empty(y.shape, y)
        ent = tape.bn_layers.get(id(y)) if (sole_consumer and OUTCONV_SUMS_FUSED and tape.peek_grad(y) is None) else None
        if ent is not None:
            sb = ent["bn"]
            acc_bn = tape.zeros((2 * cin,), y, torch.float64)
            _lib.check(lib.sfh_outconv_bwd_bn(_ptr(ent["z"]), _ptr(ent["mi"]), _ptr(sb.weight.detach()), _ptr(sb.bias.detach()),
                                              cin, _ptr(wt), _ptr(dlogits), nc, B, H, W, _ptr(dy), _ptr(acc_w), _ptr(acc_b),
                                              _ptr(acc_bn), _stream()), "outconv_bwd_bn")
        else:
            _lib.check(lib.sfh_outconv_bwd(_ptr(y), cin, _ptr(wt), _ptr(dlogits), nc, B, H, W, _ptr(dy), _ptr(acc_w),
                                           _ptr(acc_b), _stream()), "outconv_bwd")
        g = tape.param_grads
        g[names(oc.conv.weight)] = acc_w.to(torch.float32).view(nc, cin, 1, 1)
        g[names(oc.conv.bias)] = acc_b.to(torch.float32)
        tape.add_grad(y, dy)
        if ent is not None:
            tape.bwd_sums[id(y)] = acc_bn     # (after add_grad: a later gradient for y would be an ordering bug and raises)

    return logits, stn_in, backward


# ----------------------------------------------------------------------------------------- UNet
class UNetTrainer:
    """forward_unet (models/reconstructor.py:132-158) in training mode at the UNet's own resolution
    (both Up variants; the resizes around it live in run_forward)."""

    def __init__(self, net):
        self.net = net
        self.names = _Names(net)

    def forward(self, tape, x_nchw, want_stn_in=False, stn_cs=8):
        net, names = self.net, self.names
        lib = tape.lib
        B, _, H, W = x_nchw.shape
        x = E.nchw_to_nhwc(x_nchw, 4)

        def dconv(block, srcs, h, w, need_dx=True, s3_out=True, pool=False, to_up=False):
            (cv1, bn1), (cv2, bn2) = block.convs()
            # y1 feeds cv2 only: forward, backward-data and backward-filter of cv2 read its split copy
            mid_f32 = not (tape.use_s3 and cv1.out_channels % 32 == 0 and cv2.out_channels % 64 == 0 and CAPTURE is None)
            y1 = conv_bn_act(tape, names, cv1, bn1, srcs, B, h, w, need_dx=need_dx, f32_out=mid_f32)
            # pool: this block's output is a skip tensor (consumers: MaxPool2d(2), and an Up block's conv with all its
            # channels - conv and backward-filter read the split copy) - BatchNorm, ReLU and pooling in one pass
            pool = (pool and POOL_FUSED and tape.use_s3 and cv2.out_channels % 64 == 0 and CAPTURE is None
                    and not net.unet_bilinear and h >= 2 and w >= 2)
            # to_up: the only consumer is an Up block's ConvTranspose2d (forward, backward-filter and backward-data read
            # the split copy; its backward-data launch leaves this layer's BatchNorm backward sums)
            to_up = (to_up and UP_SUMS_FUSED and tape.use_s3 and cv2.out_channels % 64 == 0 and CAPTURE is None and not net.unet_bilinear
                     and not pool)
            out = conv_bn_act(tape, names, cv2, bn2, [(y1, y1.shape[3], 0, 0)], B, h, w, s3_out=s3_out, pool=pool,
                              f32_out=not to_up)
            return out if pool else (out, None)

        x1, p = dconv(net.inc, [(x, 3, 0, 0)], H, W, need_dx=False, pool=True)
        feats = [x1]
        h, w = H, W
        for i in range(1, 5):
            if p is None:
                p = maxpool2(tape, feats[-1])
            h, w = h // 2, w // 2
            f, p = dconv(getattr(net, f"down{i}").block, [(p, p.shape[3], 0, 0)], h, w, pool=i < 4, to_up=i == 4)
            feats.append(f)
        y = feats[4]
        for i in range(1, 5):
            up = getattr(net, f"up{i}")
            skip = feats[4 - i]
            u = upsample2x(tape, y) if net.unet_bilinear else conv_transpose2x2(tape, names, up.up, y)
            hs, ws = skip.shape[1], skip.shape[2]
            dy_, dx_ = hs - u.shape[1], ws - u.shape[2]
            # the last block feeds the 1x1 heads (fp32) only: no S3 copy of it
            y, _ = dconv(up.conv, [(skip, skip.shape[3], 0, 0), (u, u.shape[3], dy_ // 2, dx_ // 2)], hs, ws, s3_out=i < 4,
                         to_up=i < 4)
        frame = x if want_stn_in else None
        logits, stn_in, oc_bwd = out_conv(tape, names, net.outc, y, B, H, W, frame, stn_cs, sole_consumer=not net.unet_uv)
        heads = [(logits, oc_bwd)]
        uv = None
        if net.unet_uv:  # second 1x1 head on the same features (models/reconstructor.py:79,148)
            uv, _, uv_bwd = out_conv(tape, names, net.outuv, y, B, H, W)
            heads.append((uv, uv_bwd))
        return {"logits": logits, "uv": uv, "x_top": feats[4], "stn_in": stn_in, "heads": heads}


# --------------------------------------------------------------------------------------- ResNetSTN
class ResNetTrainer:
    """ResNetSTN (models/resnet.py:235-254) in training mode; BasicBlock and Bottleneck depths."""

    def __init__(self, net):
        self.net = net
        self.rn = net.resnet_reg
        self.names = _Names(net)

    def forward(self, tape, stn_in, targets, cin):
        """stn_in: (B,H,W,cs) NHWC = the STN input (models/reconstructor.py:174-183) zero-padded to cs
        channels; targets: [(NCHW tensor, first channel, channels)] - the pieces of the input that come
        from the UNet heads and whose gradient slots receive the stem's backward-data.  Returns theta (B,9)."""
        lib, rn, names = tape.lib, self.rn, self.names
        B, H, W, cs = stn_in.shape
        st = _stream
        H2, W2 = (H + 1) // 2, (W + 1) // 2
        s2d = _empty((B, H2, W2, 4 * cs), stn_in)
        _lib.check(lib.sfh_space_to_depth2(_ptr(stn_in), _ptr(s2d), B, H, W, cs, st()), "space_to_depth2")
        w0 = rn.conv0.weight.detach()
        z0 = _empty((B, H2, W2, 64), stn_in)
        if tape.fmt is not None and cs == 8 and w0.shape[0] == 64 and os.environ.get("SFH_TRAIN_STEM7", "1") != "0":
            # the tap-packed stem kernel of the inference path (csrc/stem.hip: 0.16 ms against 0.63 ms for the 4x4 fp32 conv
            # over the space-to-depth copy), in the tape's arithmetic, writing the raw conv output z
            E.StemConv(rn.conv0, None, cin, tag="train_fwd", fmt=tape.fmt, overflow=tape.overflow,
                       wexp=tape.wexp_of(rn.conv0.weight)).run(stn_in, B, H, W, z0)
        else:
            pc = PackedConv(w0, None, None, 4, 4 * cs, relu=False, stem_cin=cin, tag="train_fwd")
            pc.run(s2d, B, H2, W2, z0)
        c1, mi0 = _bn_forward(lib, z0, rn.bn1, True, None, tape, want_s3=False)
        h, w = (H2 - 1) // 2 + 1, (W2 - 1) // 2 + 1
        x = _empty((B, h, w, 64), stn_in)
        _lib.check(lib.sfh_maxpool3x3s2_fwd(_ptr(c1), _ptr(x), B, H2, W2, 64, st()), "maxpool3x3s2")
        x_pool = x

        def stem_backward():
            dx = tape.pop_grad(x_pool)
            dc1 = _empty(c1.shape, c1)
            _lib.check(lib.sfh_maxpool3x3s2_bwd(_ptr(c1), _ptr(dx), _ptr(dc1), B, H2, W2, 64, st()), "maxpool3x3s2_bwd")
            dz, dgamma, dbeta, _, _ = _bn_backward(lib, tape, dc1, c1, z0, mi0, rn.bn1, True, False)
            g = tape.param_grads
            g[names(rn.bn1.weight)], g[names(rn.bn1.bias)] = dgamma, dbeta
            # the 4x4 backward-filter instance holds 32 input channels: wider inputs go in slices
            raw = _wgrad(lib, dz, [(s2d[..., c:], min(32, 4 * cs - c), c, 0, 0) for c in range(0, 4 * cs, 32)],
                         B, H2, W2, 4, 4 * cs)                                         # (64, 16, 4*cs)
            # 4x4 taps over the space-to-depth input -> 7x7: ky = 2*ty + py - 1 (sfh_pack_conv_weights mode 2)
            r = raw.view(64, 4, 4, 2, 2, cs).permute(0, 5, 1, 3, 2, 4).reshape(64, cs, 8, 8)
            g[names(rn.conv0.weight)] = r[:, :cin, 1:, 1:].contiguous()
            for (t, c_off, nct) in targets:   # e.g. the logits = the first nc channels of the STN input
                dl = tape.peek_grad(t)
                if dl is not None:
                    _lib.check(lib.sfh_stem_bwd_data(_ptr(dz), _ptr(w0), cin, c_off, nct, B, H, W, _ptr(dl), st()),
                               "stem_bwd_data")

        tape.push(stem_backward)

        def cba(conv, bn, src, hh, ww, relu=True, residual=None, f32_out=True):
            return conv_bn_act(tape, names, conv, bn, [(src, src.shape[3], 0, 0)], B, hh, ww, relu=relu,
                               residual=residual, f32_out=f32_out)

        for li in range(1, 5):
            for bi, blk in enumerate(getattr(rn, f"layer{li}")):
                s = blk.stride
                ho, wo = (h - 1) // s + 1, (w - 1) // s + 1
                x_in = x
                idn = cba(blk.downsample[0], blk.downsample[1], x, h, w, relu=False) if blk.downsample is not None else x
                if hasattr(blk, "conv3"):
                    t = cba(blk.conv1, blk.bn1, x, h, w)
                    t = cba(blk.conv2, blk.bn2, t, h, w)
                    x = cba(blk.conv3, blk.bn3, t, ho, wo, residual=idn)
                else:
                    # t feeds conv2 only (3x3 stride 1): forward, backward-data and backward-filter read its split copy
                    mid_f32 = not (tape.use_s3 and blk.conv1.out_channels % 64 == 0 and blk.conv2.out_channels % 64 == 0
                                   and CAPTURE is None)
                    t = cba(blk.conv1, blk.bn1, x, h, w, f32_out=mid_f32)
                    x = cba(blk.conv2, blk.bn2, t, ho, wo, residual=idn)
                if CAPTURE is not None:
                    CAPTURE[f"layer{li}.{bi}"] = {"in": x_in, "t": t, "out": x}
                h, w = ho, wo
        feat = x
        C = feat.shape[3]
        wr, br = rn.reg.weight.detach(), rn.reg.bias.detach()
        theta = _empty((B, 9), stn_in)
        pooled = _empty((B, C), stn_in)
        _lib.check(lib.sfh_avgpool_linear_fwd(_ptr(feat), _ptr(wr), _ptr(br), B, h, w, C, 9, _ptr(pooled), _ptr(theta),
                                              st()), "avgpool_linear")
        fh, fw = h, w

        def head_backward():
            dth = tape.pop_grad(theta)
            acc_w = tape.zeros((9 * C,), feat, torch.float64)
            acc_b = tape.zeros((9,), feat, torch.float64)
            dfeat = _empty(feat.shape, feat)
            _lib.check(lib.sfh_avgpool_linear_bwd(_ptr(feat), _ptr(wr), _ptr(dth), B, fh, fw, C, 9, _ptr(dfeat),
                                                  _ptr(acc_w), _ptr(acc_b), st()), "avgpool_linear_bwd")
            g = tape.param_grads
            g[names(rn.reg.weight)] = acc_w.to(torch.float32).view(9, C)
            g[names(rn.reg.bias)] = acc_b.to(torch.float32)
            tape.add_grad(feat, dfeat)

        tape.push(head_backward)
        return theta


# ------------------------------------------------------------------------------- whole model
def warp_backward_theta(theta, court_img, h, w, dout, shared_template):
    """d loss / d theta of the bilinear warp (models/reconstructor.py:109-118,185-190)."""
    lib = _lib.load()
    B = theta.shape[0]
    ht, wt = court_img.shape[2], court_img.shape[3]
    acc = _zeros((B * 9,), theta, torch.float64)
    _lib.check(lib.sfh_homography_warp_bwd_theta(_ptr(theta), _ptr(court_img), 0 if shared_template else ht * wt, ht, wt,
                                                 B, h, w, _ptr(dout), _ptr(acc), _stream()), "homography_warp_bwd")
    return acc.to(torch.float32).view(B, 9)


def poi_backward_theta(theta, court_poi, dout, normalize=True):
    lib = _lib.load()
    B = theta.shape[0]
    p = court_poi[:B].contiguous()
    dth = _empty((B, 9), theta)
    _lib.check(lib.sfh_poi_project_bwd_theta(_ptr(theta), _ptr(p), B, p.shape[1], 1 if normalize else 0, _ptr(dout),
                                             _ptr(dth), _stream()), "poi_project_bwd")
    return dth


def stn_channels(net):
    """(real channels, stored channels) of the STN input for net.resnet_input (models/reconstructor.py:84-97)."""
    nc = net.mask_classes
    cin = {"IMG": 3, "MASK": nc, "IMG_AND_MASK": nc + 3, "IMG_AND_MASK_AND_UV": nc + 5}[net.resnet_input.name]
    return cin, -(-cin // 4) * 4


def run_forward(net, tape, x):
    """Reconstructor.forward (models/reconstructor.py:160-194) on the training kernels.
    -> dict: logits / uv / theta (B,9) / poi / warp_mask tensors, `heads` [(NCHW output, backward fn)]."""
    B, _, H, W = x.shape
    tape.prepare_h2(net, B * H * W)
    mode = net.resnet_input.name
    nc = net.mask_classes
    f = {"logits": None, "uv": None, "theta": None, "poi": None, "warp_mask": None, "heads": [], "shared": False}
    cin, cs = stn_channels(net) if net.use_resnet else (0, 0)
    fused = False
    if net.use_unet:
        uw, uh = net.unet_size
        tw, th = net.target_size
        resized_in, resized_out = (H, W) != (uh, uw), (tw, th) != (uw, uh)
        # OutConv writes cat((logits, x)) itself when nothing is resized around the UNet
        fused = net.use_resnet and mode == "IMG_AND_MASK" and not (resized_in or resized_out)
        xin = E.resize_nchw(x, (uh, uw), "bilinear", align_corners=False) if resized_in else x   # :134-136
        u = UNetTrainer(net).forward(tape, xin, want_stn_in=fused, stn_cs=cs)
        heads = u["heads"]
        if resized_out:   # nearest resize of logits / uv to target_size (:151-156) and its backward
            lib = tape.lib

            def resized(t, bwd):
                Bc, C = t.shape[0], t.shape[1]
                r = E.resize_nchw(t, (th, tw), "nearest")

                def back(d):
                    dt = _empty(t.shape, t)
                    _lib.check(lib.sfh_resize_nearest_nchw_bwd(_ptr(d), _ptr(dt), Bc * C, uh, uw, th, tw, _stream()),
                               "resize_nearest_bwd")
                    bwd(dt)
                return r, back

            heads = [resized(t, bwd) for t, bwd in heads]
        f.update(logits=heads[0][0], uv=heads[1][0] if len(heads) > 1 else None, heads=heads)
    if net.use_resnet:
        targets = []
        if mode == "IMG":
            stn_in = E.nchw_to_nhwc(x, cs)
        elif mode == "MASK":
            stn_in, targets = E.nchw_to_nhwc(f["logits"], cs), [(f["logits"], 0, nc)]
        elif mode == "IMG_AND_MASK":
            stn_in = u["stn_in"] if fused else E.nchw_to_nhwc(torch.cat((f["logits"], x), 1).contiguous(), cs)
            targets = [(f["logits"], 0, nc)]
        else:  # IMG_AND_MASK_AND_UV
            stn_in = E.nchw_to_nhwc(torch.cat((f["logits"], x, f["uv"]), 1).contiguous(), cs)
            targets = [(f["logits"], 0, nc), (f["uv"], nc + 3, 2)]
        theta = ResNetTrainer(net).forward(tape, stn_in, targets, cin)
        theta4 = theta.view(B, 1, 3, 3)
        f["theta"] = theta
        f["poi"] = E.poi_project(theta4, net.court_poi)
        if net.warper:
            f["shared"] = net._template_is_shared(net.court_img, B)
            ww, wh = net.warp_size
            f["warp_mask"], _ = E.homography_warp(theta4, net.court_img, wh, ww, net.warp_with_nearest,
                                                  shared_template=f["shared"])
    # the split-bf16 copies of the conv inputs stay on the tape: backward-filter reads them again (1.5x the
    # fp32 activations in HBM, released with the tape after the backward pass)
    return f


def run_backward(net, tape, f, dheads, dtheta, unscale=True, device_scale=False):
    """dheads: gradients of the head outputs in the order of f['heads'] (None = zero); dtheta (B,9) or None.
    -> {state_dict key: gradient}.  ResNet closures run first (pushed last) and add the stem's gradient
    into the head gradients; then the heads; then the UNet.
    device_scale (with unscale=False, the two-plane fp16 format): the power-of-two scale of the pass is chosen ON THE DEVICE
    (sfh_grad_scale from sfh_multi_absminmax's words) and applied through a pointer - no host read-back in the middle of the
    step; the caller divides it out with tape.gscale_dev[1] (= 1 / S) and learns of a seed that fits no scale from the
    overflow word at the end of the step."""
    S = 1.0
    tape.gscale_dev = None
    if tape.fmt == "h2" and device_scale and not unscale:
        heads = [d for d in dheads if d is not None]
        if any(not d.is_contiguous() for d in heads) or (dtheta is not None and not dtheta.is_contiguous()):
            raise ValueError("run_backward(device_scale=True): contiguous gradient tensors expected")
        seeds = heads + ([dtheta] if dtheta is not None else [])
        if seeds:
            lib = _lib.load()
            words = E.absminmax_words(seeds)
            sbuf = _empty((2,), seeds[0])
            _lib.check(lib.sfh_grad_scale(_ptr(words), len(heads), 1 if dtheta is not None else 0,
                                          int(getattr(tape, "gshift", 0)), _ptr(sbuf), _ptr(tape.overflow), _stream()),
                       "grad_scale")
            tape.gscale_dev, tape.gscale = sbuf, None
            one = sbuf[0:1]
            if CAPTURE is not None and dtheta is not None:
                CAPTURE["dtheta"] = dtheta.clone()
            dtheta = None if dtheta is None else E.vec_op(dtheta, one, "mul")
            dheads = [None if d is None else E.vec_op(d, one, "mul", out=d) for d in dheads]
    elif tape.fmt == "h2":
        # the whole backward pass is linear in its seeds: carry it at a scale the H2 copies of the gradients can hold.
        # The dense per-pixel gradient of the heads sets it: its largest element -> [2, 4); the pre-activation
        # gradients of this model then peak at 1e2..1e3 (two to three orders of magnitude above the seeds at the last
        # decoder layers, below them in the ResNet: tests/probes/train_grad_range_probe.py), inside the format's
        # window (16376 down to an absolute floor of 2^-27).  Without a head (ResNet-only models) the gradient of
        # theta does: -> [2^12, 2^13), the average pool divides it by the pixels of layer4 first.
        import math
        heads = [d for d in dheads if d is not None]
        # one read-back for both seeds: the largest head gradient and the largest theta gradient
        mx = [m for m, _ in E.absminmax([d if d.is_contiguous() else d.contiguous() for d in heads]
                                        + ([dtheta.contiguous()] if dtheta is not None else []))]
        mh = max(mx[:len(heads)]) if heads else 0.0
        mt = mx[-1] if dtheta is not None else 0.0
        if heads:
            m, target = mh, 2
        else:
            m, target = mt, 13
        if not (math.isfinite(mh) and math.isfinite(mt)):
            raise FP16RangeError("non-finite gradient at the outputs of the model")
        target += int(getattr(tape, "gshift", 0))        # TrainStep lowers it for good after a gradient overflow (sticky)
        if m > 0.0:
            S = 2.0 ** (target - math.frexp(m)[1])       # m = f * 2^e, 0.5 <= f < 1  ->  m * S in [2^(target-1), 2^target)
        if heads and 0.0 < mt * S < 2.0 ** -16:
            # the ResNet-STN branch starts from dtheta: far below the head gradients its H2 copies would sit at the
            # format's absolute floor (2^-27) and lose their bits silently.  Raise the common scale as far as the head
            # gradients allow (their layers peak 2-3 orders of magnitude above the seeds: stay below 2^6 at the
            # seeds); if that is not enough the step does not fit the format.
            S2 = 2.0 ** (-16 - math.frexp(mt)[1] + 1)
            if mh * S2 >= 2.0 ** 6:
                raise FP16RangeError("the gradients of theta are more than 2^22 below the head gradients: they do not "
                                     "fit one power-of-two scale of the two-plane fp16 format")
            S = S2
        tape.gscale = S
    if CAPTURE is not None and dtheta is not None and tape.gscale_dev is None:
        CAPTURE["dtheta"] = dtheta.clone()
    if S != 1.0:
        dtheta = None if dtheta is None else dtheta * S
        dheads = [None if d is None else d.mul_(S) for d in dheads]
    if f["theta"] is not None:
        tape.grads[id(f["theta"])] = dtheta if dtheta is not None else torch.zeros_like(f["theta"])
    for (t, _), d in zip(f["heads"], dheads):
        tape.grads[id(t)] = torch.zeros_like(t) if d is None else d
    ops, tape.ops = tape.ops, []
    k = ctx_split(ops)
    for fn in reversed(ops[k:]):
        fn()
    for t, bwd in f["heads"]:
        bwd(tape.pop_grad(t))
    for fn in reversed(ops[:k]):
        fn()
    g = tape.param_grads
    if S != 1.0 and unscale:     # (unscale=False: the caller divides tape.gscale out itself, e.g. once over a flat buffer)
        torch._foreach_mul_([t for t in g.values() if t is not None], 1.0 / S)
    ov = tape.overflow.cpu().tolist() if tape.overflow is not None else [0]
    if ov[0]:
        tape.ops, tape.param_grads = [], {}
        tape.grads.clear()
        tape._s3.clear()
        err = FP16RangeError("training step with SFH_TRAIN_PRECISION=f16x3: an activation or gradient left the range of "
                             "the two-plane fp16 format (or was not finite); use SFH_TRAIN_PRECISION=bf16x6 for this model")
        # the forward pass was clean (its copy of the word is 0): only a GRADIENT overflowed - a lower scale can hold it.
        # Bits 2 / 4 (sfh_grad_scale): a non-finite seed / head and theta gradients that fit no common scale - no scale helps.
        err.phase = "backward" if (len(ov) > 1 and not ov[1] and not (ov[0] & 6)) else "forward"
        raise err
    # the closures reference the tape and the tape the closures: break the cycle so the activations
    # are released now rather than at the next garbage collection
    tape.param_grads = {}
    tape.grads.clear()
    tape._s3.clear()
    f["heads"] = []
    return g


def theta_gradient(net, f, dtheta, dpoi, dwarp):
    """total d loss / d theta (B,9): direct + through transform_poi + through the bilinear warp"""
    theta = f["theta"]
    B = theta.shape[0]
    lib = _lib.load()
    dth = torch.zeros_like(theta) if dtheta is None else dtheta.reshape(B, 9).to(torch.float32).clone()
    if dpoi is not None:
        dth = _add_small(lib, dth, poi_backward_theta(theta, net.court_poi, dpoi.contiguous()))
    if dwarp is not None and net.warper and not net.warp_with_nearest:
        ww, wh = net.warp_size
        dth = _add_small(lib, dth, warp_backward_theta(theta, net.court_img, wh, ww, dwarp.contiguous(), f["shared"]))
    return dth


_OUT_KEYS = ("logits", "uv", "theta", "poi", "warp_mask")


class _TrainForward(torch.autograd.Function):
    """Reconstructor.forward under net.train() as one autograd node: the forward runs the HIP
    training kernels and keeps the tape; backward() turns the output gradients into parameter
    gradients (returned in the order of net.parameters())."""

    @staticmethod
    def forward(ctx, net, info, x, *params):
        tape = Tape()
        x = E._f32c(x.detach(), "input frames")
        ctx.snap_gen = None
        if tape.fmt == "h2":
            # the forward pass updates the BatchNorm statistics: keep a copy, so that a pass whose activations leave
            # the fp16 range - now, or a gradient in backward() - can be repeated with bf16x6 operands from the same
            # statistics
            snap = net.__dict__.get("_bn_snapshot")
            if snap is None:
                snap = net.__dict__["_bn_snapshot"] = _BNSnapshot(net)
            ctx.snap_gen = snap.save()
            f = run_forward(net, tape, x)
            if int(tape.overflow[0].item()):
                snap.restore()
                _warn_range_fallback()
                net.__dict__["train_range_fallbacks"] = net.__dict__.get("train_range_fallbacks", 0) + 1
                tape = Tape(fmt="s3")
                f = run_forward(net, tape, x)
        else:
            f = run_forward(net, tape, x)
        B = x.shape[0]
        keys = [k for k in _OUT_KEYS if f[k] is not None]
        outs = [f[k].view(B, 1, 3, 3) if k == "theta" else f[k] for k in keys]
        ctx.tape, ctx.f, ctx.net, ctx.keys, ctx.x = tape, f, net, keys, x
        info["keys"] = keys          # the caller labels the outputs with the node's own key list
        net.invalidate_engines()     # BatchNorm running statistics were updated through raw device pointers
        return tuple(outs)

    @staticmethod
    def backward(ctx, *douts):
        if ctx.tape is None:
            raise RuntimeError("the HIP training node was already backpropagated: its activations are released "
                               "after the first backward (retain_graph is not supported)")
        net = ctx.net
        d = dict(zip(ctx.keys, douts))

        def attempt(tape, f):
            dth = theta_gradient(net, f, d.get("theta"), d.get("poi"), d.get("warp_mask")) if f["theta"] is not None else None
            head_keys = ["logits"] + (["uv"] if f["uv"] is not None else []) if f["logits"] is not None else []
            dheads = [None if d.get(k) is None else d[k].contiguous().clone() for k in head_keys]
            return run_backward(net, tape, f, dheads, dth)
        try:
            g = attempt(ctx.tape, ctx.f)
        except FP16RangeError:
            # a gradient (or a non-finite seed) did not fit the two-plane fp16 format: like TrainStep, repeat the step
            # with the three-plane bf16 operands - forward from the BatchNorm statistics of before this step, then
            # backward; a non-finite loss then gives non-finite gradients, as in the reference.  Possible while the
            # statistics copy is still this step's (no other forward of the model since).
            snap = net.__dict__.get("_bn_snapshot")
            if ctx.tape.fmt != "h2" or snap is None or snap.generation != ctx.snap_gen:
                ctx.tape = ctx.f = ctx.x = None
                raise
            snap.restore()
            _warn_range_fallback()
            net.__dict__["train_range_fallbacks"] = net.__dict__.get("train_range_fallbacks", 0) + 1
            tape = Tape(fmt="s3")
            g = attempt(tape, run_forward(net, tape, ctx.x))
            net.invalidate_engines()
        names = _Names(net)
        grads = tuple(g.get(names(p)) for p in net.parameters())
        ctx.tape = ctx.f = ctx.x = None
        return (None, None, None) + grads


def ctx_split(ops):
    """index of the first ResNet closure on the tape (the stem's)."""
    for i, fn in enumerate(ops):
        if fn.__name__ == "stem_backward":
            return i
    return len(ops)


def train_forward(net, x):
    """models/reconstructor.py:160-194 under net.train(): dict with logits[, uv], theta, poi[, warp_mask]."""
    if not net.use_unet and net.resnet_input.name != "IMG":
        raise NotImplementedError  # like the reference: without the UNet only the frame can feed the STN
    params = tuple(net.parameters())
    info = {}
    outs = _TrainForward.apply(net, info, x, *params)
    return dict(zip(info["keys"], outs))


# ------------------------------------------------------------------- the whole step on HIP kernels
class TrainStep:
    """One iteration of the reference's training loop (train.py:155-237) without torch autograd:
    forward under batch-statistics BatchNorm, the four losses with their lambdas, backward,
    ``clip_grad_value_`` and the RMSprop update - every arithmetic step a HIP kernel of libsfh_amd.so.

    Hyper-parameter defaults are the reference's (utils/config.py:106-139, train.py:88)."""

    def __init__(self, net, lr=1e-4, weight_decay=1e-8, momentum=0.9, alpha=0.99, eps=1e-8, clip_value=0.1,
                 seg_lambda=2.0, rec_lambda=2.0, reproj_lambda=8.0, consist_lambda=1.0, rec_loss="SmoothL1",
                 seg_loss="CE", consist_loss="CE", consist_start_iter=0, optimizer="RMSprop", betas=(0.9, 0.999),
                 uv_loss="MSE", uv_lambda=2.0):
        """optimizer: "RMSprop" (momentum, alpha, eps), "SGD" (momentum) or "Adam" (betas, eps) - train.py:87-95, each behind
        clip_grad_value_(clip_value) with L2 weight_decay; uv_loss / uv_lambda: the UV head's criterion ("MSE" | "SmoothL1" |
        None) for models built with unet_uv=True (train.py:136-144,203-208; defaults utils/config.py:120,134)."""
        if optimizer not in ("RMSprop", "SGD", "Adam"):
            print(f"optimizer {optimizer} does not support yet")        # (train.py:94)
            raise NotImplementedError(f"optimizer={optimizer!r}")
        if uv_loss not in (None, "MSE", "SmoothL1"):
            raise NotImplementedError(f"uv_loss={uv_loss!r}")
        self.optimizer = optimizer
        self.betas = (float(betas[0]), float(betas[1]))
        self.uv_loss, self.uv_lambda = uv_loss, float(uv_lambda)
        if rec_loss not in ("SmoothL1", "MSE"):
            raise NotImplementedError(f"rec_loss={rec_loss!r}")
        if seg_loss not in ("CE", "focal") or consist_loss not in ("CE", "focal"):
            raise NotImplementedError(f"seg_loss={seg_loss!r} consist_loss={consist_loss!r}")
        self.focal_flags = (1 if seg_loss == "focal" else 0) | (2 if consist_loss == "focal" else 0)
        self.net = net
        self.hp = dict(lr=lr, wd=weight_decay, mu=momentum, alpha=alpha, eps=eps, clip=clip_value)
        self.lam = dict(seg=seg_lambda, rec=rec_lambda, reproj=reproj_lambda, consist=consist_lambda)
        self.rec_mse = 1 if rec_loss == "MSE" else 0
        self.consist_start_iter = consist_start_iter
        self.global_step = 0
        self._bn_snapshot = None     # f16x3 only: copies of the BatchNorm statistics for a repeated step
        self._assemble = None        # _MultiCopy into self.grads
        self.range_fallbacks = 0     # steps repeated with bf16x6 because a value left the fp16 range
        self.range_rescales = 0      # steps repeated in f16x3 with a lower gradient scale (kept for the steps that follow)
        self.grad_scale_shift = 0    # power of two the backward pass's scale is lowered by (sticky, <= 0 in normal use)
        self.force_collective = False  # one-rank world: run the gradient all-reduce anyway (sharding.py, RCCL rehearsal)
        self._init_optimizer([p for p in net.parameters()])

    def _init_optimizer(self, params):
        """gradient / state tensors and the device tables of the multi-tensor optimizer kernel"""
        import numpy as np
        self.params = params
        dev = self.params[0].device
        if dev.type != "cuda":
            raise RuntimeError("TrainStep needs the model on the GPU (there is no CPU fallback)")
        self.names = _Names(self.net)
        from . import sharding
        # all gradients live in one flat buffer: a data-parallel step is a single all-reduce
        self.gflat, self.grads = sharding.flat_views([tuple(p.shape) for p in self.params], dev)
        self.sq = [torch.zeros_like(p) for p in self.params]
        self.buf = [torch.zeros_like(p) for p in self.params]
        CH = 65536
        chunks = []
        for i, p in enumerate(self.params):
            n = p.numel()
            for off in range(0, n, CH):
                chunks.append((i, min(CH, n - off), off))
        ch = np.zeros(len(chunks), dtype=np.dtype([("t", "<i4"), ("c", "<i4"), ("o", "<i8")]))
        for j, (t, c, o) in enumerate(chunks):
            ch[j] = (t, c, o)
        self.chunks = torch.from_numpy(ch.view(np.uint8).reshape(-1).copy()).to(dev)
        self.nchunks = len(chunks)
        self._build_table()

    def _build_table(self):
        """device table (parameter, gradient, square_avg, momentum_buffer addresses) of the optimizer kernel"""
        import numpy as np
        tab = np.zeros((len(self.params), 4), dtype=np.int64)
        for i, (p, g, s, b) in enumerate(zip(self.params, self.grads, self.sq, self.buf)):
            tab[i] = (p.data_ptr(), g.data_ptr(), s.data_ptr(), b.data_ptr())
        self._pptrs = [p.data_ptr() for p in self.params]
        self.table = torch.from_numpy(tab.view(np.uint8).reshape(-1).copy()).to(self.gflat.device)

    def _check_parameter_storage(self):
        """The optimizer kernel writes the parameters through the addresses captured in the table: if the model's
        storage was replaced since (net.to(), .float(), p.data = ...), rebuild the table - or refuse when the
        parameters left this device or changed shape / dtype."""
        cur = list(self.net.parameters())
        if len(cur) != len(self.params) or any(a is not b for a, b in zip(cur, self.params)):
            raise RuntimeError("TrainStep: the model's parameter objects changed since this TrainStep was built; "
                               "create a new TrainStep (optimizer state can be carried over with state_dict())")
        for p, g in zip(self.params, self.grads):
            if p.device != g.device or p.dtype != torch.float32 or tuple(p.shape) != tuple(g.shape) or not p.is_contiguous():
                raise RuntimeError(f"TrainStep: a parameter is now {p.dtype} {tuple(p.shape)} on {p.device}; expected "
                                   f"contiguous float32 {tuple(g.shape)} on {g.device}")
        if [p.data_ptr() for p in self.params] != self._pptrs:
            self._build_table()

    def state_dict(self):
        """Optimizer state for checkpoint / resume, keyed like ``net.state_dict()`` (train.py:321-322 saves
        the model only; resuming RMSprop needs its running averages too)."""
        names = [self.names(p) for p in self.params]
        return {"global_step": self.global_step, "optimizer": self.optimizer, "betas": self.betas,
                "grad_scale_shift": self.grad_scale_shift,
                "hyper_parameters": dict(self.hp), "lambdas": dict(self.lam), "rec_mse": self.rec_mse,
                "focal_flags": self.focal_flags, "consist_start_iter": self.consist_start_iter,
                "square_avg": {n: t.detach().clone() for n, t in zip(names, self.sq)},
                "momentum_buffer": {n: t.detach().clone() for n, t in zip(names, self.buf)}}

    def load_state_dict(self, state):
        names = [self.names(p) for p in self.params]
        if state.get("optimizer", "RMSprop") != self.optimizer:
            raise RuntimeError(f"TrainStep.load_state_dict: the state belongs to {state.get('optimizer', 'RMSprop')}, this "
                               f"TrainStep runs {self.optimizer} (square_avg / momentum_buffer hold that optimizer's moments)")
        if set(state["square_avg"]) != set(names) or set(state["momentum_buffer"]) != set(names):
            raise RuntimeError("TrainStep.load_state_dict: parameter names do not match this model")
        for n, sq, buf in zip(names, self.sq, self.buf):
            for what, src in (("square_avg", state["square_avg"][n]), ("momentum_buffer", state["momentum_buffer"][n])):
                if tuple(src.shape) != tuple(sq.shape) or src.dtype != sq.dtype:
                    raise RuntimeError(f"TrainStep.load_state_dict: {what}[{n}] is {src.dtype} {tuple(src.shape)}, "
                                       f"expected {sq.dtype} {tuple(sq.shape)}")
        for n, sq, buf in zip(names, self.sq, self.buf):   # in place: the device tables point at these tensors
            sq.copy_(state["square_avg"][n])
            buf.copy_(state["momentum_buffer"][n])
        self.global_step = int(state["global_step"])
        self.grad_scale_shift = int(state.get("grad_scale_shift", self.grad_scale_shift))
        # hyper-parameters travel with the state (a learning rate lowered by ReduceLROnPlateau survives a resume)
        if "hyper_parameters" in state:
            self.hp.update(state["hyper_parameters"])
            self.lam.update(state.get("lambdas", {}))
            self.rec_mse = int(state.get("rec_mse", self.rec_mse))
            self.focal_flags = int(state.get("focal_flags", self.focal_flags))
            self.consist_start_iter = int(state.get("consist_start_iter", self.consist_start_iter))

    def loss_and_grads(self, x, batch):
        """forward + losses + backward; fills self.grads, returns a float64 device tensor of 4 values in the
        order [seg, rec, consist, reproj] (each already times its lambda).  batch: mask (B,H,W) int64, weight (B), poi (B,N,2),
        nonzeros (B,N), num_nonzero (B)[, uv (B,2,H,W)].  With SFH_TRAIN_PRECISION=f16x3 a step whose GRADIENTS leave the fp16
        range is repeated in f16x3 with a lower gradient scale that is kept from then on (range_rescales counts); one whose
        ACTIVATIONS do, or that no scale holds, is repeated with bf16x6 operands (range_fallbacks) - always from the same
        BatchNorm statistics."""
        if _train_fmt() != "h2":
            return self._loss_and_grads(x, batch, "env")
        if self._bn_snapshot is None:
            self._bn_snapshot = _BNSnapshot(self.net)
        self._bn_snapshot.save()
        for attempt in range(4):
            try:
                return self._loss_and_grads(x, batch, "h2")
            except FP16RangeError as e:
                self._bn_snapshot.restore()
                if getattr(e, "phase", None) != "backward" or self.grad_scale_shift <= -24 or attempt == 3:
                    break
                # only a gradient left the range (the forward pass was clean): the backward pass is linear in its seeds,
                # so a lower power-of-two scale holds the spike.  Lowered FOR GOOD - like the inference path's exponents -
                # and the step repeated on the same two-plane fp16 kernels; bf16x6 only if that does not help either.
                self.grad_scale_shift -= 6
                self.range_rescales += 1
        self.range_fallbacks += 1
        _warn_range_fallback()
        return self._loss_and_grads(x, batch, "s3")

    def _loss_and_grads(self, x, batch, fmt):
        net, lib = self.net, _lib.load()
        if not net.training:
            raise RuntimeError("TrainStep: call net.train() first")
        mode = net.resnet_input.name
        if not (net.use_unet and net.use_resnet and net.warper) or mode not in ("IMG_AND_MASK", "IMG_AND_MASK_AND_UV"):
            raise NotImplementedError("TrainStep covers the reference's training configurations: UNet + ResNetSTN + warper "
                                      "with resnet_input 'img+mask', or 'img+mask+uv' with the uv head")
        tape = Tape(fmt=fmt)
        tape.gshift = self.grad_scale_shift
        B, _, H, W = x.shape
        x = E._f32c(x, "input frames")
        f = run_forward(net, tape, x)
        tape.mark_forward_done()
        logits, theta, poi, warp = f["logits"], f["theta"], f["poi"], f["warp_mask"]
        self.last_outputs = {"logits": logits, "theta": theta, "poi": poi, "warp_mask": warp}   # of the step just run
        ww, wh = net.warp_size
        if (wh, ww) != (H, W):
            raise NotImplementedError("TrainStep needs warp_size == frame size (the losses compare per pixel)")
        st = _stream()
        has_uv = bool(net.unet_uv and self.uv_loss is not None)
        losses = _zeros((5 if has_uv else 4,), x, torch.float64)   # seg, rec, consist, reproj[, uv]
        mask = batch["mask"]
        if mask.dtype != torch.int64 or not mask.is_contiguous():
            raise ValueError("batch['mask'] must be a contiguous int64 tensor (B,H,W)")
        dlogits = _empty(logits.shape, x)
        dwarp = _empty(warp.shape, x)
        # train.py:214 gates on global_step * batch_size with the GLOBAL batch
        from . import sharding
        l_cons = self.lam["consist"] if self.global_step * B * sharding.world_size() >= self.consist_start_iter else 0.0
        _lib.check(lib.sfh_train_losses(_ptr(logits), _ptr(mask), _ptr(E._f32c(batch["weight"], "weight")), _ptr(warp),
                                        net.mask_classes, B, H, W, self.lam["seg"], self.lam["rec"], self.rec_mse,
                                        l_cons, self.focal_flags, _ptr(dlogits), _ptr(dwarp), _ptr(losses), st), "train_losses")
        gt_poi = E._f32c(batch["poi"], "gt poi")
        dpoi = _empty(poi.shape, x)
        _lib.check(lib.sfh_reproj_loss(_ptr(poi), _ptr(gt_poi), _ptr(E._f32c(batch["nonzeros"], "nonzeros")),
                                       _ptr(E._f32c(batch["num_nonzero"], "num_nonzero")), B, poi.shape[1],
                                       self.lam["reproj"], _ptr(dpoi), ctypes.c_void_p(losses.data_ptr() + 24), st),
                   "reproj_loss")
        dheads = [dlogits]
        if net.unet_uv:
            duv = None
            if has_uv:
                uv = f["uv"]
                gt_uv = E._f32c(batch["uv"], "gt uv")
                if tuple(gt_uv.shape) != tuple(uv.shape):
                    raise ValueError(f"batch['uv'] must have the uv head's shape {tuple(uv.shape)}")
                wgt = E._f32c(batch["weight"], "weight")
                duv = _empty(uv.shape, x)
                # models/losses.py:38-39 on a 4-D map: mean over (channel, row) -> (B, W), times the (B,) weights along the
                # LAST axis - the reference's own broadcasting rule (B == 1 or B == W), its shape error otherwise
                _lib.check(lib.sfh_uv_loss(_ptr(uv), _ptr(gt_uv), _ptr(wgt), wgt.numel(), B, uv.shape[1], H, W, self.uv_lambda,
                                           1 if self.uv_loss == "MSE" else 0, _ptr(duv),
                                           ctypes.c_void_p(losses.data_ptr() + 32), st), "uv_loss")
            dheads.append(duv)
        g = run_backward(net, tape, f, dheads, theta_gradient(net, f, None, dpoi, dwarp), unscale=False,
                         device_scale=DEVICE_GRAD_SCALE)
        srcs = []
        for p, dst in zip(self.params, self.grads):
            src = g[self.names(p)]
            srcs.append(src if tuple(src.shape) == tuple(dst.shape) else src.reshape(dst.shape))
        # one launch assembles all 182 gradients (the weight gradients are permuted views of the backward-filter buffers)
        # and divides out the power of two the backward pass was carried with (exact)
        if self._assemble is None or any(a is not b for a, b in zip(self._assemble.dsts, self.grads)):
            self._assemble = _MultiCopy(self.grads)
        if tape.gscale_dev is not None:
            self._assemble.run(srcs, scale_dev=tape.gscale_dev[1:2])
        else:
            self._assemble.run(srcs, 1.0 / tape.gscale)
        return losses

    def step(self, x, batch):
        """-> float64 device tensor [seg, rec, consist, reproj(, uv)] (each already times its lambda)."""
        self._check_parameter_storage()
        losses = self.loss_and_grads(x, batch)
        lib, hp = _lib.load(), self.hp
        from . import sharding
        # one process per GPU: frames shard by batch, gradients are summed over RCCL and averaged in the
        # optimizer kernel (BatchNorm statistics stay per rank, like DistributedDataParallel's default)
        gscale = sharding.allreduce_gradients(self.gflat, force_collective=self.force_collective)
        if self.optimizer == "RMSprop":
            _lib.check(lib.sfh_rmsprop_step(_ptr(self.table), _ptr(self.chunks), self.nchunks, hp["lr"], hp["alpha"],
                                            hp["eps"], hp["wd"], hp["mu"], hp["clip"], gscale, _stream()), "rmsprop_step")
        elif self.optimizer == "SGD":
            _lib.check(lib.sfh_sgd_step(_ptr(self.table), _ptr(self.chunks), self.nchunks, hp["lr"], hp["wd"], hp["mu"],
                                        hp["clip"], gscale, _stream()), "sgd_step")
        else:
            _lib.check(lib.sfh_adam_step(_ptr(self.table), _ptr(self.chunks), self.nchunks, hp["lr"], self.betas[0],
                                         self.betas[1], hp["eps"], hp["wd"], hp["clip"], gscale, self.global_step + 1,
                                         _stream()), "adam_step")
        self.global_step += 1
        self.net.invalidate_engines()   # weights changed through raw device pointers: predict() must re-pack
        return losses


def _add_small(lib, a, b):
    """a += b for two small (B,9) fp32 tensors (theta gradients), on the HIP mover kernel."""
    n = a.numel()
    pad = (-n) % 4
    if pad:   # slice_add works on channel quads
        a4 = torch.zeros(n + pad, dtype=a.dtype, device=a.device)
        b4 = torch.zeros(n + pad, dtype=a.dtype, device=a.device)
        a4[:n].copy_(a.reshape(-1))
        b4[:n].copy_(b.reshape(-1))
        _lib.check(lib.sfh_slice_add(_ptr(b4), 1, 1, n + pad, 0, 0, 0, _ptr(a4), 1, 1, 1, n + pad, 1, _stream()), "slice_add")
        return a4[:n].reshape(a.shape).contiguous()
    _lib.check(lib.sfh_slice_add(_ptr(b), 1, 1, n, 0, 0, 0, _ptr(a), 1, 1, 1, n, 1, _stream()), "slice_add")
    return a
