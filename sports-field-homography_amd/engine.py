"""Launch plans for the HIP hot path: weight packing, NHWC workspaces and the sequence
of C-ABI calls that implements ``forward_unet`` / ``ResNetSTN`` / warp / POI / CE.

PyTorch is used here for device memory (``torch.empty``), the current HIP stream and the
parameter storage only; every arithmetic step is a call into ``libsfh_amd.so``.
"""
import ctypes
import math
import os
import threading

import torch

from . import _lib
from ._lib import ConvDesc

BN_EPS = 1e-5

# (tile id, rows, cols) of the stride-1 workgroup tiles; stride-2 tiles have half the rows.
_TILES = ((_lib.TILE_8x32, 8, 32), (_lib.TILE_16x16, 16, 16), (_lib.TILE_32x8, 32, 8))


# the HIP stream of the engine run() on this THREAD's stack (torch.cuda.current_stream() costs ~9 us per launch);
# thread-local: another thread's run() - another model, device or torch.cuda.stream() context - has its own
_STREAM_TLS = threading.local()


def _stream():
    stack = getattr(_STREAM_TLS, "stack", None)
    if stack:
        return stack[-1]
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


class _stream_scope:
    """Resolve the current HIP stream once for all launches of one engine run (of the calling thread)."""

    def __enter__(self):
        stack = getattr(_STREAM_TLS, "stack", None)
        if stack is None:
            stack = _STREAM_TLS.stack = []
        stack.append(ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))

    def __exit__(self, *exc):
        _STREAM_TLS.stack.pop()
        return False


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _f32c(t, what):
    if t.dtype != torch.float32 or not t.is_contiguous():
        raise ValueError(f"{what}: expected a contiguous float32 tensor, got {t.dtype} "
                         f"contiguous={t.is_contiguous()}")
    if not t.is_cuda:
        raise RuntimeError(f"{what}: tensor is on {t.device}; the HIP path needs a GPU tensor "
                           "(there is no CPU fallback)")
    return t


# ---- engine-build helpers (csrc/hostprep.hip): the work around packing a checkpoint on this library's own kernels
def filled(shape, dtype, device, value=0):
    """torch.full / torch.zeros for 4-byte element types on this library's fill kernel"""
    import struct
    t = torch.empty(shape, dtype=dtype, device=device)
    if t.element_size() != 4:
        raise ValueError("filled: 4-byte element types only")
    if t.numel():
        word = struct.unpack("<I", struct.pack("<f" if dtype.is_floating_point else "<i", value))[0]
        with torch.cuda.device(t.device):
            _lib.check(_lib.load().sfh_fill_words(_ptr(t), t.numel(), word, _stream()), "fill_words")
    return t


_ABSMAX_TABLES = {}


def absminmax_words(tensors):
    """device int32 tensor of 2 words per tensor (sfh_multi_absminmax: bits of max |x|, 0x7FFFFFFF - bits of min |x|), no
    read-back - for consumers that stay on the device (sfh_grad_scale)"""
    import numpy as np
    lib = _lib.load()
    dev = tensors[0].device
    for t in tensors:
        _f32c(t, "absminmax operand")
    # the (address, size) table lives on the device; a blocking upload would synchronise the caller's stream, so the table of
    # a recurring set of tensors (a training step's gradient seeds come back at the same addresses) is uploaded once
    key = (str(dev),) + tuple((t.data_ptr(), t.numel()) for t in tensors)
    dtab = _ABSMAX_TABLES.get(key)
    if dtab is None:
        tab = np.array([(t.data_ptr(), t.numel()) for t in tensors], dtype=np.int64)
        dtab = torch.from_numpy(tab.view(np.uint8).reshape(-1)).to(dev)
        if len(_ABSMAX_TABLES) >= 16:
            _ABSMAX_TABLES.pop(next(iter(_ABSMAX_TABLES)))
        _ABSMAX_TABLES[key] = dtab
    words = filled((2 * len(tensors),), torch.int32, dev)
    _lib.check(lib.sfh_multi_absminmax(_ptr(dtab), len(tensors), _ptr(words), _stream()), "multi_absminmax")
    return words


def absminmax(tensors):
    """[(max |x|, min |x|)] of float32 device tensors - ONE launch over all of them (sfh_multi_absminmax) and ONE
    read-back, where torch would run an abs + a reduction + a host sync per tensor.  A non-finite element gives inf / nan."""
    import numpy as np
    if not tensors:
        return []
    w = absminmax_words(tensors).cpu().numpy().view(np.uint32)
    mx = w[0::2].copy().view(np.float32)
    mn = (np.uint32(0x7FFFFFFF) - w[1::2]).astype(np.uint32).view(np.float32)
    return [(float(a), float(b)) for a, b in zip(mx, mn)]


def h2_weight_exp(wmax, top=14):
    """exponent e with max |w| * 2^e in [2^(top-1), 2^top) (include/sfh_amd.h, H2 weights); 0 for an all-zero tensor"""
    import math
    if not math.isfinite(wmax):
        raise ValueError("conv weight holds non-finite values")
    return max(-100, min(100, top - math.frexp(wmax)[1])) if wmax > 0 else 0


def vec_op(a, b=None, op="scale", factor=1.0, out=None):
    """out = a * factor ("scale"), a / b ("div") or a * b * factor ("mul") on the HIP helper kernel; b is indexed modulo
    its length (a tiled operand); out may be a itself.  Small float32 vectors: a layer's folded scale / shift."""
    lib = _lib.load()
    code = {"scale": 0, "div": 1, "mul": 2}[op]
    a = _f32c(a, "vec_op operand")
    if out is None:
        out = torch.empty_like(a)
    if b is not None:
        b = _f32c(b, "vec_op operand")
    _lib.check(lib.sfh_vec_op(code, _ptr(a), _ptr(b) if b is not None else None, a.numel(), b.numel() if b is not None else 0,
                              float(factor), _ptr(out), _stream()), "vec_op")
    return out


def snapshot(t):
    """a private copy of a small float32 tensor (engines keep NO live reference to a parameter: load_state_dict writes
    parameters in place, and an engine that finishes batches in flight must still see the weights it was built from)"""
    return vec_op(_f32c(t.detach(), "snapshot operand"))


def rows_all_equal(t):
    """do all t[k] hold the bits of t[0]?  (4-byte elements, contiguous; one launch, one word read back)"""
    lib = _lib.load()
    if t.shape[0] <= 1:
        return True
    if not t.is_contiguous() or t.element_size() != 4 or not t.is_cuda:
        raise ValueError("rows_all_equal: expected a contiguous GPU tensor of 4-byte elements")
    flag = filled((1,), torch.int32, t.device)
    _lib.check(lib.sfh_rows_differ(_ptr(t), t[0].numel(), t.shape[0], _ptr(flag), _stream()), "rows_differ")
    return int(flag.cpu()[0]) == 0


def stn_input_assemble(logits, frame, uv, cs):
    """(B,H,W,cs) NHWC = cat((logits, frame, uv), 1) zero-padded (any of the three may be None): the ResNet-STN input of
    the modes the fused OutConv epilogue does not cover (models/reconstructor.py:174-183,214)"""
    lib = _lib.load()
    srcs = [_f32c(t.contiguous(), "stn input source") if t is not None else None for t in (logits, frame, uv)]
    ref = next(t for t in srcs if t is not None)
    B, _, H, W = ref.shape
    for t in srcs:
        if t is not None and (t.shape[0], t.shape[2], t.shape[3]) != (B, H, W):
            raise ValueError("stn_input_assemble: sources of different batch / size")
    out = torch.empty((B, H, W, cs), dtype=torch.float32, device=ref.device)
    ch = [t.shape[1] if t is not None else 0 for t in srcs]
    _lib.check(lib.sfh_stn_input_assemble(_ptr(srcs[0]) if srcs[0] is not None else None, ch[0],
                                          _ptr(srcs[1]) if srcs[1] is not None else None, ch[1],
                                          _ptr(srcs[2]) if srcs[2] is not None else None, ch[2], B, H, W, cs, _ptr(out),
                                          _stream()), "stn_input_assemble")
    return out


def slice_in_channels(w, c0, c1):
    """w[:, c0:c1] of an OIHW weight as a contiguous tensor (sfh_copy2d_words)"""
    lib = _lib.load()
    w = _f32c(w.detach(), "conv weight")
    cout, cin, kh, kw = w.shape
    out = torch.empty((cout, c1 - c0, kh, kw), dtype=torch.float32, device=w.device)
    _lib.check(lib.sfh_copy2d_words(ctypes.c_void_p(w.data_ptr() + 4 * c0 * kh * kw), cin * kh * kw, _ptr(out),
                                    (c1 - c0) * kh * kw, (c1 - c0) * kh * kw, cout, _stream()), "copy2d_words")
    return out


# Split ("plane") activation formats of include/sfh_amd.h, identified by the tensor dtype:
#   "s3": (B,H,C/32,3,4,W,8) bfloat16 - three bf16 planes, exact fp32 value            (precision "bf16x6")
#   "h2": (B,H,C/32,2,4,W,8) float16  - two fp16 planes of v * 2^2, 22 significand bits (precision "f16x3")
_SPLIT = {"s3": (torch.bfloat16, 3, _lib.FMT_S3), "h2": (torch.float16, 2, _lib.FMT_H2)}
_SPLIT_DTYPES = {torch.bfloat16: "s3", torch.float16: "h2"}
PRECISIONS = {"bf16x6": "s3", "f16x3": "h2", "fp32": None}


def _fmt_of(t):
    """"s3" / "h2" for a split tensor, None for fp32 NHWC"""
    return _SPLIT_DTYPES.get(t.dtype)


def _fmt_code(t):
    f = _fmt_of(t)
    return _SPLIT[f][2] if f else _lib.FMT_F32


def _chan(t):
    """channels per pixel of an activation tensor: fp32 NHWC (B,H,W,C) or split (B,H,C/32,planes,4,W,8)"""
    return t.shape[2] * 32 if t.dtype in _SPLIT_DTYPES else t.shape[3]


def _hw(t):
    """(H, W) of an activation tensor in either format"""
    return (t.shape[1], t.shape[5]) if t.dtype in _SPLIT_DTYPES else (t.shape[1], t.shape[2])


def split_shape(fmt, b, h, w, c):
    if c % 32:
        raise ValueError(f"split-format tensors need a multiple of 32 channels, got {c}")
    return (b, h, c // 32, _SPLIT[fmt][1], 4, w, 8)


def s3_shape(b, h, w, c):
    return split_shape("s3", b, h, w, c)


def s3_empty(b, h, w, c, device):
    """uninitialised split-bf16 activation tensor for c channels (c multiple of 32)"""
    return torch.empty(s3_shape(b, h, w, c), dtype=torch.bfloat16, device=device)


class FP16RangeExhausted(RuntimeError):
    """an "f16x3" activation tensor is saturated and its exponent cannot be lowered any further"""


class H2Ranges:
    """Exponents and range words of the H2 (two-plane fp16, "f16x3") activation tensors of one model.

    An H2 tensor stores u = v * 2^e and saturates beyond |v| = 65504 * 2^-e (include/sfh_amd.h); below |u| = 2^-3 its
    low plane is an fp16 subnormal and the element keeps fewer than 22 bits.  Every tensor NAME has an exponent KEY -
    tensors that enter one conv as its two sources share a key, and so do a conv output and the pooled copy its
    producer writes - and a device word that the producing kernels raise (atomic max) to the largest bit pattern of
    |v * 2^e| they produced, before saturation.  After a forward pass the host reads the words (`read`) and decides
    in BOTH directions, one decision per key:

    * a word above H2_LIMIT_BITS: the tensor was saturated; how far it overshot gives the exponent that fits (`lower`);
    * every written word of a key below RAISE_BELOW (= 4: the largest element of the key's tensors sits within 2^5 of
      the subnormal boundary, so a typical element, an order of magnitude below the peak, has lost bits - a "quiet"
      layer): the exponent is RAISED so that the peak lands in [2^12, 2^13) like after `lower` (`quiet` / `raise_`).

    The engines then repeat the pass from the first step that writes such a tensor.  Exponents start at the conventional
    2; the state is sticky, shared by the UNet and ResNet engines of a Reconstructor and kept across engine rebuilds
    (same model, new weights).  Hysteresis: the words are running maxima since the last reset (a reset happens only
    together with a decision), so a key is raised only if EVERYTHING since then was quiet, and a key that `lower` has
    touched is never raised above the exponent `lower` gave it until the weights change (`new_generation`)."""

    LIMIT = _lib.H2_LIMIT_BITS
    NONFINITE = 0x7F800000
    DEFAULT = _lib.H2_ACT_EXP
    MIN_EXP = -64
    MAX_EXP = 48          # |v| down to 2^-36 reaches [2^12, 2^13); the kernels take -64 .. 64
    RAISE_BELOW = 4.0     # stored peak |v * 2^e| below which a key counts as quiet (see above)

    def __init__(self, device, capacity=1024):
        self.device = device
        self.words = filled((capacity,), torch.int32, device) if torch.device(device).type == "cuda" else \
            torch.zeros(capacity, dtype=torch.int32, device=device)        # (CPU: the host-logic tests)
        self.exps = {}     # key -> exponent (absent = DEFAULT)
        self.slot = {}     # tensor name -> (key, word index)
        self.peak = {}     # tensor name -> largest |v| seen so far (host side, from read())
        self.ceiling = {}  # key -> exponent `lower` gave it in this weights generation: `raise_` never exceeds it
        self._nwords = 0

    def register(self, name, key=None, word_of=None):
        """name: tensor; key: name of an already registered tensor whose exponent it shares; word_of: name of a
        tensor whose word it shares (a pooled copy: its values are a subset of the other tensor's)."""
        cur = self.slot.get(name)
        k = self.slot[key][0] if key is not None else name
        if cur is not None:
            if cur[0] != k:
                raise ValueError(f"H2 tensor {name!r} is tied to exponent key {cur[0]!r}, not {k!r}")
            return
        if word_of is not None:
            idx = self.slot[word_of][1]
        else:
            idx = self._nwords
            self._nwords += 1
            if idx >= self.words.numel():
                raise RuntimeError("H2Ranges: out of range words")
        self.slot[name] = (k, idx)

    def exp(self, name):
        s = self.slot.get(name)
        return self.exps.get(s[0], self.DEFAULT) if s is not None else self.DEFAULT

    def key(self, name):
        return self.slot[name][0]

    def word_ptr(self, name):
        s = self.slot.get(name)
        return self.words.data_ptr() + 4 * s[1] if s is not None else None

    def args(self, src=None, dst=None, res=None):
        """keyword arguments of PackedConv.run / StemConv.run for a launch reading `src`, writing `dst` (+ residual)"""
        return {"exp_src": self.exp(src), "exp_dst": self.exp(dst), "exp_res": self.exp(res),
                "range_word": self.word_ptr(dst)}

    def read(self):
        """One device read-back: {tensor name: bit pattern of the largest |v * 2^e| since the words were zeroed}."""
        n = self._nwords
        if n == 0:
            return {}
        vals = self.words[:n].cpu().numpy().view("uint32")
        out = {}
        for name, (key, idx) in self.slot.items():
            b = int(vals[idx])
            out[name] = b
            if 0 < b < self.NONFINITE:
                v = _bits_to_float(b) * 2.0 ** -self.exps.get(key, self.DEFAULT)
                if v > self.peak.get(name, 0.0):
                    self.peak[name] = v
        return out

    def saturated(self, bits):
        """names whose tensor left the fp16 range, and whether any of them holds a non-finite value"""
        bad = [n for n, b in bits.items() if b > self.LIMIT]
        return bad, any(bits[n] >= self.NONFINITE for n in bad)

    def lower(self, bad, bits):
        """Lower the exponents of the saturated tensors `bad` (names; bits = read()'s dict): ONE decision per exponent
        key - a conv output and its pooled copy share word and key, a skip tensor and its up tensor share a key - from
        the largest word of the key's tensors, converted with the exponent that was in force when the words were
        written.  The new exponent puts the observed maximum into [2^12, 2^13) (8x headroom).  -> the set of keys.
        Raises FP16RangeExhausted if a key cannot go lower (the caller falls back to the three-plane operands)."""
        worst = {}
        for n in bad:
            key = self.slot[n][0]
            worst[key] = max(worst.get(key, 0), bits[n])
        before = {key: self.exps.get(key, self.DEFAULT) for key in worst}
        for key, b in worst.items():
            e = before[key]
            vmax = _bits_to_float(b) * 2.0 ** -e
            new = 13 - math.frexp(vmax)[1]                 # frexp: vmax = m * 2^x, 0.5 <= m < 1
            new = max(self.MIN_EXP, min(new, e - 1))
            if new >= e:
                raise FP16RangeExhausted(f"H2 tensor group {key!r} is saturated at the lowest exponent {e}")
            self.exps[key] = new
            self.ceiling[key] = new
        return set(worst)

    def quiet(self, bits):
        """The other direction (bits = read()'s dict of a pass WITHOUT saturated tensors): {key: larger exponent} for
        every key whose written tensors all peaked below RAISE_BELOW in stored units - the largest word of the key,
        converted with the exponent in force, goes to [2^12, 2^13).  A tensor whose word is still zero (a resumed pass
        starts behind it) is judged by the peak it showed earlier in this weights generation; one that was never seen
        says nothing, and a key with no seen tensor is left alone.  Never above the
        exponent `lower` gave the key in this weights generation, never above MAX_EXP."""
        top, skip = {}, set()
        for n, b in bits.items():
            s = self.slot.get(n)
            if s is None:
                continue
            key = s[0]
            if b > self.LIMIT:
                skip.add(key)                              # saturated / non-finite: `lower`'s business
                continue
            if b:
                stored = _bits_to_float(b)
            else:
                # not written since the words were zeroed (a resumed pass starts behind this tensor): what it showed
                # BEFORE the reset still counts - its largest |v| of this weights generation, in today's stored units -
                # so that a key shared by an early and a late tensor is never judged on the late one alone
                v = self.peak.get(n)
                if v is None:
                    continue                               # never seen (or all zeros): says nothing
                stored = v * 2.0 ** self.exps.get(key, self.DEFAULT)
            top[key] = max(top.get(key, 0.0), stored)
        plan = {}
        for key, stored in top.items():
            if key in skip or stored >= self.RAISE_BELOW or stored <= 0.0:
                continue
            e = self.exps.get(key, self.DEFAULT)
            new = 13 - math.frexp(stored * 2.0 ** -e)[1]
            new = min(new, self.MAX_EXP, self.ceiling.get(key, self.MAX_EXP))
            if new > e:
                plan[key] = new
        return plan

    def raise_(self, plan):
        """apply quiet()'s plan -> the set of keys (the caller zeroes the words and repeats the pass from the first
        launch that writes one of them)"""
        self.exps.update(plan)
        return set(plan)

    def new_generation(self):
        """The model's weights changed (or its mode): what the words and the `lower` ceilings say belongs to the old
        weights.  The exponents stay - they are the best guess for the new weights - and are re-examined in both
        directions by the first pass."""
        self._zero_words()
        self.ceiling.clear()
        self.peak.clear()

    def reset_words(self):
        self._zero_words()

    def _zero_words(self):
        if self.words.is_cuda:
            with torch.cuda.device(self.words.device):
                _lib.check(_lib.load().sfh_fill_words(_ptr(self.words), self.words.numel(), 0, _stream()), "fill_words")
        else:
            self.words.zero_()

    def headroom(self):
        """{tensor name: 65504 * 2^-e / largest |v| seen} - how far each tensor is from saturating"""
        return {n: 65504.0 * 2.0 ** -self.exp(n) / v for n, v in self.peak.items() if v > 0}


def _bits_to_float(b):
    import struct
    return struct.unpack("<f", struct.pack("<I", b & 0xFFFFFFFF))[0]


class _NoRanges:
    """stands in for H2Ranges in the other precisions: default exponents, no words"""

    def register(self, *a, **k):
        pass

    def exp(self, name):
        return _lib.H2_ACT_EXP

    def key(self, name):
        return name

    def word_ptr(self, name):
        return None

    def args(self, src=None, dst=None, res=None):
        return {}


# half-size tiles of the split-bf16 kernel (8 pixel groups): small maps and stride 2
_TILES_S3_HALF = ((_lib.TILE_8x16, 8, 16), (_lib.TILE_16x8, 16, 8))
_WG_SLOTS = 512  # workgroups resident on the chip at two per CU


def choose_tile(batch, ho, wo, stride, zrows=1):
    """fp32 kernel: pick the workgroup tile that wastes the fewest padded output pixels."""
    best = None
    for tid, th, tw in _TILES:
        if stride == 2:
            th //= 2
            ty = batch * -(-ho // th)
        else:
            ty = -(-(batch * (ho + zrows)) // th)  # flattened rows, `zrows` shared zero rows per frame
        cost = ty * th * (-(-wo // tw)) * tw
        if best is None or cost < best[0]:
            best = (cost, tid)
    return best[1]


def choose_tile_s3(batch, ho, wo, stride, zrows, nblk, ksize=3, wg_slots=_WG_SLOTS):
    """split-bf16 kernel: estimated time = rounds of resident workgroups x pixels per tile; the
    half-size tiles win when the full-size grid would leave most of the chip idle (ResNet layer3/4)
    and are the only ones whose stride-2 halo fits LDS."""
    best = None
    if stride == 2:
        cands = _TILES_S3_HALF
    else:  # stride 1: full-size tiles (half-size tiles measured no faster on ResNet layer3/4)
        cands = _TILES
    for tid, th, tw in cands:
        if stride == 2:
            ty = batch * -(-ho // th)
        else:
            ty = -(-(batch * (ho + zrows)) // th)
        ntiles = ty * (-(-wo // tw))
        rounds = -(-(ntiles * nblk) // wg_slots)
        # equal estimates: prefer the larger tile (less per-workgroup overhead, more operand reuse)
        cost = (rounds * th * tw, ntiles * th * tw, -th * tw)
        if best is None or cost < best[0]:
            best = (cost, tid)
    return best[1]


def choose_small_map(batch, ho, wo, zr, cout, tile_id, max_wgs=None):
    """Should a plain 3x3 stride-1 H2 launch take the small-map kernel (csrc/conv_small.hip: 12x20-pixel x 32-cout workgroups)
    instead of conv_s3_kernel with workgroup tile `tile_id`?  Yes when the standard grid is at most one workgroup per CU - one
    wave per SIMD: the launch then takes what its busiest SIMD takes, and only finer work units shorten it - AND the finer tiling
    gives at least 1.2x the workgroups.  At batch 16 and 640x360: ResNet layer4 (192 -> 256 workgroups: 79 -> 50 us) and layer3
    (240 -> 512: equal alone, better under the pipeline); at batch 1 most of the net."""
    th, tw = next((a, b) for t_, a, b in _TILES + _TILES_S3_HALF if t_ == tile_id)
    std = -(-(batch * (ho + zr)) // th) * -(-wo // tw) * (cout // 64)
    fine = batch * -(-ho // 12) * -(-wo // 20) * (cout // 32)
    return std <= (_SMALL_MAP_MAX if max_wgs is None else max_wgs) and fine * 5 >= std * 6


def choose_ksplit(batch, ho, wo, stride, cout, nstages, ksize=3, wg_slots=_WG_SLOTS):
    """Split-K factor for a conv whose (pixel tile, 64-cout block) grid fills at most a QUARTER of the chip's
    workgroup slots (small batches: one frame of 640x360 has 4-60 workgroups per ResNet layer): the largest factor
    that keeps the grid within one round of resident workgroups and leaves every split at least two 32-channel
    stages.  1 = no split.  Measured at batch 16 (profiles/r03_resnet_table_*.txt): layer3 (240 workgroups) 53 us
    unsplit against 49 + 10.5 us (conv x2 + finish), layer4 (192) 82 against 74 + 8 - grids of that size are bound
    by the weight stream every pixel tile pulls from L2, not by idle CUs, so they are left alone."""
    zr = ksize // 2
    zr += (ho + zr) & 1
    cands = _TILES_S3_HALF if stride == 2 else _TILES
    ntiles = None
    for tid, th, tw in cands:
        ty = batch * -(-ho // th) if stride == 2 else -(-(batch * (ho + zr)) // th)
        n = ty * (-(-wo // tw))
        ntiles = n if ntiles is None else min(ntiles, n)
    wgs = ntiles * (cout // 64)
    if wgs * 4 > wg_slots:
        return 1
    return max(1, min(wg_slots // (2 * wgs), nstages // 2, 8))


class ConvTimer:
    """Optional HIP-event timing of conv launches on the launch stream (used by bench.py for the
    live roofline figure).  Records (tag, algorithmic FLOPs, start event, end event)."""

    def __init__(self, only=None):
        """only: set of tags to time (None = every launch).  bench.py times just the dominant kernel's launches inside the
        headline region - two event records per launch are not free (all 59 timed launches of a step: +0.2 ms) - and every
        group in the unpipelined pass behind it."""
        self.records = []
        self.only = None if only is None else set(only)

    def wants(self, tag):
        return self.only is None or tag in self.only

    def summary(self):
        """-> {tag: (launches, total_flops, total_ms)}; call after a device synchronize."""
        out = {}
        for rec in self.records:
            tag, flops, e0, e1 = rec[:4]
            n, f, t = out.get(tag, (0, 0.0, 0.0))
            out[tag] = (n + 1, f + flops, t + e0.elapsed_time(e1))
        return out

    def traffic(self):
        """-> {tag: ALGORITHMIC HBM bytes of the launches}: every operand tensor read once, every result written once
        (sources at their stored width, packed weights, fp32 seed / residual where a launch has one) - the figure the
        counters' FETCH_SIZE + WRITE_SIZE are compared with (bench.py roofline.traffic_algorithmic)."""
        out = {}
        for rec in self.records:
            if len(rec) > 5 and rec[5] is not None:
                out[rec[0]] = out.get(rec[0], 0.0) + rec[5]
        return out

    def executed(self):
        """-> {tag: FLOPs the launches EXECUTED} where that differs from the algorithmic work they are credited with
        (the composed 2x2 Up conv runs 4 taps x 2C channels for the reference's 9 taps x C)."""
        out = {}
        for rec in self.records:
            if len(rec) > 4 and rec[4] is not None:
                out[rec[0]] = out.get(rec[0], 0.0) + rec[4]
        return out


# conv_small.hip (12 x 20-pixel x 32-cout workgroups, halo + weights through LDS; bit-identical results) for 3x3 stride-1 H2
# launches whose standard grid is at most one workgroup per CU, i.e. one wave per SIMD (_SMALL_MAP_MAX workgroups): ResNet layer4
# at batch 16 (192 workgroups -> 256: 79 -> 50 us per launch, profiles/r05_small_map_probe.txt), layer3 (240 -> 512: equal
# alone, +0.2 % under the pipeline), small batches.  Same-device A/B (profiles/r05_ab_small_map.txt): 13.31 -> 13.19 ms per batch
# with the threshold at 224 (layer4 only), 13.09 at 256.  SFH_SMALL_MAP=0 switches it off.
_SMALL_MAP = os.environ.get("SFH_SMALL_MAP", "1") != "0"
_SMALL_MAP_MAX = int(os.environ.get("SFH_SMALL_MAP_MAX", "256"))
_SMALL_MAP_WS_ORDER = os.environ.get("SFH_SMALL_MAP_ORDER", "") == "weights"
_SMALL_MAP_BUFFERS = int(os.environ.get("SFH_SMALL_MAP_BUFFERS", "0"))   # experiment: 1 / 2 = force one / two LDS buffers
_W8_HALF = os.environ.get("SFH_W8_HALF", "1") != "0"
_W8_HALF_ROUNDS = float(os.environ.get("SFH_W8_HALF_ROUNDS", "3"))   # rounds of 512 resident workgroups from which the shape is requested


class LaunchOrder:
    """Direction in which consecutive conv launches of ONE engine (or one training tape) walk their pixel tiles:
    alternating ("snake", sfh_conv_desc.reverse_tiles), so that a launch starts on what the same XCD wrote
    last.  Per owner, not process-wide: the order a model sees does not depend on what else ran."""

    def __init__(self, snake=None):
        self.snake = (os.environ.get("SFH_SNAKE", "1") != "0") if snake is None else bool(snake)
        self._flip = False

    def next(self):
        r = self.snake and self._flip
        self._flip = not self._flip
        return r


_UNIT = {}


def _unit_epilogue(n, dev, scale=1.0):
    """(scale * ones, zeros) of n floats on dev, shared read-only by every backward-data conv (56 per training step;
    scale: the power of two an H2 layer's accumulator is multiplied with)"""
    key = (n, str(dev), float(scale))
    v = _UNIT.get(key)
    if v is None:
        v = _UNIT[key] = (filled((n,), torch.float32, dev, float(scale)), filled((n,), torch.float32, dev))
    return v


class PackedConv:
    """One conv-shaped layer: fragment-ordered weights + folded per-channel epilogue."""

    timer = None   # set to a ConvTimer to time every launch (bench.py / profiling only)
    order = None   # LaunchOrder of the owning engine (set by the engine); None = always forward
    exp_src = _lib.H2_ACT_EXP   # H2 layers: exponent of the source tensors currently folded into `scale`
    _shared_scale = False       # `scale` is a tensor shared with other layers (training): never rewritten in place

    def __init__(self, weight, bias, bn, ksize, c0, c1=0, relu=True, transposed=False, stride=1,
                 stem_cin=0, tag="conv", s3=False, fmt=None, wexp=None, shared_unit_scale=False, frame_h2=False):
        """fmt="s3" (or s3=True): sources are split-bf16 (S3) tensors and the contraction runs as six bf16 MFMAs
        per product; fmt="h2": two-plane fp16 (H2) sources, three fp16 MFMAs per product (both sfh_conv_s3_fwd);
        otherwise fp32 sources and fp32 MFMA (sfh_conv_fwd).  frame_h2 (with fmt=None, a 3x3 conv over <= 4 channels: the
        UNet's first layer): the source is the FH2 frame tensor of sfh_frame_to_h2 - held as a float32 (B,H,W,4) tensor, 16
        bytes per pixel - and the contraction runs as three fp16 MFMAs per product (sfh_conv3x3_c4h2_fwd)."""
        lib = _lib.load()
        self.tag = tag
        self.fmt = fmt if fmt is not None else ("s3" if s3 else None)
        if self.fmt not in (None, "s3", "h2"):
            raise ValueError(f"fmt={fmt!r}: expected None, 's3' or 'h2'")
        self.s3 = self.fmt is not None     # "runs on the split-operand kernel"
        self.escale = 1.0
        dev = weight.device
        w = _f32c(weight.detach(), "conv weight")
        self.ksize, self.c0, self.c1, self.relu, self.stride = ksize, c0, c1, relu, stride
        self.transposed = transposed
        mode, aux = (1 if transposed else 0), 0
        self.stem_cin = stem_cin
        if stem_cin:  # 7x7 s2 stem re-expressed as a 4x4 conv over the space-to-depth input
            cout = w.shape[0]
            assert ksize == 4 and tuple(w.shape[1:]) == (stem_cin, 7, 7) and c1 == 0
            self.cout_real, self.cout = cout, cout
            rep, mode, aux = 1, 2, stem_cin
        elif transposed:
            cin, cout = w.shape[0], w.shape[1]
            assert ksize == 1 and cin == c0 and c1 == 0 and tuple(w.shape[2:]) == (2, 2)
            self.cout_real, self.cout = cout, 4 * cout
            rep = 4
        else:
            cout, cin = w.shape[0], w.shape[1]
            assert cin == c0 + c1 and tuple(w.shape[2:]) == (ksize, ksize), (w.shape, c0, c1, ksize)
            self.cout_real, self.cout = cout, cout
            rep = 1
        if self.cout_real % 64:
            raise ValueError(f"conv with {self.cout_real} output channels: the MFMA kernel needs a multiple of 64")
        # 3x3 conv over <= 4 input channels (the UNet's first layer): tap-packed kernel
        self.c4 = (not self.s3 and not stem_cin and not transposed and ksize == 3 and stride == 1
                   and c1 == 0 and c0 <= 4)
        self.c4h2 = bool(frame_h2) and self.c4
        if frame_h2 and not self.c4:
            raise ValueError("frame_h2 is the first-layer kernel: a 3x3 stride-1 conv over at most 4 channels, fmt=None")
        if self.c4h2:
            wx = int(wexp) if wexp is not None else h2_weight_exp(absminmax([w])[0][0])    # max |w| * 2^wx in [2^13, 2^14)
            self.escale = 2.0 ** -(wx + _lib.H2_ACT_EXP)
            self.wpacked = torch.empty(lib.sfh_packed_c4h2_weight_bytes(self.cout), dtype=torch.uint8, device=dev)
            _lib.check(lib.sfh_pack_c4h2_weights(_ptr(w), _ptr(self.wpacked), c0, self.cout, wx, _stream()), "pack_c4h2_weights")
        elif self.c4:
            self.wpacked = torch.empty((self.cout // 64) * 9 * 256, dtype=torch.float32, device=dev)
            _lib.check(lib.sfh_pack_c4_weights(_ptr(w), _ptr(self.wpacked), c0, self.cout, _stream()), "pack_c4_weights")
        elif self.s3:
            if stem_cin:
                mode, aux = 2, stem_cin
            self._pack_split(w, ksize, c0, c1, mode, aux, wexp)
        else:
            n = lib.sfh_packed_weight_floats(ksize, c0, c1, self.cout)
            if n <= 0:
                raise ValueError(f"unsupported conv geometry ksize={ksize} c0={c0} c1={c1} cout={self.cout}")
            self.wpacked = torch.empty(n, dtype=torch.float32, device=dev)
            _lib.check(lib.sfh_pack_conv_weights(_ptr(w), _ptr(self.wpacked), ksize, c0, c1, self.cout,
                                                 mode, aux, _stream()), "pack_conv_weights")
        b = _f32c(bias.detach(), "conv bias") if bias is not None else None
        if shared_unit_scale and bn is None:
            # training convs (one PackedConv per layer and step): no BatchNorm to fold - the scale is the layer's
            # power-of-two factor times ones, shared read-only between all layers of that size, the shift the bias
            # itself: no kernel launch here (a step builds 114 of these objects)
            self.scale, zeros = _unit_epilogue(self.cout, dev, self.escale)
            self.shift = zeros if b is None else (b if rep == 1 else b.repeat(rep))
            self._shared_scale = True
            return
        self.scale = torch.empty(self.cout, dtype=torch.float32, device=dev)
        self.shift = torch.empty(self.cout, dtype=torch.float32, device=dev)
        if bn is not None:
            args = [_f32c(t.detach(), "bn tensor") for t in (bn.weight, bn.bias, bn.running_mean, bn.running_var)]
            eps = float(bn.eps)
        else:
            args, eps = [None] * 4, 0.0
        _lib.check(lib.sfh_fold_bn(_ptr(b), *[_ptr(a) for a in args], eps, self.cout_real, rep,
                                   _ptr(self.scale), _ptr(self.shift), _stream()), "fold_bn")
        if self.escale != 1.0:
            vec_op(self.scale, factor=self.escale, out=self.scale)     # a power of two: exact

    @property
    def stats_ok(self):
        """this layer's launch can leave BatchNorm batch statistics from its epilogue (run(stats=...))"""
        return (self.fmt == "h2" and self.ksize in (1, 3) and self.stride == 1 and not self.relu and not self.transposed
                and not self.stem_cin and not self.c4)

    def _pack_split(self, w, ksize, c0, c1, mode, aux, wexp=None):
        """Pack w for the split-operand kernel in this layer's format.  H2: planes of w * 2^wexp with
        max |w| * 2^wexp in [2^13, 2^14) (wexp given by a caller that has the maximum already, else one
        device read-back here); self.escale = 2^-(wexp + H2_ACT_EXP) is what the accumulator has to be
        multiplied with (the caller folds it into `scale`)."""
        lib = _lib.load()
        if self.fmt == "h2":
            n = lib.sfh_packed_h2_weight_bytes(ksize, c0, c1, self.cout)
        else:
            n = lib.sfh_packed_s3_weight_bytes(ksize, c0, c1, self.cout)
        if n <= 0:
            raise ValueError(f"unsupported split-kernel conv geometry ksize={ksize} c0={c0} c1={c1} cout={self.cout}")
        self.wpacked = torch.empty(n, dtype=torch.uint8, device=w.device)
        if self.fmt == "h2":
            if wexp is None:
                wexp = h2_weight_exp(absminmax([w])[0][0])     # (callers that pack many layers pass it: one batched read-back)
            wexp = max(-100, min(100, int(wexp)))
            self.escale = 2.0 ** -(wexp + _lib.H2_ACT_EXP)
            _lib.check(lib.sfh_pack_h2_weights(_ptr(w), _ptr(self.wpacked), ksize, c0, c1, self.cout, mode, aux, wexp,
                                               _stream()), "pack_h2_weights")
        else:
            _lib.check(lib.sfh_pack_s3_weights(_ptr(w), _ptr(self.wpacked), ksize, c0, c1, self.cout, mode, aux,
                                               _stream()), "pack_s3_weights")

    @classmethod
    def fused_up(cls, conv, bn, up, c0, tag="fusedup2x2", fmt="s3", defer_pack=False):
        """The u-half of conv3x3(cat([skip, ConvTranspose2d(x)])) (+bias, BN, ReLU) as ONE 2x2 conv over the
        low-resolution x with quadrant scatter (sfh_compose_up_weights): takes x (S3), adds the fp32
        partial of the skip-half conv as residual and writes the activated S3 output.  Split-bf16 kernel only."""
        lib = _lib.load()
        self = cls.__new__(cls)
        wc = _f32c(conv.weight.detach(), "conv weight")
        wt = _f32c(up.weight.detach(), "up weight")
        bt = _f32c(up.bias.detach(), "up bias")
        dev = wc.device
        cout, cin = wc.shape[0], wc.shape[1]
        cx, c1 = wt.shape[0], wt.shape[1]
        assert cin == c0 + c1 and tuple(wc.shape[2:]) == (3, 3) and tuple(wt.shape[2:]) == (2, 2)
        if cout % 64 or cx % 32:
            raise ValueError("fused Up conv needs cout % 64 == 0 and a multiple of 32 low-resolution channels")
        self.tag, self.s3, self.c4, self.stem_cin = tag, True, False, 0
        self.fmt, self.escale = fmt, 1.0
        self.ksize, self.c0, self.c1, self.relu, self.stride, self.transposed = 2, cx, 0, True, 1, True
        self.cout, self.cout_real = 4 * cout, cout
        self.flops_per_out_pixel = 2.0 * cout * 9 * c1   # the part of the reference conv this launch stands for
        self.scale = torch.empty(self.cout, dtype=torch.float32, device=dev)
        self.shift = torch.empty(self.cout, dtype=torch.float32, device=dev)
        args = [_f32c(t.detach(), "bn tensor") for t in (bn.weight, bn.bias, bn.running_mean, bn.running_var)]
        b = _f32c(conv.bias.detach(), "conv bias") if conv.bias is not None else None
        _lib.check(lib.sfh_fold_bn(_ptr(b), *[_ptr(a) for a in args], float(bn.eps), cout, 4, _ptr(self.scale),
                                   _ptr(self.shift), _stream()), "fold_bn")
        w2 = torch.empty((4 * cout, cx, 2, 2), dtype=torch.float32, device=dev)
        self.shift_border = torch.empty((16, 4 * cout), dtype=torch.float32, device=dev)
        _lib.check(lib.sfh_compose_up_weights(_ptr(wc), cout, c0, c1, _ptr(wt), cx, _ptr(bt), _ptr(self.scale),
                                              _ptr(self.shift), _ptr(w2), _ptr(self.shift_border), _stream()),
                   "compose_up_weights")
        # the skip-half conv of the block applies the same BatchNorm scale to ITS accumulator (UNetEngine)
        self.scale_bn = snapshot(self.scale)
        self._w2 = w2
        if not defer_pack:          # defer_pack: the engine packs all composed weights behind ONE batched |w| read-back
            self.finish_pack()
        return self

    def finish_pack(self, wexp=None):
        """second half of fused_up(): pack the composed weights (H2: with exponent wexp, else from a read-back here)"""
        self._pack_split(self._w2, 2, self.c0, 0, 0, 0, wexp)
        self._w2 = None
        if self.escale != 1.0:
            vec_op(self.scale, factor=self.escale, out=self.scale)

    @classmethod
    def backward_data(cls, weight, ksize, transposed=False, tag="bwd_data", s3=False, fmt=None, wexp=None):
        """The conv that maps dz -> dx for a stride-1 nn.Conv2d (OIHW weight; taps flipped, channels
        swapped: pack mode 3) or for nn.ConvTranspose2d k2 s2 (IOHW weight; a 1x1 conv over
        space_to_depth2(dY): pack mode 4).  fp32 kernel; output channels padded to a multiple of 64."""
        lib = _lib.load()
        self = cls.__new__(cls)
        w = _f32c(weight.detach(), "conv weight")
        dev = w.device
        if transposed:
            cin, cout = w.shape[0], w.shape[1]
            assert ksize == 1 and tuple(w.shape[2:]) == (2, 2)
            c0, mode, aux = 4 * cout, 4, cout
        else:
            cout, cin = w.shape[0], w.shape[1]
            assert tuple(w.shape[2:]) == (ksize, ksize) and ksize in (1, 3)
            c0, mode, aux = cout, 3, cin
        self.fmt = fmt if fmt is not None else ("s3" if s3 else None)
        s3 = self.fmt is not None
        self.tag, self.s3, self.c4, self.stem_cin = tag, s3, False, 0
        self.escale = 1.0
        self.ksize, self.c0, self.c1, self.relu, self.stride, self.transposed = ksize, c0, 0, False, 1, False
        if cin % 64:
            raise ValueError(f"backward-data conv needs a multiple of 64 input channels, got {cin}")
        self.cout = self.cout_real = cin
        if s3:
            self._pack_split(w, ksize, c0, 0, mode, aux, wexp)
        else:
            n = lib.sfh_packed_weight_floats(ksize, c0, 0, self.cout)
            if n <= 0:
                raise ValueError(f"unsupported backward-data geometry ksize={ksize} c0={c0} cout={self.cout}")
            self.wpacked = torch.empty(n, dtype=torch.float32, device=dev)
            _lib.check(lib.sfh_pack_conv_weights(_ptr(w), _ptr(self.wpacked), ksize, c0, 0, self.cout, mode, aux,
                                                 _stream()), "pack_conv_weights")
        self.scale, self.shift = _unit_epilogue(self.cout, dev, self.escale)
        self._shared_scale = True
        return self

    def _fold_exp_src(self, e):
        """H2 layer: its sources now carry v * 2^e.  `scale` holds the factor 2^-(wexp + exp_src) that takes the
        accumulator back to real units: multiply the difference in (a power of two: exact)."""
        if e == self.exp_src:
            return
        if self._shared_scale:
            raise ValueError("this layer's epilogue scale is shared with other layers: its source exponent is fixed")
        f = 2.0 ** (self.exp_src - e)
        vec_op(self.scale, factor=f, out=self.scale)
        self.escale *= f
        self.exp_src = e

    def run(self, src0, batch, H, W, dst, src1=None, pool0=False, pad1=(0, 0), residual=None, tile=None,
            dst_pool=None, up_dst=None, head=None, wg_couts=0, exp_src=None, exp_dst=None, exp_res=None,
            range_word=None, ksplit=0, slabs=None, acc_init=None, scale=None, shift_border=None, stats=None, bwd=None,
            small=None):
        """src0/src1: NHWC float32 tensors, or split tensors of the layer's format (S3 bfloat16 / H2 float16);
        dst/residual/dst_pool: float32 NHWC or the same split format (by dtype).  H, W: conv input frame.
        H2 tensors: exp_src / exp_dst / exp_res = exponents of the sources / dst and dst_pool / an H2 residual
        (None: the conventional SFH_H2_ACT_EXP), range_word: device address of dst's range word (H2Ranges).
        ksplit > 1 (split-operand kernel, one source): the K loop is split over ksplit copies of the grid that write
        fp32 partial slabs (`slabs`: float32 tensor (ksplit, B, Ho, Wo, cout)), and sfh_splitk_finish adds them up
        with this layer's shift, residual and ReLU into dst - for grids that alone leave most of the chip idle.
        acc_init (3x3 stride 1): float32 NHWC (B, H, W, cout) tensor the accumulators start from, in accumulator units
        (sfh_conv_desc.acc_init); scale / shift_border: tensors used instead of the layer's own for this launch.
        stats (training, H2 3x3 stride-1 layers with a plain fp32 dst): zero-filled float64 table (rows, 2, cout), rows a
        power of two - the epilogue adds the per-wave sums of z and z^2 for batch-statistics BatchNorm into it
        (sfh_conv_desc.stats_partial; PackedConv.stats_ok says whether a layer qualifies).  bwd (with stats, this launch
        being a backward-data conv): (z, mean_invstd, gamma, beta) of the BatchNorm + ReLU layer whose only gradient dst
        is - the table then receives sum g and sum g * xhat (sfh_conv_desc.bwd_z).
        small: True / False forces / forbids the small-map kernel (sfh_conv_small_fwd: plain 3x3 stride-1 H2 launches, same
        bits); None: the engine's rule - the standard grid is at most one workgroup per CU (one wave per SIMD)
        and the finer tiling gives at least 1.2x the workgroups."""
        lib = _lib.load()
        d = ConvDesc()
        if (self.fmt == "h2" or getattr(self, "c4h2", False)) and exp_src is not None:
            self._fold_exp_src(int(exp_src))
        d.h2_exp_src = self.exp_src
        if exp_dst is not None:
            d.h2_exp_dst = int(exp_dst)
        if exp_res is not None:
            d.h2_exp_res = int(exp_res)
        d.src0 = src0.data_ptr()
        d.c0, d.cs0 = self.c0, _chan(src0)
        d.h0, d.w0 = _hw(src0)
        if _fmt_of(src0) != self.fmt or (src1 is not None and _fmt_of(src1) != self.fmt):
            raise ValueError(f"layer of format {self.fmt} got a source of dtype {src0.dtype}")
        d.src_fmt = _fmt_code(src0)
        d.dst_fmt = _fmt_code(dst)
        if self.fmt is not None and _fmt_of(dst) not in (None, self.fmt):
            raise ValueError(f"layer of format {self.fmt} cannot write a {dst.dtype} destination")
        for t in (dst_pool, residual):
            if t is not None and _fmt_of(t) not in (None, _fmt_of(dst)):
                raise ValueError(f"dst_pool / residual of dtype {t.dtype} beside a {dst.dtype} destination")
        ovf = getattr(self, "overflow", None)
        d.h2_overflow = ovf.data_ptr() if (ovf is not None and _fmt_of(dst) == "h2") else None
        d.h2_range = range_word if (range_word and _fmt_of(dst) == "h2") else None
        if dst_pool is not None:
            d.dst_pool, d.pool_cs = dst_pool.data_ptr(), _chan(dst_pool)
        d.pool0 = 1 if pool0 else 0
        if src1 is not None:
            d.src1 = src1.data_ptr()
            d.c1, d.cs1 = self.c1, _chan(src1)
            d.h1, d.w1 = _hw(src1)
            d.pad_top1, d.pad_left1 = pad1
        else:
            if self.c1:
                raise ValueError("layer was packed for two sources")
            d.src1 = None
        d.batch, d.H, d.W = batch, H, W
        d.ksize, d.stride = self.ksize, self.stride
        pad2 = self.ksize // 2 + (self.ksize - 1) // 2  # pad before + pad after
        ho = (H + pad2 - self.ksize) // self.stride + 1
        wo = (W + pad2 - self.ksize) // self.stride + 1
        zr = self.ksize // 2
        if self.s3:  # even rows per frame (fused 2x2 pool windows never straddle a tile edge)
            zr += (ho + zr) & 1
        if tile is not None:
            d.tile = tile
        elif self.s3:
            d.tile = choose_tile_s3(batch, ho, wo, self.stride, zr, self.cout // 64, self.ksize)
            # 128 x 128 double-buffered workgroups (conv_s3.hip): measured +3 % for 128 / 256 input channels on grids of many
            # rounds (64 -> 128: +4 %), slower for longer K or few rounds (profiles/r03_conv_rate_probe_w8half.txt)
            if (_W8_HALF and self.fmt == "h2" and self.ksize == 3 and self.stride == 1 and wg_couts == 0 and head is None
                    and stats is None and not (ksplit and ksplit > 1) and self.cout % 128 == 0
                    and 64 <= self.c0 + self.c1 <= 256):
                nt = -(-(batch * (ho + zr)) // 8) * -(-wo // 16)
                if nt * (self.cout // 128) >= _W8_HALF_ROUNDS * _WG_SLOTS:
                    d.tile, wg_couts = _lib.TILE_8x16, 128
        else:
            d.tile = choose_tile(batch, ho, wo, self.stride, zr)
        d.wpacked, d.shift = self.wpacked.data_ptr(), self.shift.data_ptr()
        d.scale = (scale if scale is not None else self.scale).data_ptr()
        d.cout, d.relu = self.cout, 1 if self.relu else 0
        if acc_init is not None:
            if acc_init.dtype != torch.float32 or tuple(acc_init.shape) != (batch, H, W, self.cout) or not acc_init.is_contiguous():
                raise ValueError(f"acc_init must be a contiguous float32 tensor {(batch, H, W, self.cout)}")
            d.acc_init = acc_init.data_ptr()
        if stats is not None:
            if (stats.dtype != torch.float64 or not stats.is_contiguous() or stats.dim() != 3
                    or tuple(stats.shape[1:]) != (2, self.cout)):
                raise ValueError(f"stats must be a contiguous float64 tensor (rows, 2, {self.cout})")
            d.stats_partial, d.stats_rows = stats.data_ptr(), int(stats.shape[0])
            if bwd is not None:
                z, mi, gamma, beta = bwd
                if (z.dtype != torch.float32 or not z.is_contiguous() or tuple(z.shape) != tuple(dst.shape)
                        or dst.dtype != torch.float32 or mi.numel() != 2 * self.cout):
                    raise ValueError("bwd: z must be a contiguous float32 tensor of dst's shape, mean_invstd 2 * cout floats")
                d.bwd_z, d.bwd_mi = z.data_ptr(), mi.data_ptr()
                d.bwd_gamma, d.bwd_beta = gamma.data_ptr(), beta.data_ptr()
        elif bwd is not None:
            raise ValueError("bwd needs the stats table")
        if head is not None:   # OutConv fused behind this conv (sfh_conv_desc.head_*)
            d.head_w, d.head_b, d.head_nc = head["w"].data_ptr(), head["b"].data_ptr(), head["nc"]
            d.head_logits = head["logits"].data_ptr()
            d.head_stn = head["stn"].data_ptr() if head.get("stn") is not None else None
            d.head_frame = head["frame"].data_ptr() if head.get("frame") is not None else None
            d.head_skip_dst = 1 if head.get("skip_dst") else 0
        d.reverse_tiles = 1 if (self.s3 and self.order is not None and self.order.next()) else 0
        d.wg_couts = wg_couts   # 0: the launcher decides (sfh_conv_desc.wg_couts)
        d.residual = residual.data_ptr() if residual is not None else None
        d.residual_f32 = 1 if (residual is not None and residual.dtype == torch.float32
                               and dst.dtype in _SPLIT_DTYPES) else 0
        sb = shift_border if shift_border is not None else getattr(self, "shift_border", None)
        d.shift_border = sb.data_ptr() if sb is not None else None
        d.dst, d.dst_cs = dst.data_ptr(), _chan(dst)
        d.out_mode = _lib.OUT_UPSCATTER2 if self.transposed else _lib.OUT_NHWC
        exp = (batch, 2 * ho, 2 * wo) if self.transposed else (batch, ho, wo)
        if up_dst is not None:   # fused Up block with F.pad: the destination is one row / column short of 2*H x 2*W
            d.up_dst_h, d.up_dst_w = up_dst
            exp = (batch,) + tuple(up_dst)
        if (dst.shape[0],) + _hw(dst) != exp or _chan(dst) < self.cout_real:
            raise ValueError(f"conv dst shape {tuple(dst.shape)} does not match {exp + (self.cout_real,)}")
        for t in (dst, dst_pool, residual, src0, src1):
            if t is not None and t.numel() * t.element_size() >= 0xFFFFFFF0:
                raise ValueError(f"tensor of {t.numel() * t.element_size()} bytes exceeds the 4 GiB buffer-descriptor "
                                 "range of the conv kernels; split the batch")
        fwd = lib.sfh_conv_s3_fwd if self.s3 else lib.sfh_conv_fwd
        finish = None
        if ksplit and ksplit > 1:
            if not self.s3 or src1 is not None or dst_pool is not None or head is not None or self.transposed:
                raise ValueError("split-K needs a plain single-source conv on the split-operand kernel")
            if slabs is None or slabs.dtype != torch.float32 or tuple(slabs.shape) != (ksplit, batch, ho, wo, self.cout):
                raise ValueError(f"split-K slabs must be a float32 tensor {(ksplit, batch, ho, wo, self.cout)}")
            zero_shift = _unit_epilogue(self.cout, dst.device)[1]
            d.dst, d.dst_cs, d.dst_fmt = slabs.data_ptr(), self.cout, _lib.FMT_F32
            d.shift, d.relu, d.residual, d.residual_f32 = zero_shift.data_ptr(), 0, None, 0
            d.h2_overflow = d.h2_range = None
            d.ksplit, d.ksplit_stride = int(ksplit), slabs.stride(0) * 4
            res_fmt = _fmt_code(residual) if residual is not None else 0

            def finish():
                _lib.check(lib.sfh_splitk_finish(
                    _ptr(slabs), int(ksplit), slabs.stride(0) * 4, _ptr(self.shift), _ptr(residual), res_fmt,
                    int(exp_res) if exp_res is not None else _lib.H2_ACT_EXP, 1 if self.relu else 0, batch * ho, wo,
                    _chan(dst), _ptr(dst), _fmt_code(dst), int(exp_dst) if exp_dst is not None else _lib.H2_ACT_EXP,
                    ctypes.c_void_p(ovf.data_ptr()) if (ovf is not None and _fmt_of(dst) == "h2") else None,
                    ctypes.c_void_p(range_word) if (range_word and _fmt_of(dst) == "h2") else None, _stream()),
                    "splitk_finish")
            if _chan(dst) != self.cout:
                raise ValueError("split-K writes all channels of dst")
        small_ok = (self.fmt == "h2" and self.ksize == 3 and self.stride == 1 and not self.transposed and src1 is None
                    and dst_pool is None and head is None and acc_init is None and stats is None and not (ksplit and ksplit > 1)
                    and not d.residual_f32 and sb is None and not self.c4)
        if small and not small_ok:
            raise ValueError("the small-map kernel takes a plain 3x3 stride-1 H2 conv (one source, no pooled output / head / "
                             "acc_init / statistics / split-K)")
        if small is None and small_ok and _SMALL_MAP and wg_couts == 0 and tile is None:
            small = choose_small_map(batch, ho, wo, zr, self.cout, d.tile)
        if small:
            fwd = lib.sfh_conv_small_fwd
            # 0: the launcher decides (two LDS buffers for grids of at most 256 workgroups); + 16 (experiment knob): keep the
            # weight-stationary block order also where the layer's weights fit an XCD's L2
            d.wg_couts = _SMALL_MAP_BUFFERS + (16 if _SMALL_MAP_WS_ORDER else 0)
        if self.c4:
            if src0.shape[-1] != 4 or pool0 or dst_pool is not None:
                raise ValueError("the <=4-channel first-layer kernel needs an fp32 NHWC source with 4 stored channels")
            fwd = lib.sfh_conv3x3_c4h2_fwd if self.c4h2 else lib.sfh_conv3x3_c4_fwd
            if self.c4h2:
                d.src_fmt = _lib.FMT_FH2   # the (B,H,W,4) float32 tensor holds sfh_frame_to_h2's 16-byte pixels, not floats
        tm = PackedConv.timer
        if tm is not None and not tm.wants(self.tag):
            tm = None
        if tm is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        _lib.check(fwd(ctypes.byref(d), _stream()), "conv_s3_fwd" if self.s3 else "conv_fwd")
        if finish is not None:
            finish()
        if tm is not None:
            e1.record()
            # algorithmic work: 2 * MACs of the reference op (real cin, real taps)
            kk = 49 if self.ksize == 4 else (4 if self.transposed else self.ksize * self.ksize)
            cin = self.stem_cin if self.ksize == 4 else self.c0 + self.c1
            flops = 2.0 * batch * ho * wo * self.cout_real * kk * cin
            fpp = getattr(self, "flops_per_out_pixel", None)
            executed = None
            if fpp is not None:   # fused Up conv: credited with the u-half of the reference's 3x3 conv only
                # what the composed conv really multiplies: 4 quadrants x 4 taps x all low-resolution channels (8/9 of the credit)
                executed = 2.0 * batch * ho * wo * self.cout * 4 * self.c0
                flops = fpp * batch * (2 * ho) * (2 * wo)
            bpe = {"h2": 4, "s3": 6}.get(self.fmt, 4)          # stored bytes per activation element
            nbytes = batch * d.h0 * d.w0 * self.c0 * bpe + self.wpacked.numel() * self.wpacked.element_size()
            if src1 is not None:
                nbytes += batch * d.h1 * d.w1 * self.c1 * bpe
            obpe = {"h2": 4, "s3": 6}.get(_fmt_of(dst), 4)
            if not (head is not None and head.get("skip_dst")):
                nbytes += exp[0] * exp[1] * exp[2] * self.cout_real * (4 if (ksplit and ksplit > 1) else obpe)
            if dst_pool is not None:
                nbytes += batch * (ho // 2) * (wo // 2) * self.cout_real * obpe
            if head is not None:
                nbytes += batch * ho * wo * head["nc"] * 4
            for t in (residual, acc_init):
                if t is not None:
                    nbytes += t.numel() * t.element_size()
            tm.records.append((self.tag, flops, e0, e1, executed, float(nbytes)))
        return dst


_UPFUSED_LEGACY_ORDER = os.environ.get("SFH_UPFUSED_ORDER", "") == "legacy"


def run_upfused(fu, sk, skip, ylow, dst, batch, H, W, exp_dst=None, range_word=None):
    """The first conv of a fused Up block as ONE launch (sfh_conv_upfused_fwd): fu / sk = the composed 2x2 conv and the skip-half
    3x3 conv of the level (PackedConv.fused_up / the skip-half PackedConv) with fu._seed_scale / fu._seed_border up to date (the
    composed conv's scale and border shifts in sk's accumulator units, as the two-launch form passes them to its first launch);
    skip / ylow / dst: H2 tensors.  Bit-identical to fu.run(...part...) followed by sk.run(..., acc_init=part)."""
    lib = _lib.load()
    d = ConvDesc()
    d.src0, d.c0, d.cs0 = skip.data_ptr(), sk.c0, _chan(skip)
    d.h0, d.w0 = _hw(skip)
    d.src1, d.c1, d.cs1 = ylow.data_ptr(), fu.c0, _chan(ylow)
    d.h1, d.w1 = _hw(ylow)
    d.batch, d.H, d.W, d.ksize, d.stride = batch, H, W, 3, 1
    d.wpacked, d.scale, d.shift = sk.wpacked.data_ptr(), sk.scale.data_ptr(), sk.shift.data_ptr()
    d.cout, d.relu = sk.cout, 1
    d.up_wpacked, d.up_scale = fu.wpacked.data_ptr(), fu._seed_scale.data_ptr()
    d.shift_border = fu._seed_border.data_ptr()
    d.dst, d.dst_cs = dst.data_ptr(), _chan(dst)
    d.src_fmt = d.dst_fmt = _lib.FMT_H2
    d.out_mode = _lib.OUT_NHWC
    d.wg_couts = 1 if _UPFUSED_LEGACY_ORDER else 0      # experiment knob: 1 = the round-5 block order (no XCD-aware mapping)
    if exp_dst is not None:
        d.h2_exp_dst = int(exp_dst)
    ovf = getattr(sk, "overflow", None)
    d.h2_overflow = ovf.data_ptr() if ovf is not None else None
    d.h2_range = range_word if range_word else None
    if (dst.shape[0],) + _hw(dst) != (batch, H, W) or _hw(skip) != (H, W) or _fmt_of(dst) != "h2" or _fmt_of(skip) != "h2":
        raise ValueError("run_upfused: skip and dst must be H2 tensors of the output's size")
    tm = PackedConv.timer
    if tm is not None and not tm.wants("upfused"):
        tm = None
    if tm is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    _lib.check(lib.sfh_conv_upfused_fwd(ctypes.byref(d), _stream()), "conv_upfused_fwd")
    if tm is not None:
        e1.record()
        # credited like the two launches it replaces: the whole reference conv over cat([skip, up]) (9 taps x (c_skip + c_up))
        c_up = fu.flops_per_out_pixel / (2.0 * sk.cout * 9)
        # executed: 9 taps x the skip channels + 4 taps x all low-resolution channels (= 8/9 of the u-half's credit)
        nbytes = (batch * H * W * (sk.c0 + sk.cout) + batch * d.h1 * d.w1 * fu.c0) * 4 \
            + (sk.wpacked.numel() * sk.wpacked.element_size() + fu.wpacked.numel() * fu.wpacked.element_size())
        tm.records.append(("upfused", 2.0 * batch * H * W * sk.cout * 9 * (sk.c0 + c_up), e0, e1,
                           2.0 * batch * H * W * sk.cout * (9 * sk.c0 + 4 * fu.c0), float(nbytes)))
    return dst


class _Workspace:
    """Named activation buffers, one per name: a call with another shape (a different batch size, the tail
    chunk of a sub-batched call) replaces the buffer instead of keeping a second full activation set in HBM."""

    def __init__(self, device):
        self.device = device
        self.bufs = {}

    def get(self, name, shape, dtype=torch.float32, zero=False):
        key = (tuple(shape), dtype)
        cur = self.bufs.get(name)
        if cur is None or cur[0] != key:
            self.bufs.pop(name, None)      # release the old block to the allocator before asking for the new one
            t = (filled(shape, dtype, self.device) if (zero and torch.empty((), dtype=dtype).element_size() == 4)
                 else (torch.zeros if zero else torch.empty)(shape, dtype=dtype, device=self.device))
            self.bufs[name] = (key, t)
            return t
        return cur[1]


class UNetEngine:
    """forward_unet (models/reconstructor.py:132-158) on the HIP kernels."""

    def __init__(self, net, device, precision="bf16x6", overflow=None, ranges=None):
        """precision: "bf16x6" - activations in split-bf16 (S3) format, contractions as six bf16
        MFMAs per product with fp32 accumulation (fp32-equivalent accuracy); "f16x3" - two-plane fp16 (H2)
        activations, three fp16 MFMAs per product (22-bit operands; `ranges`: the model's H2Ranges - per-tensor
        exponents and the device words the kernels raise to the largest magnitude they produced; `overflow`:
        optional int32 device word OR-ed with 1 on any saturation); "fp32" - fp32 activations and fp32 MFMA."""
        self.bilinear = bool(net.unet_bilinear)
        # fused Up levels where the composed 2x2 conv runs first (see run()) and the skip-half 3x3 conv finishes.  Round 2
        # (the partial added as a residual at the end of the 3x3 conv): none 629, {4} 634, {3,4} 638, all four 635 frames/s;
        # round 3 (the 3x3 conv STARTS from the partial, sfh_conv_desc.acc_init), one device, ms per batch: {3,4} without
        # seeding 14.58 / 14.70, {3,4} seeded 14.58 / 14.66, {2,3,4} 14.60 / 14.57, all four 14.43 / 14.56
        self.up_swap = {int(c) for c in os.environ.get("SFH_UP_SWAP", "1234") if c.isdigit()}
        self.up_seed = {}          # level -> the skip-half conv starts from the partial (sfh_conv_desc.acc_init)
        # level -> frames per band of the fused Up block's two launches (0 / absent: the whole batch per launch)
        self.up_bands = {int(a): int(b) for a, b in (t.split(":") for t in os.environ.get("SFH_UP_BANDS", "").split(",") if t)}
        # levels whose fused Up block runs as ONE kernel (csrc/conv_upfused.hip, round 5; bit-identical to the two-launch
        # "swap + seed" form).  Same-device A/B at 640x360 x 16 (profiles/r05_ab_up_single.txt), ms per batch pipelined: none 13.13,
        # {4} 12.93, {3,4} 12.91-12.94, {2,3,4} 12.94, all four 13.15 (at long K a wave per parity class streams too many weights);
        # 1280x720: none 50.97, {3,4} 49.9.  SFH_UP_SINGLE="" switches it off.
        self.up_single = {int(c) for c in os.environ.get("SFH_UP_SINGLE", "34") if c.isdigit()}
        if precision not in PRECISIONS:
            raise ValueError(f"precision={precision!r}: expected one of {sorted(PRECISIONS)}")
        self.device = device
        self.ws = _Workspace(device)
        self.nc = net.mask_classes
        self.fmt = fmt = PRECISIONS[precision]
        self.s3 = fmt is not None          # split-format activations
        s3 = self.s3
        self.overflow = overflow if fmt == "h2" else None
        self.ranges = (ranges if ranges is not None else H2Ranges(device)) if fmt == "h2" else _NoRanges()
        self.steps = []            # launches of the last run(), in order: [(names of the H2 tensors written, fn)]
        self._last_out = None
        L = {}

        # "f16x3": the 3-channel first layer too runs on the fp16 matrix cores, from a frame tensor split once (FH2)
        self.frame_h2 = fmt == "h2" and os.environ.get("SFH_INC0_H2", "1") != "0"

        fuse_up = not self.bilinear and s3 and os.environ.get("SFH_FUSE_UP", "1") != "0"
        ups = [(i, cin, getattr(net, f"up{i}")) for i, cin in enumerate((1024, 512, 256, 128), start=1)]
        # the skip halves of the Up blocks' first convs (unet/unet_parts.py:67: cat([skip, up])) as tensors of their own
        skip_w = {i: slice_in_channels(up.conv.convs()[0][0].weight, 0, cin // 2) for i, cin, up in ups} if fuse_up else {}
        # "f16x3": the exponent of every weight tensor from ONE batched |w| reduction and one read-back
        wx = {}
        if fmt == "h2":
            ws_ = [p.detach() for n, p in net.named_parameters()
                   if p.dim() == 4 and not n.startswith("resnet_reg.") and p.is_contiguous()] + list(skip_w.values())
            for w, (mx, _) in zip(ws_, absminmax(ws_)):
                wx[w.data_ptr()] = h2_weight_exp(mx)

        def wexp(w):
            return wx.get(w.data_ptr())

        def dc(name, block, c0, c1=0, first_fmt=fmt):
            (cv1, bn1), (cv2, bn2) = block.convs()
            L[name + ".0"] = PackedConv(cv1.weight, cv1.bias, bn1, 3, c0, c1, tag="doubleconv3x3", fmt=first_fmt,
                                        frame_h2=(self.frame_h2 and first_fmt is None and c0 <= 4), wexp=wexp(cv1.weight))
            L[name + ".3"] = PackedConv(cv2.weight, cv2.bias, bn2, 3, cv1.out_channels, tag="doubleconv3x3", fmt=fmt,
                                        wexp=wexp(cv2.weight))

        dc("inc", net.inc, 3, first_fmt=None)  # 3-channel input: fp32 kernel (writes the split format itself)
        for i, cin in enumerate((64, 128, 256, 512), start=1):
            dc(f"down{i}", getattr(net, f"down{i}").block, cin)
        for i, cin, up in ups:
            if fuse_up:
                # ConvTranspose2d folded into the consumer conv (used when no F.pad is needed): the
                # skip-half 3x3 conv leaves an fp32 partial, the composed 2x2 conv over the low-resolution
                # tensor finishes it - the up-sampled tensor is never written
                (cv1, bn1), _ = up.conv.convs()
                c0s = cin // 2
                L[f"up{i}.skip"] = PackedConv(skip_w[i], None, None, 3, c0s, relu=False, tag="doubleconv3x3", fmt=fmt,
                                              wexp=wexp(skip_w[i]))
                L[f"up{i}.fused"] = PackedConv.fused_up(cv1, bn1, up.up, c0s, fmt=fmt, defer_pack=True)
            if not self.bilinear:  # bilinear variant (A3b): parameter-free 2x upsampling kernel instead
                L[f"up{i}.up"] = PackedConv(up.up.weight, up.up.bias, None, 1, cin, relu=False, transposed=True,
                                            tag="convT2x2", fmt=fmt, wexp=wexp(up.up.weight))
            dc(f"up{i}.conv", up.conv, cin // 2, cin // 2)  # cat([skip, up]): cin/2 channels each in both variants
        if fuse_up:
            # second batched read-back: the composed 2x2 weights (they exist only now) and the smallest |BatchNorm scale| of
            # each level (accumulator seeding divides by it)
            fus = [L[f"up{i}.fused"] for i, _, _ in ups]
            mm = absminmax([f._w2 for f in fus] + [f.scale_bn for f in fus])
            for k, (i, cin, up) in enumerate(ups):
                fu, sk = fus[k], L[f"up{i}.skip"]
                fu.finish_pack(h2_weight_exp(mm[k][0]) if fmt == "h2" else None)
                cout = up.conv.convs()[0][0].out_channels
                # the partial enters the fused conv's epilogue as a residual, i.e. after the BatchNorm scale:
                # the skip-half carries that scale itself (shift stays 0), times its own operand scaling
                vec_op(fu.scale_bn[:cout], factor=sk.escale, out=sk.scale)
                # accumulator seeding (run()) divides by this scale: only where no channel's BatchNorm scale vanishes
                smin = mm[len(fus) + k][1] * sk.escale
                self.up_seed[i] = (os.environ.get("SFH_UP_SEED", "1") != "0") and 1e-30 < smin < float("inf")
        self.L = L
        self.order = LaunchOrder()
        for layer in L.values():
            layer.order = self.order
            layer.overflow = self.overflow
        # private copies: an engine holds NO live reference to a parameter (see snapshot())
        self.outc_w = snapshot(net.outc.conv.weight)
        self.outc_b = snapshot(net.outc.conv.bias)
        self.outuv = None
        if net.outuv is not None:
            self.outuv = (snapshot(net.outuv.conv.weight), snapshot(net.outuv.conv.bias))

    def run(self, x, want_stn_in=False, want_argmax=False, want_uv=False, stn_slot=0):
        """x: (B,3,H,W) float32 NCHW on the GPU.  Returns dict with logits (NCHW, fresh),
        and optionally stn_in (NHWC8 workspace), argmax (B,H,W uint8), uv, plus the NHWC
        workspace tensors x_top / y4 for callers that need them (x_top_exp: exponent of x_top if it is H2)."""
        with _stream_scope():
            return self._run(x, want_stn_in, want_argmax, want_uv, stn_slot)

    def first_step(self, keys):
        """index of the first launch of the last run() that writes an H2 tensor whose exponent key is in `keys`"""
        rg = self.ranges
        for i, (outs, _) in enumerate(self.steps):
            if any(rg.key(n) in keys for n in outs):
                return i
        return None

    def rerun(self, first):
        """Repeat the launches of the last run() from index `first` on (same buffers, same output tensors) with the
        exponents H2Ranges holds NOW: what the range guard does after lowering the exponent of a saturated tensor."""
        with _stream_scope():
            for _, fn in self.steps[first:]:
                fn()
        return self._last_out

    def _run(self, x, want_stn_in, want_argmax, want_uv, stn_slot=0):
        """stn_slot: which of the STN-input buffers this pass writes (Reconstructor.predict_async alternates two, so that
        the ResNet of batch k can still read its input while the UNet of batch k + 1 writes the other)"""
        lib = _lib.load()
        x = _f32c(x, "input frames")
        B, C, H, W = x.shape
        if C != 3:
            raise ValueError(f"expected 3 input channels, got {C}")
        if H < 16 or W < 16:
            raise ValueError("frames smaller than 16x16 cannot pass four 2x2 poolings")
        ws, L, rg = self.ws, self.L, self.ranges
        steps = self.steps = []

        def do(outs, fn):
            """record + execute one launch; outs = names of the H2 tensors it writes; everything that depends on an
            exponent is looked up inside fn, i.e. again when the step is repeated"""
            steps.append((tuple(outs), fn))
            fn()

        xin = ws.get("xin", (B, H, W, 4))
        first = (xin, None)
        if self.frame_h2:
            # one pass writes the fp32 NHWC frame (the fused head reads it) AND its two fp16 planes, 16 bytes per pixel,
            # held as a float32 (B,H,W,4) tensor; "frame" has an exponent and a range word like every H2 tensor
            fh2 = ws.get("frame_h2", (B, H, W, 4))
            rg.register("frame")
            ovf = self.overflow
            do(("frame",), lambda: _lib.check(lib.sfh_frame_to_h2(
                _ptr(x), _ptr(xin), _ptr(fh2), B, 3, H, W, rg.exp("frame"),
                ctypes.c_void_p(ovf.data_ptr()) if ovf is not None else None,
                ctypes.c_void_p(rg.word_ptr("frame")) if rg.word_ptr("frame") else None, _stream()), "frame_to_h2"))
            first = (fh2, "frame")
        else:
            do((), lambda: _lib.check(lib.sfh_nchw_to_nhwc(_ptr(x), _ptr(xin), B, 3, H, W, 4, _stream()), "nchw_to_nhwc"))

        s3, fmt = self.s3, self.fmt

        def act(name, shape_bhw, c, f32=False, key=None, word_of=None):
            """activation workspace: the engine's split format (S3 bf16 / H2 fp16), else fp32 NHWC -> (tensor, name)"""
            if s3 and not f32:
                rg.register(name, key, word_of)
                return ws.get(name, split_shape(fmt, *shape_bhw, c), _SPLIT[fmt][0]), name
            return ws.get(name, tuple(shape_bhw) + (c,)), None

        def dconv(name, src0, h, w, cout, src1=None, pool0=False, pad1=(0, 0), want_pool=False, out_f32=False, head=None):
            """src0 / src1: (tensor, name) pairs"""
            (t0, n0), (t1, _) = src0, (src1 if src1 is not None else (None, None))
            mid, nmid = act(name + ".mid", (B, h, w), L[name + ".0"].cout_real)
            out, nout = act(name + ".out", (B, h, w), cout, f32=out_f32)
            pooled, npool = (act(name + ".pool", (B, h // 2, w // 2), cout, key=nout, word_of=nout)
                             if (want_pool and s3) else (None, None))
            l0, l3 = L[name + ".0"], L[name + ".3"]
            do((nmid,) if nmid else (), lambda: l0.run(t0, B, h, w, mid, src1=t1, pool0=pool0, pad1=pad1, **rg.args(n0, nmid)))
            do((nout,) if nout else (), lambda: l3.run(mid, B, h, w, out, dst_pool=pooled, head=head, **rg.args(nmid, nout)))
            return (out, nout), (pooled, npool)

        # encoder: in bf16x6 mode every Down's MaxPool2d(2) is written by the producer's epilogue;
        # in fp32 mode it is applied while the consumer loads its halo (pool0)
        f0, p0 = dconv("inc", first, H, W, 64, want_pool=True)
        feats, pooled = [f0], [p0]
        h, w = H, W
        for i in range(1, 5):
            h, w = h // 2, w // 2
            src = pooled[-1] if s3 else feats[-1]
            cout = L[f"down{i}.3"].cout_real
            f, p = dconv(f"down{i}", src, h, w, cout, pool0=not s3, want_pool=i < 4)
            feats.append(f)
            pooled.append(p)
        y, ny = feats[4]
        # OutConv (+ the STN input) rides in the epilogue of the last 3x3 conv when nothing else needs y
        logits = torch.empty((B, self.nc, H, W), dtype=torch.float32, device=x.device)
        if want_stn_in and self.nc + 3 > 8:
            raise NotImplementedError("mask_classes > 5 with resnet_input='img+mask' needs a wider STN input buffer")
        stn_in = ws.get("stn_in" if not stn_slot else f"stn_in{stn_slot}", (B, H, W, 8), zero=True) if want_stn_in else None
        head = None
        if (s3 and not want_argmax and not (want_uv and self.outuv is not None)
                and L["up4.conv.3"].cout_real == 64 and os.environ.get("SFH_FUSE_HEAD", "1") != "0"):
            head = {"w": self.outc_w, "b": self.outc_b, "nc": self.nc, "logits": logits, "stn": stn_in,
                    "frame": xin if want_stn_in else None, "skip_dst": True}
        for i in range(1, 5):
            skip, nskip = feats[4 - i]
            hs, ws_ = _hw(skip)
            hy, wy = _hw(y)
            cout = L[f"up{i}.conv.3"].cout_real
            ey, ex = hs - 2 * hy, ws_ - 2 * wy   # F.pad of Up: diff 1 pads one row / column AFTER the tensor
            if f"up{i}.fused" in L and ey in (0, 1) and ex in (0, 1):
                part = ws.get(f"up{i}.part", (B, hs, ws_, L[f"up{i}.skip"].cout_real))     # fp32 partial
                mid, nmid = act(f"up{i}.conv.mid", (B, hs, ws_), L[f"up{i}.fused"].cout_real)
                fu, sk = L[f"up{i}.fused"], L[f"up{i}.skip"]
                up_dst = (hs, ws_) if (ey or ex) else None

                def level(y=y, ny=ny, skip=skip, nskip=nskip, part=part, mid=mid, nmid=nmid, fu=fu, sk=sk,
                          up_dst=up_dst, hs=hs, ws_=ws_, hy=hy, wy=wy, ey=ey, ex=ex, swap=i in self.up_swap,
                          seed=self.up_seed.get(i, False), bands=self.up_bands.get(i, 0), single=i in self.up_single):
                    if swap and seed:
                        # as below, but the partial is written in the skip-half conv's ACCUMULATOR units (divided by its
                        # scale) and that conv STARTS from it (sfh_conv_desc.acc_init): sixteen loads in its prologue
                        # instead of sixteen dependent reads at its end
                        fu.relu, sk.relu = False, True
                        a_fu, a_sk = rg.args(ny, None), rg.args(nskip, nmid)
                        if fu.fmt == "h2":      # bring both scales up to date with the exponents before dividing them
                            fu._fold_exp_src(a_fu["exp_src"])
                            sk._fold_exp_src(a_sk["exp_src"])
                        key = (fu.exp_src, sk.exp_src)
                        if getattr(fu, "_seed_key", None) != key:
                            # (sk.scale is indexed modulo its length: the four sub-positions share it)
                            fu._seed_scale, fu._seed_border, fu._seed_key = (vec_op(fu.scale, sk.scale, "div"),
                                                                             vec_op(fu.shift_border, sk.scale, "div"), key)
                        # frame bands (experiment, SFH_UP_BANDS="4:4" = level 4 in bands of 4 frames): the two launches of a
                        # band run back to back, so that the band's fp32 partial is read back from the Infinity Cache
                        if single and fu.fmt == "h2" and not bands:
                            run_upfused(fu, sk, skip, y, mid, B, hs, ws_, a_sk["exp_dst"], a_sk["range_word"])
                            return
                        nb_ = bands if bands else B
                        for b0 in range(0, B, nb_):
                            b1 = min(B, b0 + nb_)
                            fu.run(y[b0:b1], b1 - b0, hy + ey, wy + ex, part[b0:b1], up_dst=up_dst, scale=fu._seed_scale,
                                   shift_border=fu._seed_border, **a_fu)
                            sk.run(skip[b0:b1], b1 - b0, hs, ws_, mid[b0:b1], acc_init=part[b0:b1], **a_sk)
                    elif swap:
                        # composed 2x2 conv first: it writes the 4 B fp32 partial instead of reading one and writing
                        # 6 B of S3; the MFMA-bound skip-half 3x3 conv then absorbs the residual, the ReLU and the split
                        fu.relu, sk.relu = False, True
                        fu.run(y, B, hy + ey, wy + ex, part, up_dst=up_dst, **rg.args(ny, None))
                        sk.run(skip, B, hs, ws_, mid, residual=part, **rg.args(nskip, nmid))
                    else:
                        fu.relu, sk.relu = True, False
                        sk.run(skip, B, hs, ws_, part, **rg.args(nskip, None))
                        fu.run(y, B, hy + ey, wy + ex, mid, residual=part, up_dst=up_dst, **rg.args(ny, nmid))
                do((nmid,) if nmid else (), level)
                ymid, l3 = mid, L[f"up{i}.conv.3"]
                y, ny = act(f"up{i}.conv.out", (B, hs, ws_), cout, f32=(i == 4))
                do((ny,) if ny else (), lambda ymid=ymid, nmid=nmid, y=y, ny=ny, l3=l3, hs=hs, ws_=ws_, i=i:
                   l3.run(ymid, B, hs, ws_, y, head=head if i == 4 else None, **rg.args(nmid, ny)))
                continue
            cup = _chan(y) if self.bilinear else L[f"up{i}.up"].cout_real
            # the up-sampled tensor is the second source of the conv over cat([skip, up]): it shares the skip's exponent
            upb, nup = act(f"up{i}.up", (B, 2 * hy, 2 * wy), cup, key=nskip)
            if self.bilinear:  # nn.Upsample(2x, bilinear, align_corners=True) on an fp32 view of y
                yf = ws.get(f"up{i}.yf", (B, hy, wy, cup)) if s3 else y
                uf = ws.get(f"up{i}.uf", (B, 2 * hy, 2 * wy, cup)) if s3 else upb

                def upsample(y=y, ny=ny, yf=yf, uf=uf, upb=upb, nup=nup, hy=hy, wy=wy, cup=cup):
                    if s3:
                        _split_to_f32_into(y, yf, rg.exp(ny))
                    _lib.check(lib.sfh_upsample2x_bilinear_nhwc(_ptr(yf), _ptr(uf), B, hy, wy, cup, _stream()), "upsample2x")
                    if s3:
                        _f32_to_split_into(uf, upb, self.overflow, rg.exp(nup), rg.word_ptr(nup))
                do((nup,) if nup else (), upsample)
            else:
                lu = L[f"up{i}.up"]
                do((nup,) if nup else (), lambda y=y, ny=ny, upb=upb, nup=nup, lu=lu, hy=hy, wy=wy:
                   lu.run(y, B, hy, wy, upb, **rg.args(ny, nup)))
            dy, dx = hs - 2 * hy, ws_ - 2 * wy
            (y, ny), _ = dconv(f"up{i}.conv", (skip, nskip), hs, ws_, cout, src1=(upb, nup), pad1=(dy // 2, dx // 2),
                               out_f32=(i == 4), head=head if i == 4 else None)
        out = {"x_top": feats[4][0], "x_top_name": feats[4][1], "y4": y}
        amax = torch.empty((B, H, W), dtype=torch.uint8, device=x.device) if want_argmax else None
        if head is None:
            do((), lambda y=y: _lib.check(
                lib.sfh_outconv_fwd(_ptr(y), 64, _ptr(self.outc_w), _ptr(self.outc_b), self.nc, B, H, W,
                                    _ptr(logits), _ptr(amax), _ptr(stn_in), 8 if want_stn_in else 0,
                                    _ptr(xin) if want_stn_in else None, 4, _stream()), "outconv"))
        else:
            out.pop("y4")   # not materialised: the fused head consumed it in registers
        out["logits"] = logits
        if want_argmax:
            out["argmax"] = amax
        if want_stn_in:
            out["stn_in"] = stn_in
        if want_uv and self.outuv is not None:
            uv = torch.empty((B, 2, H, W), dtype=torch.float32, device=x.device)
            do((), lambda y=y: _lib.check(
                lib.sfh_outconv_fwd(_ptr(y), 64, _ptr(self.outuv[0]), _ptr(self.outuv[1]), 2, B, H, W,
                                    _ptr(uv), None, None, 0, None, 0, _stream()), "outconv(uv)"))
            out["uv"] = uv
        self._last_out = out
        return out

    def x_top_exp(self, out):
        """exponent of out["x_top"] when it is an H2 tensor (for nhwc_to_nchw)"""
        return self.ranges.exp(out.get("x_top_name"))


class StemConv:
    """ResNetSTN stem (7x7 s2 conv + BatchNorm + ReLU) on the tap-packed split-bf16 kernel (csrc/stem.hip):
    reads the fp32 NHWC STN input (8 stored channels) directly, no space-to-depth copy."""

    def __init__(self, conv, bn, cin, tag="resnet", fmt="s3", overflow=None, wexp=None):
        """fmt: arithmetic of the kernel - "s3": the input is split into three bf16 planes, six products; "h2": two
        fp16 planes, three products (overflow: the engine's fp16-range word; wexp: the weight exponent, max |w| * 2^wexp
        in [2^13, 2^14), from a caller that has the maximum already - the training tape reads all of them back in one
        batched synchronisation - else one device read-back here)."""
        lib = _lib.load()
        w = _f32c(conv.weight.detach(), "stem weight")
        if tuple(w.shape) != (64, cin, 7, 7) or cin > 8:
            raise ValueError(f"stem kernel needs a (64, <=8, 7, 7) weight, got {tuple(w.shape)}")
        if fmt not in ("s3", "h2"):
            raise ValueError(f"fmt={fmt!r}: expected 's3' or 'h2'")
        dev = w.device
        self.tag, self.cin, self.fmt, self.overflow = tag, cin, fmt, overflow
        self.exp_src = _lib.H2_ACT_EXP     # h2: the input planes carry x * 2^exp_src (folded into `scale`)
        self.wpacked = torch.empty(lib.sfh_packed_stem_weight_bytes(), dtype=torch.uint8, device=dev)
        self.escale = 1.0
        wexp_given, wexp = wexp, 0
        if fmt == "h2":
            if wexp_given is None:
                wexp_given = h2_weight_exp(absminmax([w])[0][0])
            wexp = max(-100, min(100, int(wexp_given)))
            self.escale = 2.0 ** -(wexp + _lib.H2_ACT_EXP)
        _lib.check(lib.sfh_pack_stem_weights(_ptr(w), _ptr(self.wpacked), cin, _SPLIT[fmt][2], wexp, _stream()),
                   "pack_stem_weights")
        self.relu = bn is not None
        if bn is None:   # training: the raw conv output z (batch-statistics BatchNorm follows as its own pass)
            self.scale = torch.full((64,), float(self.escale), dtype=torch.float32, device=dev)
            self.shift = torch.zeros(64, dtype=torch.float32, device=dev)
            return
        self.scale = torch.empty(64, dtype=torch.float32, device=dev)
        self.shift = torch.empty(64, dtype=torch.float32, device=dev)
        args = [_f32c(t.detach(), "bn tensor") for t in (bn.weight, bn.bias, bn.running_mean, bn.running_var)]
        _lib.check(lib.sfh_fold_bn(None, *[_ptr(a) for a in args], float(bn.eps), 64, 1, _ptr(self.scale),
                                   _ptr(self.shift), _stream()), "fold_bn")
        if self.escale != 1.0:
            vec_op(self.scale, factor=self.escale, out=self.scale)

    def run(self, x_nhwc8, B, H, W, dst, exp_src=None, range_word=None):
        """exp_src / range_word (h2 arithmetic): exponent of the split the kernel makes of its fp32 input, and the
        device word that receives the largest |x * 2^exp_src| (H2Ranges).  Built without a BatchNorm (bn=None) the
        launch writes the raw conv output (no ReLU)."""
        lib = _lib.load()
        d = ConvDesc()
        if self.fmt == "h2" and exp_src is not None and int(exp_src) != self.exp_src:
            f = 2.0 ** (self.exp_src - int(exp_src))
            vec_op(self.scale, factor=f, out=self.scale)
            self.escale *= f
            self.exp_src = int(exp_src)
        d.h2_exp_src = self.exp_src
        d.h2_range = range_word if (range_word and self.fmt == "h2") else None
        d.src0, d.c0, d.cs0, d.h0, d.w0 = x_nhwc8.data_ptr(), self.cin, 8, H, W
        d.batch, d.H, d.W, d.ksize, d.stride = B, H, W, 7, 2
        d.wpacked, d.scale, d.shift = self.wpacked.data_ptr(), self.scale.data_ptr(), self.shift.data_ptr()
        d.cout, d.relu = 64, 1 if self.relu else 0
        d.dst, d.dst_cs, d.out_mode = dst.data_ptr(), dst.shape[3], _lib.OUT_NHWC
        d.src_fmt = d.dst_fmt = _lib.FMT_F32
        d.split_arith = _SPLIT[self.fmt][2]
        d.h2_overflow = self.overflow.data_ptr() if (self.overflow is not None and self.fmt == "h2") else None
        ho, wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        if tuple(dst.shape) != (B, ho, wo, 64) or tuple(x_nhwc8.shape) != (B, H, W, 8):
            raise ValueError(f"stem: shapes {tuple(x_nhwc8.shape)} -> {tuple(dst.shape)} do not match {(B, ho, wo, 64)}")
        tm = PackedConv.timer
        if tm is not None and not tm.wants(self.tag):
            tm = None
        if tm is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        _lib.check(lib.sfh_stem7x7_fwd(ctypes.byref(d), _stream()), "stem7x7_fwd")
        if tm is not None:
            e1.record()
            tm.records.append((self.tag, 2.0 * B * ho * wo * 64 * 49 * self.cin, e0, e1))
        return dst


class ResNetEngine:
    """ResNetSTN forward (models/resnet.py:235-254) on the HIP kernels (BasicBlock and Bottleneck depths)."""

    def __init__(self, rn, in_channels, device, precision="bf16x6", overflow=None, ranges=None):
        """precision "bf16x6" / "f16x3": the 3x3 convs (stride 1 and 2) and the 1x1 stride-2 downsample convs run
        on the split-operand kernel with S3 / H2 activations; the stem stays on the fp32 kernel (or, with at most
        8 input channels, on the tap-packed split-bf16 stem kernel).  ranges / overflow: as UNetEngine."""
        if precision not in PRECISIONS:
            raise ValueError(f"precision={precision!r}: expected one of {sorted(PRECISIONS)}")
        self.fmt = fmt = PRECISIONS[precision]
        self.s3 = fmt is not None
        s3 = self.s3
        self.overflow = overflow if fmt == "h2" else None
        self.ranges = (ranges if ranges is not None else H2Ranges(device)) if fmt == "h2" else _NoRanges()
        self.steps = []
        self._last_out = None
        self.splitk = os.environ.get("SFH_SPLITK", "1") != "0"
        self.device = device
        self.ws = _Workspace(device)
        self.cin = in_channels
        self.cs_in = -(-in_channels // 4) * 4
        if (4 * self.cs_in) % 16:
            self.cs_in = -(-in_channels // 8) * 8
        L = {}
        wx = {}
        if fmt == "h2":       # every weight exponent from ONE batched |w| reduction and one read-back
            ws_ = [p.detach() for p in rn.parameters() if p.dim() == 4 and p.is_contiguous()]
            for w, (mx, _) in zip(ws_, absminmax(ws_)):
                wx[w.data_ptr()] = h2_weight_exp(mx)
        def PC(w, *a, **k):      # (PackedConv with this engine's exponent table behind it)
            return PackedConv(w, *a, wexp=wx.get(w.data_ptr()), **k)
        # the stem stays on the fp32 kernel: the 16-tap split-bf16 instance spills registers and
        # measured 1.59 ms against 0.55 ms
        L["stem"] = PC(rn.conv0.weight, None, rn.bn1, 4, 4 * self.cs_in, stem_cin=in_channels, tag="resnet")
        # bf16x6 mode with <= 8 input channels (every resnet_input mode but img+mask+uv): the tap-packed stem kernel
        self.stem7 = (StemConv(rn.conv0, rn.bn1, in_channels, fmt=fmt, overflow=self.overflow, wexp=wx.get(rn.conv0.weight.data_ptr()))
                      if (s3 and self.cs_in == 8 and rn.conv0.out_channels == 64
                                                                  and os.environ.get("SFH_STEM7", "1") != "0") else None)
        self.blocks = []
        for li in range(1, 5):
            for bi, blk in enumerate(getattr(rn, f"layer{li}")):
                name = f"layer{li}.{bi}"
                cin = blk.conv1.in_channels
                if hasattr(blk, "conv3"):  # Bottleneck (models/resnet.py:120-140): 1x1, 3x3 (stride), 1x1
                    width, cout = blk.conv1.out_channels, blk.conv3.out_channels
                    L[name + ".conv1"] = PC(blk.conv1.weight, None, blk.bn1, 1, cin, tag="resnet", fmt=fmt)
                    L[name + ".conv2"] = PC(blk.conv2.weight, None, blk.bn2, 3, width, stride=blk.stride,
                                                    tag="resnet", fmt=fmt)
                    L[name + ".conv3"] = PC(blk.conv3.weight, None, blk.bn3, 1, width, tag="resnet",
                                                    fmt=fmt)  # ReLU after the residual add
                else:  # BasicBlock (models/resnet.py:64-82)
                    width = cout = blk.conv1.out_channels
                    L[name + ".conv1"] = PC(blk.conv1.weight, None, blk.bn1, 3, cin, stride=blk.stride,
                                                    tag="resnet", fmt=fmt)
                    L[name + ".conv2"] = PC(blk.conv2.weight, None, blk.bn2, 3, width, tag="resnet",
                                                    fmt=fmt)  # ReLU after the residual add
                if blk.downsample is not None:
                    ds = blk.downsample
                    L[name + ".down"] = PC(ds[0].weight, None, ds[1], 1, cin, relu=False, stride=blk.stride,
                                                   tag="resnet", fmt=fmt)
                self.blocks.append((name, width, cout, blk.stride, blk.downsample is not None, hasattr(blk, "conv3")))
        self.L = L
        self.order = LaunchOrder()
        for layer in L.values():
            layer.order = self.order
            layer.overflow = self.overflow
        self.reg_w = snapshot(rn.reg.weight)
        self.reg_b = snapshot(rn.reg.bias)

    def run(self, y_nhwc, B, H, W, splitk=None):
        """y_nhwc: (B,H,W,cs_in) float32 with channels >= cin zero.  Returns theta (B,1,3,3).
        splitk=False: no split-K at small batches for this pass (Reconstructor.predict_async with pipeline_splitk = False:
        beside another batch's UNet the unsplit launches - fewer, without their finish launches - overlap better)."""
        with _stream_scope():
            return self._run(y_nhwc, B, H, W, splitk)

    first_step = UNetEngine.first_step
    rerun = UNetEngine.rerun

    def _run(self, y_nhwc, B, H, W, splitk=None):
        lib = _lib.load()
        ws, L, rg = self.ws, self.L, self.ranges
        use_splitk = self.splitk and (splitk is None or bool(splitk))
        if y_nhwc.shape[3] != self.cs_in:
            raise ValueError(f"STN input has {y_nhwc.shape[3]} stored channels, engine expects {self.cs_in}")
        steps = self.steps = []

        def do(outs, fn):   # as UNetEngine._run
            steps.append((tuple(outs), fn))
            fn()

        s3, fmt = self.s3, self.fmt
        H2, W2 = (H + 1) // 2, (W + 1) // 2
        c1 = ws.get("stem", (B, H2, W2, 64))
        if self.stem7 is not None:
            # the stem kernel splits its fp32 input itself: "rn.stem.in" names that (never stored) split
            rg.register("rn.stem.in")
            do(("rn.stem.in",) if fmt == "h2" else (),
               lambda: self.stem7.run(y_nhwc, B, H, W, c1, exp_src=rg.exp("rn.stem.in"), range_word=rg.word_ptr("rn.stem.in")))
        else:
            s2d = ws.get("s2d", (B, H2, W2, 4 * self.cs_in))
            do((), lambda: _lib.check(lib.sfh_space_to_depth2(_ptr(y_nhwc), _ptr(s2d), B, H, W, self.cs_in, _stream()),
                                      "space_to_depth2"))
            if L["stem"].s3:
                s2d3 = ws.get("s2d.s3", s3_shape(B, H2, W2, 4 * self.cs_in), torch.bfloat16)
                do((), lambda: _lib.check(lib.sfh_f32_to_s3(_ptr(s2d), _ptr(s2d3), B * H2, W2, 4 * self.cs_in, _stream()),
                                          "f32_to_s3"))
                s2d = s2d3
            do((), lambda s2d=s2d: L["stem"].run(s2d, B, H2, W2, c1))
        h, w = (H2 - 1) // 2 + 1, (W2 - 1) // 2 + 1

        def act(name, hh, ww, c):
            if s3:
                rg.register("rn." + name)
                return ws.get(name, split_shape(fmt, B, hh, ww, c), _SPLIT[fmt][0]), "rn." + name
            return ws.get(name, (B, hh, ww, c)), None

        nx = None
        if s3:  # the pooled stem output enters the split domain: pooling and split in one pass (round 4; two launches before)
            x, nx = act("pool.s3", h, w, 64)
            ovf = self.overflow
            do((nx,), lambda x=x, nx=nx: _lib.check(lib.sfh_maxpool3x3s2_split_fwd(
                _ptr(c1), _ptr(x), B, H2, W2, 64, _SPLIT[fmt][2], rg.exp(nx),
                ctypes.c_void_p(ovf.data_ptr()) if (ovf is not None and fmt == "h2") else None,
                ctypes.c_void_p(rg.word_ptr(nx)) if rg.word_ptr(nx) else None, _stream()), "maxpool3x3s2_split"))
        else:
            x = ws.get("pool", (B, h, w, 64))
            do((), lambda x=x: _lib.check(lib.sfh_maxpool3x3s2_fwd(_ptr(c1), _ptr(x), B, H2, W2, 64, _stream()), "maxpool3x3s2"))
        for name, width, cout, stride, has_down, bottleneck in self.blocks:
            ho, wo = (h - 1) // stride + 1, (w - 1) // stride + 1

            def conv(layer, src, nsrc, hh, ww, dst, ndst, residual=None, nres=None):
                pc = L[layer]
                # layer3 / layer4 at batch 16: 60-170 workgroups for 512 slots - split the K loop to fill the chip
                ks, slabs = 1, None
                if s3 and use_splitk:
                    oh, ow = (hh - 1) // pc.stride + 1, (ww - 1) // pc.stride + 1
                    ks = choose_ksplit(B, oh, ow, pc.stride, pc.cout, (pc.c0 + pc.c1) // 32, pc.ksize)
                    if ks > 1:
                        slabs = ws.get(f"slabs{ks}x{oh}x{ow}x{pc.cout}", (ks, B, oh, ow, pc.cout))
                do((ndst,) if ndst else (), lambda: pc.run(src, B, hh, ww, dst, residual=residual, ksplit=ks, slabs=slabs,
                                                           **rg.args(nsrc, ndst, nres)))
            if has_down:
                idn, nidn = act(name + ".idn", ho, wo, cout)
                conv(name + ".down", x, nx, h, w, idn, nidn)
            else:
                idn, nidn = x, nx
            out, nout = act(name + ".out", ho, wo, cout)
            if bottleneck:
                t1, nt1 = act(name + ".t1", h, w, width)
                conv(name + ".conv1", x, nx, h, w, t1, nt1)
                t2, nt2 = act(name + ".t2", ho, wo, width)
                conv(name + ".conv2", t1, nt1, h, w, t2, nt2)
                conv(name + ".conv3", t2, nt2, ho, wo, out, nout, residual=idn, nres=nidn)
            else:
                t, nt = act(name + ".t", ho, wo, width)
                conv(name + ".conv1", x, nx, h, w, t, nt)
                conv(name + ".conv2", t, nt, ho, wo, out, nout, residual=idn, nres=nidn)
            x, nx, h, w = out, nout, ho, wo
        if s3:
            xf = ws.get("final.f32", (B, h, w, _chan(x)))
            do((), lambda x=x, nx=nx, xf=xf: _split_to_f32_into(x, xf, rg.exp(nx)))
            x = xf
        theta = torch.empty((B, 9), dtype=torch.float32, device=x.device)
        pooled = ws.get("pooled", (B, x.shape[3]))
        do((), lambda x=x, h=h, w=w: _lib.check(
            lib.sfh_avgpool_linear_fwd(_ptr(x), _ptr(self.reg_w), _ptr(self.reg_b), B, h, w, x.shape[3], 9,
                                       _ptr(pooled), _ptr(theta), _stream()), "avgpool_linear"))
        self._last_out = theta.view(B, 1, 3, 3)
        return self._last_out


def _split_to_f32_into(t, out, exp=_lib.H2_ACT_EXP):
    """exp: exponent of an H2 tensor (ignored for S3)"""
    lib = _lib.load()
    B = t.shape[0]
    H, W = _hw(t)
    C = _chan(t)
    if tuple(out.shape) != (B, H, W, C) or out.dtype != torch.float32:
        raise ValueError(f"split_to_f32: destination {tuple(out.shape)} does not match {(B, H, W, C)}")
    if _fmt_of(t) == "h2":
        _lib.check(lib.sfh_h2_to_f32(_ptr(t), _ptr(out), B * H, W, C, int(exp), _stream()), "h2_to_f32")
    else:
        _lib.check(lib.sfh_s3_to_f32(_ptr(t), _ptr(out), B * H, W, C, _stream()), "s3_to_f32")
    return out


def _f32_to_split_into(t, out, overflow=None, exp=_lib.H2_ACT_EXP, range_word=None):
    """H2 destinations: exp = the tensor's exponent, overflow / range_word: optional device words (OR 1 on
    saturation / atomic max of |v * 2^exp|, see H2Ranges)"""
    lib = _lib.load()
    B, H, W, C = t.shape
    if (out.shape[0],) + _hw(out) + (_chan(out),) != (B, H, W, C):
        raise ValueError(f"f32_to_split: destination {tuple(out.shape)} does not match {(B, H, W, C)}")
    if _fmt_of(out) == "h2":
        _lib.check(lib.sfh_f32_to_h2(_ptr(t), _ptr(out), B * H, W, C, int(exp), _ptr(overflow),
                                     ctypes.c_void_p(range_word) if range_word else None, _stream()), "f32_to_h2")
    else:
        _lib.check(lib.sfh_f32_to_s3(_ptr(t), _ptr(out), B * H, W, C, _stream()), "f32_to_s3")
    return out


def s3_to_f32(t, exp=_lib.H2_ACT_EXP):
    """split tensor (S3: (B,H,C/32,3,4,W,8) bf16, exact sum of the planes; H2: (B,H,C/32,2,4,W,8) fp16 carrying
    v * 2^exp) -> (B,H,W,C) float32."""
    out = torch.empty((t.shape[0],) + _hw(t) + (_chan(t),), dtype=torch.float32, device=t.device)
    return _split_to_f32_into(t, out, exp)


def split_empty(fmt, b, h, w, c, device):
    """uninitialised split-format activation tensor ("s3" or "h2") for c channels (c multiple of 32)"""
    return torch.empty(split_shape(fmt, b, h, w, c), dtype=_SPLIT[fmt][0], device=device)


def f32_to_split(t, fmt, overflow=None, exp=_lib.H2_ACT_EXP):
    """(B,H,W,C) float32 -> split tensor of format "s3" or "h2" (h2: carrying v * 2^exp) """
    t = _f32c(t, "nhwc tensor")
    return _f32_to_split_into(t, split_empty(fmt, *t.shape, t.device), overflow, exp)


def f32_to_h2(t, overflow=None, exp=_lib.H2_ACT_EXP, range_word=None):
    """(B,H,W,C) float32 -> (B,H,C/32,2,4,W,8) fp16 two-plane tensor of v * 2^exp (include/sfh_amd.h, SFH_FMT_H2);
    range_word: an int32 tensor whose first word receives the largest bit pattern of |v * 2^exp|."""
    t = _f32c(t, "nhwc tensor")
    B, H, W, C = t.shape
    out = torch.empty(split_shape("h2", B, H, W, C), dtype=torch.float16, device=t.device)
    return _f32_to_split_into(t, out, overflow, exp, range_word.data_ptr() if range_word is not None else None)


def f32_to_s3(t):
    """(B,H,W,C) float32 -> (B,H,C/32,3,4,W,8) bf16 split tensor."""
    lib = _lib.load()
    t = _f32c(t, "nhwc tensor")
    B, H, W, C = t.shape
    out = s3_empty(B, H, W, C, t.device)
    _lib.check(lib.sfh_f32_to_s3(_ptr(t), _ptr(out), B * H, W, C, _stream()), "f32_to_s3")
    return out


_AREA_TABS = {}


def _area_tab(ssize, dsize, device):
    """device copies of one axis' INTER_AREA table (sfh_resize_area_tab: OpenCV's computeResizeAreaTab), cached per size pair"""
    key = (ssize, dsize, str(device))
    t = _AREA_TABS.get(key)
    if t is None:
        import numpy as np
        lib = _lib.load()
        cap = 2 * dsize + ssize
        ofs, si, al = np.zeros(dsize + 1, np.int32), np.zeros(cap, np.int32), np.zeros(cap, np.float32)
        n = lib.sfh_resize_area_tab(ssize, dsize, ofs.ctypes.data_as(ctypes.c_void_p), si.ctypes.data_as(ctypes.c_void_p),
                                    al.ctypes.data_as(ctypes.c_void_p), cap)
        if n < 0:
            raise ValueError(f"no INTER_AREA table for {ssize} -> {dsize}")
        t = _AREA_TABS[key] = tuple(torch.from_numpy(a).to(device) for a in (ofs, si[:max(n, 1)].copy(), al[:max(n, 1)].copy()))
    return t


def frames_u8_to_input(frames_u8, target_size=None):
    """uint8 (B,H,W,C) decoded frames on the GPU -> float32 (B,C,H,W) in [0,1], bit-identical to the
    reference dataset's `img.transpose((2,0,1)) / 255` (utils/dataset.py:154-159).  target_size = (W, H):
    like VideoDataset.preprocess_img (utils/dataset.py:310-330) frames WIDER than the target are resized first with
    cv2.INTER_AREA's rules: the integer factors 2 .. 16 take OpenCV's block-average fast paths (1280x720 -> 640x360 is the
    2x2 special case, 1920x1080 -> 640x360 the 3x3 one), any other downscale (both factors >= 1, e.g. 1920x1080 -> 1024x576
    or 1600x900 -> 640x360) the generic area tables (round 5).  Frames narrower than the target (the reference switches to
    INTER_LINEAR there) are not on the HIP path."""
    lib = _lib.load()
    if frames_u8.dtype != torch.uint8 or frames_u8.dim() != 4 or not frames_u8.is_cuda:
        raise ValueError("expected a uint8 (B,H,W,C) tensor on the GPU")
    f = frames_u8.contiguous()
    B, H, W, C = f.shape
    if target_size is not None and (int(target_size[0]), int(target_size[1])) != (W, H):
        tw, th = int(target_size[0]), int(target_size[1])
        if tw <= 0 or th <= 0 or W <= tw or H < th:      # (W > tw: the reference's own test for INTER_AREA, utils/dataset.py:314)
            raise NotImplementedError(f"GPU frame resize {W}x{H} -> {tw}x{th}: only downscales (cv2.INTER_AREA, the reference's choice "
                                      "for frames wider than the target) are on the HIP path; resize on the host as utils/dataset.py does")
        out = torch.empty((B, C, th, tw), dtype=torch.float32, device=f.device)
        k = W // tw
        if 2 <= k <= 16 and (k * tw, k * th) == (W, H):
            _lib.check(lib.sfh_u8hwc_areak_to_f32nchw(_ptr(f), _ptr(out), B, C, th, tw, k, _stream()), "u8hwc_areak_to_f32nchw")
            return out
        if (tw * (W // tw), th * (H // th)) == (W, H):
            # integer factors that differ per axis, or beyond 16: OpenCV's resizeAreaFast_ with a kx x ky block
            kx, ky = W // tw, H // th
            if kx > 64 or ky > 64:
                raise NotImplementedError(f"GPU frame resize {W}x{H} -> {tw}x{th}: integer factors beyond 64 are not on the HIP path")
            _lib.check(lib.sfh_u8hwc_areaxy_to_f32nchw(_ptr(f), _ptr(out), B, C, th, tw, kx, ky, _stream()), "u8hwc_areaxy_to_f32nchw")
            return out
        xo, xs, xa = _area_tab(W, tw, f.device)
        yo, ys, yb = _area_tab(H, th, f.device)
        _lib.check(lib.sfh_u8hwc_area_to_f32nchw(_ptr(f), _ptr(out), B, C, H, W, th, tw, _ptr(xo), _ptr(xs), _ptr(xa), _ptr(yo),
                                                 _ptr(ys), _ptr(yb), _stream()), "u8hwc_area_to_f32nchw")
        return out
    out = torch.empty((B, C, H, W), dtype=torch.float32, device=f.device)
    _lib.check(lib.sfh_u8hwc_to_f32nchw(_ptr(f), _ptr(out), B, C, H, W, _stream()), "u8hwc_to_f32nchw")
    return out


def resize_nchw(t, size_hw, mode, align_corners=False):
    """F.interpolate(t, size=size_hw, mode=mode[, align_corners]) for NCHW float32 tensors."""
    lib = _lib.load()
    t = _f32c(t.contiguous(), "nchw tensor")
    B, C, hs, ws = t.shape
    hd, wd = size_hw
    out = torch.empty((B, C, hd, wd), dtype=torch.float32, device=t.device)
    _lib.check(lib.sfh_resize_nchw(_ptr(t), _ptr(out), B * C, hs, ws, hd, wd, 1 if mode == "bilinear" else 0,
                                   1 if align_corners else 0, _stream()), "resize_nchw")
    return out


def nhwc_to_nchw(t, channels=None, exp=_lib.H2_ACT_EXP):
    if t.dtype in _SPLIT_DTYPES:
        t = s3_to_f32(t, exp)
    lib = _lib.load()
    B, H, W, cs = t.shape
    C = cs if channels is None else channels
    out = torch.empty((B, C, H, W), dtype=torch.float32, device=t.device)
    _lib.check(lib.sfh_nhwc_to_nchw(_ptr(t), _ptr(out), B, C, H, W, cs, _stream()), "nhwc_to_nchw")
    return out


def nchw_to_nhwc(t, cs=None):
    lib = _lib.load()
    t = _f32c(t, "nchw tensor")
    B, C, H, W = t.shape
    cs = cs or -(-C // 4) * 4
    out = torch.empty((B, H, W, cs), dtype=torch.float32, device=t.device)
    _lib.check(lib.sfh_nchw_to_nhwc(_ptr(t), _ptr(out), B, C, H, W, cs, _stream()), "nchw_to_nhwc")
    return out


def homography_warp(theta, template, h, w, nearest, scale=None, want_f32=True, want_i32=False,
                    shared_template=False):
    """theta (B,1,3,3)|(B,3,3); template (>=B,1,ht,wt).  Returns (f32 or None, i32 or None)."""
    lib = _lib.load()
    theta = _f32c(theta.reshape(-1, 3, 3).contiguous(), "theta")
    template = _f32c(template, "court template")
    B = theta.shape[0]
    if template.dim() != 4 or template.shape[1] != 1:
        raise ValueError(f"court template must be (B,1,H,W), got {tuple(template.shape)}")
    if template.shape[0] < B and not shared_template:
        raise ValueError(f"batch {B} exceeds the court template batch {template.shape[0]}")
    ht, wt = template.shape[2], template.shape[3]
    out_f = torch.empty((B, h, w), dtype=torch.float32, device=theta.device) if want_f32 else None
    out_i = torch.empty((B, h, w), dtype=torch.int32, device=theta.device) if want_i32 else None
    bstride = 0 if shared_template else ht * wt
    tm = PackedConv.timer
    if tm is not None and not tm.wants("warp"):
        tm = None
    if tm is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    _lib.check(lib.sfh_homography_warp_fwd(_ptr(theta), _ptr(template), bstride, ht, wt, B, h, w,
                                           0 if nearest else 1, float(scale if scale is not None else 1.0),
                                           _ptr(out_f), _ptr(out_i), _stream()), "homography_warp")
    if tm is not None:
        e1.record()
        # algorithmic BYTES: every output once, the template once (per frame if not shared), theta
        nout = (1 if want_f32 else 0) + (1 if want_i32 else 0)
        tm.records.append(("warp", float(B * h * w * 4 * nout + (1 if shared_template else B) * ht * wt * 4 + 36 * B), e0, e1))
    return out_f, out_i


def warp_consistency(theta, template, logits, scale, shared_template=False, warp_hw=None):
    """predict()'s nearest warp (* mask_classes -> int32) AND the consistency score fused (sfh_warp_consistency_fwd): 4 classes;
    the warp (warp_hw = (h, w), default the logits' size) has the logits' size or exactly twice it in both directions
    (predict.py's default geometry: the score then goes through the nearest-resized mask, as the reference's).
    -> (warp_mask int32 (B,h,w), score float32 (B,)); the mask is bit-identical to homography_warp()'s."""
    lib = _lib.load()
    theta = _f32c(theta.reshape(-1, 3, 3).contiguous(), "theta")
    template = _f32c(template, "court template")
    logits = _f32c(logits, "logits")
    B, nc, hl, wl = logits.shape
    h, w = (hl, wl) if warp_hw is None else (int(warp_hw[0]), int(warp_hw[1]))
    if (h, w) not in ((hl, wl), (2 * hl, 2 * wl)):
        raise ValueError(f"warp {w}x{h} against logits {wl}x{hl}: the fused kernel takes the same size or exactly twice it")
    if theta.shape[0] != B:
        raise ValueError(f"{theta.shape[0]} homographies for {B} frames of logits")
    if template.dim() != 4 or template.shape[1] != 1:
        raise ValueError(f"court template must be (B,1,H,W), got {tuple(template.shape)}")
    if template.shape[0] < B and not shared_template:
        raise ValueError(f"batch {B} exceeds the court template batch {template.shape[0]}")
    ht, wt = template.shape[2], template.shape[3]
    dev = theta.device
    out_i = torch.empty((B, h, w), dtype=torch.int32, device=dev)
    partial = torch.empty(lib.sfh_warp_consistency_workspace_floats(B, h, w), dtype=torch.float32, device=dev)
    score = torch.empty(B, dtype=torch.float32, device=dev)
    tm = PackedConv.timer
    if tm is not None and not tm.wants("warp+ce"):
        tm = None
    if tm is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    _lib.check(lib.sfh_warp_consistency_fwd(_ptr(theta), _ptr(template), 0 if shared_template else ht * wt, ht, wt, B, h, w,
                                            float(scale), _ptr(logits), nc, hl, wl, _ptr(out_i), _ptr(partial), _ptr(score),
                                            _stream()), "warp_consistency")
    if tm is not None:
        e1.record()
        # algorithmic BYTES: the logits once, the mask once, the template once (per frame if not shared), theta
        tm.records.append(("warp+ce", float(B * (hl * wl * 4 * nc + h * w * 4) + (1 if shared_template else B) * ht * wt * 4 + 36 * B), e0, e1))
    return out_i, score


def poi_project(theta, poi, normalize=True):
    lib = _lib.load()
    theta = _f32c(theta.reshape(-1, 3, 3).contiguous(), "theta")
    B = theta.shape[0]
    if poi.shape[0] < B:
        raise ValueError(f"batch {B} exceeds the court POI batch {poi.shape[0]}")
    p = _f32c(poi[:B].contiguous(), "court_poi")
    out = torch.empty_like(p)
    _lib.check(lib.sfh_poi_project_fwd(_ptr(theta), _ptr(p), B, p.shape[1], 1 if normalize else 0,
                                       _ptr(out), _stream()), "poi_project")
    return out


def consistency_ce(logits, mask_i32):
    lib = _lib.load()
    logits = _f32c(logits, "logits")
    B, nc, H, W = logits.shape
    if mask_i32.dtype != torch.int32 or not mask_i32.is_contiguous():
        raise ValueError("warp mask must be a contiguous int32 tensor")
    hm, wm = mask_i32.shape[1], mask_i32.shape[2]
    partial = torch.empty(lib.sfh_ce_workspace_floats(B, H, W), dtype=torch.float32, device=logits.device)
    score = torch.empty(B, dtype=torch.float32, device=logits.device)
    _lib.check(lib.sfh_consistency_ce_fwd(_ptr(logits), _ptr(mask_i32), B, nc, H, W, hm, wm, _ptr(partial),
                                          _ptr(score), _stream()), "consistency_ce")
    return score
