"""Drop-in mirror of the reference's ``models.Reconstructor`` (models/reconstructor.py:30-247).

Same constructor keywords, attributes, method names, returned dict keys, dtypes and
``state_dict`` layout; the arithmetic runs in the HIP kernels of ``libsfh_amd.so``
(see engine.py).  There is no fallback: calling the model on CPU tensors, or without
the built library, raises.

Training (``net.train(); net(x); loss.backward()``) runs on the HIP kernels of
``training.py`` for the default configuration; ``predict`` stays eval-only like a BatchNorm model should.
"""
import os
from enum import Enum

import torch
import torch.nn as nn

from . import engine as E
from .modules import DoubleConv, Down, Up, OutConv, resnet_stn


class Input(Enum):
    """ResNet input selection (reference: models/reconstructor.py:9-28)."""
    IMG = 1
    MASK = 2
    IMG_AND_MASK = 3
    IMG_AND_MASK_AND_UV = 4

    @classmethod
    def parse(cls, input):
        table = {"img": cls.IMG, "mask": cls.MASK, "img+mask": cls.IMG_AND_MASK,
                 "img+mask+uv": cls.IMG_AND_MASK_AND_UV}
        if input is None:
            return None
        if input not in table:
            raise NotImplementedError
        return table[input]


# Parameter / buffer / sub-module registrations anywhere in the process (torch.nn global hooks): Reconstructor caches the
# list of its tensors for the engine stamp and rebuilds it when this counter moved.
_REGISTRATIONS = [0]


def _count_registration(module, name, value):
    _REGISTRATIONS[0] += 1
    return None


torch.nn.modules.module.register_module_parameter_registration_hook(_count_registration)
torch.nn.modules.module.register_module_buffer_registration_hook(_count_registration)
torch.nn.modules.module.register_module_module_registration_hook(_count_registration)


# predict_async()'s two extra streams, ONE pair per device for every model of the process: HIP multiplexes streams onto a handful
# of hardware queues (four by default) in creation order, and streams that share a queue serialise - a pair per model object
# would make the overlap of every second or third model a process creates depend on which queue its side stream happened to land
# on (sfh_amd.pipeline measured exactly that with per-object copy streams: profiles/micro/e2e_parity_probe.py).
_PIPE_STREAMS = {}


def _pipe_streams(dev, prio):
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), prio)
    st = _PIPE_STREAMS.get(key)
    if st is None:
        st = _PIPE_STREAMS[key] = (torch.cuda.Stream(dev, priority=prio), torch.cuda.Stream(dev))
    return st


class _Done:
    """handle of a batch that was computed synchronously"""

    def __init__(self, out):
        self._out = out

    def result(self):
        return self._out


class _Pending:
    """handle of a batch in flight (Reconstructor.predict_async)"""

    def __init__(self, net, x, consistency, project_poi, out, done):
        self.net, self.x, self.args, self.out, self.done = net, x, (consistency, project_poi), out, done
        self.stale = False
        self._final = None

    def result(self):
        """predict()'s dict; the caller's current stream is ordered behind the batch"""
        if self._final is not None:
            return self._final
        net = self.net
        if not self.stale and net._pipe_check(self):
            cur = torch.cuda.current_stream(self.x.device)
            cur.wait_event(self.done)
            for t in self.out.values():
                t.record_stream(cur)
            self._final = self.out
        else:      # its range check (or an earlier batch's) found a saturated tensor: recompute with the fixed exponents
            self._final = net.predict(self.x, consistency=self.args[0], project_poi=self.args[1])
        self.out = self.x = None
        return self._final


class Reconstructor(nn.Module):
    """UNet segmentation + ResNet-STN homography regression + court-template warp."""

    def __init__(self, court_img, court_poi,
                 target_size=(640, 360),
                 mask_classes=4,
                 use_unet=True,
                 unet_bilinear=False,
                 unet_size=(640, 360),
                 unet_uv=False,
                 use_resnet=True,
                 resnet_name='resnet34',
                 resnet_input='img+mask',
                 resnet_pretrained=None,
                 use_warper=True,
                 warp_size=(640, 360),
                 warp_with_nearest=False):
        super().__init__()
        assert use_unet is not None or use_resnet is not None
        # plain attributes, not buffers: they are absent from the state_dict
        # (reference: models/reconstructor.py:55-56)
        self.court_img = court_img
        self.court_poi = court_poi
        self.target_size = target_size
        self.mask_classes = mask_classes
        self.use_unet = use_unet
        self.unet_size = unet_size
        self.unet_uv = unet_uv
        self.unet_bilinear = bool(unet_bilinear)
        self.use_resnet = use_resnet
        self.resnet_input = Input.parse(resnet_input)
        self.resnet_name = resnet_name
        self.warp_size = warp_size
        self.warp_with_nearest = warp_with_nearest is True

        if self.use_unet:
            factor = 2 if unet_bilinear else 1
            self.inc = DoubleConv(3, 64)
            self.down1 = Down(64, 128)
            self.down2 = Down(128, 256)
            self.down3 = Down(256, 512)
            self.down4 = Down(512, 1024 // factor)
            self.up1 = Up(1024, 512 // factor, unet_bilinear)
            self.up2 = Up(512, 256 // factor, unet_bilinear)
            self.up3 = Up(256, 128 // factor, unet_bilinear)
            self.up4 = Up(128, 64, unet_bilinear)
            self.outc = OutConv(64, mask_classes)
            self.outuv = OutConv(64, 2) if unet_uv else None

        if self.use_resnet:
            assert self.resnet_input is not None
            if self.resnet_input == Input.IMG:
                in_classes = 3
            elif self.resnet_input == Input.MASK:
                assert self.use_unet
                in_classes = mask_classes
            elif self.resnet_input == Input.IMG_AND_MASK:
                assert self.use_unet
                in_classes = mask_classes + 3
            elif self.resnet_input == Input.IMG_AND_MASK_AND_UV:
                assert self.use_unet and self.unet_uv
                in_classes = mask_classes + 3 + 2
            else:
                assert False
            self._stn_in_channels = in_classes
            self.resnet_reg = resnet_stn(resnet_name, resnet_pretrained, in_classes)

        # The reference holds a kornia HomographyWarper here (parameter-free, so it adds
        # no state_dict entries); the HIP warp kernel needs only the flag.
        self.warper = True if use_warper else None
        self._warp_hw = (warp_size[1], warp_size[0])

        # Arithmetic of the conv stack (fp32 accumulation in all modes, see csrc/conv_s3.hip):
        #   "f16x3" (default) = two-plane fp16 activations / weights (22 significand bits), three fp16 MFMAs per
        #                       product; every activation tensor carries its own power-of-two exponent (2 at first:
        #                       |v| < 16376); a tensor that leaves its range is detected on the device, its exponent
        #                       is lowered for good and the pass is repeated from that layer (see _guarded);
        #   "bf16x6"          = three-plane bf16 operands (the exact fp32 value), six bf16 MFMAs per product;
        #   "fp32"            = fp32 MFMA throughout.
        # Plain attribute (or env SFH_PRECISION) so the constructor signature stays the reference's.
        self.precision = os.environ.get("SFH_PRECISION", "f16x3")
        self._engines = None       # (UNetEngine | None, ResNetEngine | None)
        self._engines_by_precision = {}
        self._h2_ranges = None     # engine.H2Ranges: exponents + device range words of the "f16x3" activations
        self._forced_precision = None
        self.range_rescales = 0    # passes repeated from a layer whose output left its fp16 range (exponent lowered)
        self.range_raises = 0      # passes repeated from a layer whose output sat at the bottom of its range (exponent raised)
        self.range_fallbacks = 0   # batches re-run in "bf16x6" because an activation was not finite
        # False: predict() / forward() skip the 4-byte read-back (a device synchronisation) after every call; a
        # caller that pipelines several batches then asks range_overflowed() itself once it has synchronised
        self.range_guard = True
        # predict(consistency=True) with a nearest warp of the logits' size: warp + consistency CE as one kernel (False: the two
        # separate kernels; env SFH_FUSE_WARP_CE=0)
        self.fuse_warp_ce = os.environ.get("SFH_FUSE_WARP_CE", "1") != "0"
        # predict() through predict_replay() (one HIP-graph launch per batch, same bits) for batches of at most
        # graph_replay_max_batch frames; off by default: it returns the caller's thread, not GPU time
        self.graph_replay = False
        self.graph_replay_max_batch = 8
        # predict_async() at small batches: True = the ResNet-STN launches split their K loops exactly as predict() does (same
        # bits as predict()); False = unsplit on the side stream - beside the other batch's UNet the fewer, finish-free launches
        # overlap a little better (same-device A/B, profiles/r06_ab_pipeline_splitk.txt: 2 frames per call 2.23 -> 2.13 ms,
        # 4 frames 3.83 -> 3.72 ms, 1 and >= 8 frames +-0) - at the price of theta differing from predict()'s in the last
        # bits (another fp32 summation order); batches of 16 never split, so the headline is untouched either way
        self.pipeline_splitk = True
        self._engine_stamp = None
        self._weights_generation = 0
        self._tmpl_shared = None   # (data_ptr, shape) -> bool cache

    # ------------------------------------------------------------------ engine plumbing
    def _param_stamp(self):
        # walking the module tree costs ~1 ms per call (354 tensors behind ~200 modules), more than enqueueing the
        # whole UNet: the tensor list is collected once.  load_state_dict() and optimizers change these tensors in
        # place (their torch _version moves); .to() / .half() on the model OR ON A SUB-MODULE give every Parameter a
        # new storage (its data_ptr moves) and replace the buffers; a Parameter / buffer / sub-module that is REPLACED
        # anywhere in the process registers with torch's module hooks, which advance _REGISTRATIONS.  Either of the
        # last two makes this walk the tree again - and only a list that really differs forces new engines
        # (constructing an unrelated module elsewhere, e.g. a loss with a weight buffer, changes nothing here).
        lst = self.__dict__.get("_stamp_tensors")
        ptrs = None
        if lst is not None:
            ptrs = hash(tuple([t.data_ptr() for t in lst]))     # order-sensitive: two storage moves cannot cancel
        if (lst is None or self.__dict__.get("_stamp_regs") != _REGISTRATIONS[0]
                or ptrs != self.__dict__.get("_stamp_ptrs")):
            new = list(self.parameters()) + list(self.buffers())
            if lst is None or len(new) != len(lst) or any(a is not b for a, b in zip(new, lst)) \
                    or ptrs != self.__dict__.get("_stamp_ptrs"):
                self._weights_generation += 1      # a replaced tensor may carry any _version: force a new stamp
            lst = self.__dict__["_stamp_tensors"] = new
            self.__dict__["_stamp_regs"] = _REGISTRATIONS[0]
            self.__dict__["_stamp_ptrs"] = hash(tuple([t.data_ptr() for t in lst]))
        dev = None
        ver = 0
        for t in lst:
            ver += t._version
        if lst:
            dev = lst[-1].device
        return (dev, ver, self.training, self.precision, self._weights_generation)

    def _apply(self, fn, *args, **kwargs):
        # .to() / .cuda() / .float(): buffers are REPLACED by nn.Module._apply - drop the cached tensor list
        self.__dict__.pop("_stamp_tensors", None)
        return super()._apply(fn, *args, **kwargs)

    def load_state_dict(self, *args, **kwargs):
        self.__dict__.pop("_stamp_tensors", None)      # assign=True replaces the tensors
        return super().load_state_dict(*args, **kwargs)

    def invalidate_engines(self):
        """Packed weights and folded BatchNorm constants are cached per engine and rebuilt when a parameter's
        torch ``_version`` moves.  Kernels that write parameters or buffers through raw device pointers
        (training.TrainStep.step, the train-mode BatchNorm statistics) do not move it: they call this."""
        self._weights_generation += 1
        self._engines = None
        self._engines_by_precision = {}
        self.__dict__.pop("_stamp_tensors", None)

    def _get_engines(self):
        stamp = self._param_stamp()
        if self._engine_stamp != stamp:
            if self.__dict__.get("_retiring"):
                # re-entered from the drain below (a batch in flight failed its range check and is recomputed): the OLD
                # engines - they hold packed copies of the old weights - finish the old batches
                eng = self._engines_by_precision.get(self._forced_precision or self.precision)
                if eng is None:
                    raise RuntimeError("a predict_async() batch in flight when the weights changed needs an engine of the old "
                                       "weights that was never built (three-plane fallback); take its result() before "
                                       "load_state_dict / an optimizer step")
                self._engines = eng
                return eng
            old = self._engine_stamp
            if old is not None:
                # batches of predict_async() still in flight were computed with the old engines: take their results (and
                # their range checks) while engines, stamp and range words are still the old ones
                self.__dict__["_retiring"] = True
                try:
                    self._on_new_weights()
                finally:
                    self.__dict__["_retiring"] = False
            self._engines_by_precision = {}
            self._engine_stamp = stamp
            self.__dict__.pop("_replay", None)      # captured graphs launch the old engines' buffers and packed weights
            if old is not None and self._h2_ranges is not None:
                self._h2_ranges.new_generation()
        precision = self._forced_precision or self.precision
        eng = self._engines_by_precision.get(precision)
        if eng is None:
            dev = stamp[0]
            if dev is None or dev.type != "cuda":
                raise RuntimeError(
                    f"Reconstructor parameters are on {dev}: move the model to the GPU with .to('cuda'); "
                    "the HIP path has no CPU fallback")
            with torch.cuda.device(dev):
                rg = None
                if precision == "f16x3":
                    if self._h2_ranges is None or self._h2_ranges.device != dev:
                        self._h2_ranges = E.H2Ranges(dev)
                    rg = self._h2_ranges
                un = E.UNetEngine(self, dev, precision, ranges=rg) if self.use_unet else None
                rn = (E.ResNetEngine(self.resnet_reg, self._stn_in_channels, dev, precision, ranges=rg)
                      if self.use_resnet else None)
            eng = self._engines_by_precision[precision] = (un, rn)
        self._engines = eng
        return eng

    def _on_new_weights(self):
        """The stamp is about to move (new weights, another mode): batches of predict_async() still in flight were computed
        with the old engines - take their results and their range checks now, BEFORE the engines are dropped (a batch whose
        check fails is recomputed by the old engines, never with the new weights).  Afterwards the caller starts a new
        generation of the range words (engine.H2Ranges.new_generation)."""
        p = self.__dict__.get("_pipe")
        if p is not None and p["inflight"]:
            for h in list(p["inflight"]):
                h.result()

    @staticmethod
    def _run_phases(phases):
        run_unet, run_stn, tail = phases
        r = run_unet()
        return tail(r, run_stn(r))

    def _guarded(self, make_phases, x, off):
        """One forward pass = three phases, make_phases(x, off) -> (run_unet(resume=None) -> r,
        run_stn(r, resume=None) -> theta, tail(r, theta) -> dict).  In "f16x3" mode the kernels leave the largest
        magnitude of every activation tensor in a device word (engine.H2Ranges); ONE read-back per call tells whether
        a tensor left the fp16 range of its exponent.  If so, that tensor's exponent is lowered for good (8x headroom
        over what was seen) and the pass is repeated FROM THE LAUNCH THAT WRITES IT, with everything upstream kept
        (`range_rescales`) - the model stays on the two-plane fp16 path.  Only a non-finite activation (a NaN / Inf
        frame) sends the batch through the three-plane bf16 operands, which carry NaN / Inf to the outputs as the
        reference does (`range_fallbacks`).  Still the HIP path - there is no CPU fallback."""
        phases = make_phases(x, off)
        run_unet, run_stn, tail = phases
        r = run_unet()
        theta = run_stn(r)
        ret = tail(r, theta)
        rg = self._h2_ranges
        if (self._forced_precision or self.precision) != "f16x3" or rg is None or not self.range_guard:
            return ret
        un, rn = self._engines

        def three_plane_rerun():
            self._forced_precision = "bf16x6"
            try:
                return self._chunked(lambda xi, o: self._run_phases(make_phases(xi, o)), x, base=off)
            finally:
                self._forced_precision = None

        for _ in range(256):
            bits = rg.read()                     # one read-back of ~100 words per call
            bad, nonfinite = rg.saturated(bits)
            plan = None if bad else rg.quiet(bits)
            if not bad and not plan:
                return ret
            # launches of predict_async() batches still running on the side stream were enqueued with the exponents as
            # they are now and write the same words: let them finish before the words are zeroed and the exponents move
            self._pipe_drain()
            rg.reset_words()
            self._pipe_mark_stale()              # batches of predict_async() in flight wrote the same words
            if plan:
                keys = rg.raise_(plan)           # quiet tensors: larger exponents (one decision per key)
                self.range_raises += 1
            elif nonfinite:
                self.range_fallbacks += 1
                if self.range_fallbacks == 1:
                    import warnings
                    warnings.warn("sfh_amd: a non-finite activation in the 'f16x3' mode; the batch was re-run with "
                                  "precision 'bf16x6', which carries NaN / Inf to the outputs like the reference")
                return three_plane_rerun()
            else:
                try:
                    keys = rg.lower(bad, bits)   # one decision per exponent key
                except E.FP16RangeExhausted:
                    self.range_fallbacks += 1    # finite, but beyond 2^64 * 65504: outside the two-plane format for good
                    return three_plane_rerun()
                self.range_rescales += 1
            k = un.first_step(keys) if un is not None else None
            if k is not None:
                r = run_unet(resume=k)
                theta = run_stn(r)
            else:
                k = rn.first_step(keys) if rn is not None else None
                if k is None:
                    continue                     # a tensor of an earlier call shape: nothing of this pass depends on it
                theta = run_stn(r, resume=k)
            ret = tail(r, theta)
        raise RuntimeError("sfh_amd: the fp16 range guard did not converge")

    def _pipe_mark_stale(self):
        """The range words are shared by every pass of this model.  Whoever finds a saturated tensor and zeroes the words
        must also invalidate the predict_async() batches still in flight: their own check would read zeros and hand
        out clamped outputs as valid.  Their result() then recomputes them with the exponents as they are by then."""
        p = self.__dict__.get("_pipe")
        if p is not None and p["inflight"]:
            for o in p["inflight"]:
                o.stale = True
            p["inflight"].clear()

    def _pipe_drain(self):
        """wait for the predict_async() batches in flight (their side-stream launches write the shared range words)"""
        p = self.__dict__.get("_pipe")
        if p is not None and p["inflight"]:
            torch.cuda.synchronize(p["device"])

    def _pipe_wait_reads(self):
        """Batches of predict_async() may still be reading the STN-input buffers and the ResNet workspace on the side
        stream: order the caller's stream behind them before a synchronous pass writes those buffers."""
        p = self.__dict__.get("_pipe")
        if p is not None:
            for ev in p["stem_read"]:
                if ev is not None:
                    torch.cuda.current_stream(p["device"]).wait_event(ev)

    def range_overflowed(self, reset=True):
        """For callers that pipeline batches with `range_guard = False`: True if an activation tensor of the "f16x3"
        mode was saturated since the last reset - or, the other direction, if every tensor of an exponent key stayed at
        the bottom of its range, where elements lose bits (synchronises).  With reset, the exponents of those tensors
        are lowered / raised so that the following batches fit; the caller re-submits the batches it had in flight."""
        rg = self._h2_ranges
        if rg is None:
            return False
        bits = rg.read()
        bad, _ = rg.saturated(bits)
        plan = None if bad else rg.quiet(bits)
        if (bad or plan) and reset:
            self._pipe_drain()
            if plan:
                rg.raise_(plan)
            else:
                try:
                    rg.lower([n for n in bad if bits[n] < rg.NONFINITE], bits)
                except E.FP16RangeExhausted:
                    pass                         # the batches that follow meet it again; predict() then takes bf16x6
            rg.reset_words()
            self._pipe_mark_stale()
        return bool(bad or plan)

    def h2_headroom(self):
        """{activation tensor: factor between its fp16 range and the largest magnitude seen} ("f16x3" mode)"""
        return self._h2_ranges.headroom() if self._h2_ranges is not None else {}

    def __getstate__(self):
        """The model is pickled into spawned worker processes (predict.py:130,252): ship parameters and
        configuration only - packed weights and workspaces are rebuilt in the worker on first use."""
        st = self.__dict__.copy()
        st["_engines"] = st["_engine_stamp"] = st["_tmpl_shared"] = st["_h2_ranges"] = None
        st.pop("_pipe", None)
        st.pop("_replay", None)
        st.pop("_in_replay", None)
        st.pop("_stamp_tensors", None)
        st.pop("_stamp_ptrs", None)
        st.pop("_bn_snapshot", None)       # training: copies of the BatchNorm statistics (training._BNSnapshot)
        st["_engines_by_precision"] = {}
        return st

    def _require_eval(self, what):
        if self.training:
            raise NotImplementedError(
                f"{what}() is an eval-mode entry point (running-statistics BatchNorm, as in the reference's predict.py); "
                "call .eval() first.  Training runs through forward(): net.train(); net(x) gives outputs that carry the "
                "HIP backward pass (sfh_amd.training)")

    def _needs_resize(self, x):
        """True when forward_unet has to resize its input or output (models/reconstructor.py:134-156)."""
        w, h = self.unet_size
        return (x.shape[3] != w or x.shape[2] != h) or tuple(self.target_size) != (w, h)

    def _template_is_shared(self, court_img, bs):
        """The reference replicates ONE template over the batch (utils/dataset.py:59).  Detect
        that once per template tensor so the warp reads a single (cache-resident) image."""
        key = (court_img.data_ptr(), tuple(court_img.shape), court_img._version)
        if self._tmpl_shared is None or self._tmpl_shared[0] != key:
            same = bool(court_img.shape[0] == 1 or E.rows_all_equal(court_img))
            self._tmpl_shared = (key, same)
        return self._tmpl_shared[1]

    # ------------------------------------------------------------------ reference API
    def warp(self, theta, court_img):
        """Warp the template by the predicted homographies (reference: :109-118)."""
        bs = theta.shape[0]
        template = court_img[0:bs]
        if template.shape[0] < bs:
            raise ValueError(f"batch {bs} exceeds the court template batch {court_img.shape[0]}")
        h, w = self._warp_hw
        shared = self._template_is_shared(court_img, bs)
        out, _ = E.homography_warp(theta, template.contiguous(), h, w, self.warp_with_nearest,
                                   shared_template=shared)
        return out

    def transform_poi(self, theta, court_poi, normalize=True):
        """Project the court PoI into the frame with inverse(theta) (reference: :120-130)."""
        return E.poi_project(theta, court_poi, normalize)

    def _run_unet(self, x, resume=None, **kw):
        """forward_unet core.  Input bilinear resize to unet_size and nearest resize of logits / uv to
        target_size as in models/reconstructor.py:134-156; with a resize active the fused STN input
        (built from the UNet-sized frame) is not valid and is dropped.  resume: repeat the engine's last run from
        that launch on (range guard) instead of starting a new one."""
        un, _ = self._get_engines()
        w, h = self.unet_size
        with torch.cuda.device(x.device):
            tw, th = self.target_size
            if resume is not None:
                r = dict(un.rerun(resume))
            else:
                xin = x.contiguous()
                if x.shape[3] != w or x.shape[2] != h:
                    xin = E.resize_nchw(xin, (h, w), "bilinear", align_corners=False)
                    kw["want_stn_in"] = False
                if (tw, th) != (w, h):
                    kw["want_stn_in"] = False
                r = dict(un.run(xin, **kw))      # a copy: the engine keeps its own dict for a later rerun()
            if (tw, th) != (w, h):
                r["logits"] = E.resize_nchw(r["logits"], (th, tw), "nearest")
                if "uv" in r:
                    r["uv"] = E.resize_nchw(r["uv"], (th, tw), "nearest")
            return r

    def forward_unet(self, x):
        """Reference: models/reconstructor.py:132-158.  Returns (logits, x_top, uv) in NCHW."""
        self._require_eval("forward_unet")

        def phases(x, off):
            def tail(r, theta):
                un, _ = self._get_engines()
                o = {"logits": r["logits"], "x_top": E.nhwc_to_nchw(r["x_top"], exp=un.x_top_exp(r))}
                if r.get("uv") is not None:
                    o["uv"] = r["uv"]
                return o
            return (lambda resume=None: self._run_unet(x, resume=resume, want_uv=self.unet_uv),
                    lambda r, resume=None: None, tail)
        self._pipe_wait_reads()
        o = self._chunked(lambda xi, off: self._guarded(phases, xi, off), x)
        return o["logits"], o["x_top"], o.get("uv")

    def _stn(self, x, r, resume=None, splitk=None):
        """theta = resnet_reg(cat(...)) for the configured input mode (reference: :174-185).  resume: repeat the
        ResNet engine's last run from that launch on (range guard)."""
        _, rn = self._get_engines()
        if resume is not None:
            with torch.cuda.device(x.device):
                return rn.rerun(resume)
        B, _, H, W = x.shape
        if self.resnet_input == Input.IMG_AND_MASK and "stn_in" in r:
            y = r["stn_in"]
        elif self.resnet_input == Input.IMG_AND_MASK:  # resized logits: assemble like the reference (:179,214)
            y = E.stn_input_assemble(r["logits"], x, None, rn.cs_in)
        elif self.resnet_input == Input.IMG:
            y = E.stn_input_assemble(None, x, None, rn.cs_in)
        elif self.resnet_input == Input.MASK:
            y = E.stn_input_assemble(r["logits"], None, None, rn.cs_in)
        elif self.resnet_input == Input.IMG_AND_MASK_AND_UV:
            y = E.stn_input_assemble(r["logits"], x, r["uv"], rn.cs_in)
        else:
            raise NotImplementedError
        with torch.cuda.device(x.device):
            return rn.run(y, B, H, W, splitk=splitk)

    # The conv kernels address every activation tensor through 32-bit buffer descriptors (< 4 GiB).  Larger batches
    # are processed in equal sub-batches and concatenated: the 64-channel full-resolution tensor of 16 frames at
    # 1280x720 is 3.8 GB with 4 bytes per element (f16x3, fp32: one launch) and 5.7 GB in the three-plane bf16 format
    # (8 + 8 frames).
    def _max_frames(self, x):
        prec = self._forced_precision or self.precision
        per_frame = x.shape[2] * x.shape[3] * 64 * (6 if prec == "bf16x6" else 4)
        return max(1, 0xFFFFFFF0 // per_frame)

    def _chunked(self, fn, x, *args, base=0):
        mf = self._max_frames(x)
        B = x.shape[0]
        if B <= mf:
            return fn(x, base, *args)
        n = -(-B // mf)                 # fewest sub-batches that fit the descriptor range ...
        size = -(-B // n)               # ... of equal size (8 + 8, not 12 + 4)
        outs = [fn(x[i:i + size], base + i, *args) for i in range(0, B, size)]
        return {k: torch.cat([o[k] for o in outs], 0) for k in outs[0]}

    def forward(self, x):
        """Reference: :160-194 - logits, [uv], theta, poi, bilinear (or nearest) warp_mask as float.
        Under ``train()`` the batch-statistics training kernels run and the outputs carry an
        autograd node whose backward produces the parameter gradients (training.py)."""
        if self.training:
            from . import training
            return training.train_forward(self, x)
        if x.shape[0] == 0:
            return self._empty_outputs(x, predict=False)
        self._pipe_wait_reads()
        return self._chunked(self._forward_one, x)

    def _forward_one(self, x, off):
        return self._guarded(self._forward_phases, x, off)

    def _forward_phases(self, x, off):
        def run_unet(resume=None):
            if not self.use_unet:
                return None
            return self._run_unet(x, resume=resume, want_stn_in=self.resnet_input == Input.IMG_AND_MASK,
                                  want_uv=self.unet_uv)

        def run_stn(r, resume=None):
            return self._stn(x, r, resume=resume) if self.use_resnet else None

        def tail(r, theta):
            ret = {}
            if r is not None:
                ret['logits'] = r["logits"]
                if "uv" in r:
                    ret['uv'] = r["uv"]
            if theta is not None:
                ret['theta'] = theta
                ret['poi'] = self.transform_poi(theta, self.court_poi[off:])
                if self.warper:
                    ret['warp_mask'] = self.warp(theta, self.court_img[off:])
            return ret
        return run_unet, run_stn, tail

    def predict(self, x, consistency=True, project_poi=False):
        """Reference: models/reconstructor.py:196-247.  With `net.graph_replay = True` batches of at most
        `net.graph_replay_max_batch` frames go through predict_replay() (one HIP-graph launch per batch, same bits)."""
        if self.__dict__.get("graph_replay") and 0 < x.shape[0] <= self.__dict__.get("graph_replay_max_batch", 8) \
                and not self.__dict__.get("_in_replay"):
            self.__dict__["_in_replay"] = True
            try:
                return self.predict_replay(x, consistency=consistency, project_poi=project_poi)
            finally:
                self.__dict__["_in_replay"] = False
        self._require_eval("predict")
        if x.shape[0] == 0:
            return self._empty_outputs(x, predict=True, consistency=consistency, project_poi=project_poi)
        self._pipe_wait_reads()
        return self._chunked(self._predict_one, x, consistency, project_poi)

    # ------------------------------------------------------------------ HIP-graph replay (small batches)
    REPLAY_CACHE = 4      # captured (shape, outputs) combinations kept per model

    def _replay_key(self, x, consistency, project_poi):
        rg = self._h2_ranges
        prec = self._forced_precision or self.precision
        exps = tuple(sorted(rg.exps.items())) if (rg is not None and prec == "f16x3") else None
        return (tuple(x.shape), x.device, bool(consistency), bool(project_poi), self._param_stamp(), exps,
                self.court_img.data_ptr(), self.court_poi.data_ptr())

    def predict_replay(self, x, consistency=True, project_poi=False):
        """predict() with the host out of the loop: the launches of one batch shape are captured ONCE into a HIP graph
        (torch.cuda.CUDAGraph around the same ctypes launches - ~90 kernels at one frame) and every further call is one copy of
        the frames into the graph's input buffer, one graph launch, the range read-back, and copies of the outputs into fresh
        tensors: ~0.1 ms of host time per batch instead of ~1 ms of Python enqueue.  Same kernels, same buffers, same bits as
        predict() (tests/test_gpu_round6.py).  The capture belongs to (batch shape, requested outputs, weights stamp, activation
        exponents, caller's stream); new weights, a moved exponent or another shape capture anew, and a batch whose range
        check fails after the replay is recomputed by predict() (which lowers / raises the exponent) - nothing is ever
        returned from a replay that saturated.  What it buys is bounded by the GPU, not the host: at one frame the launches
        already run back to back (94.6 % busy, profiles/r06_batch1_launch_table.txt), so the wall time drops by the idle 5 % and
        the enqueue latency, 2.19 -> ~2.0 ms; at 8 frames nothing.  For a caller whose own thread is busy (decoding, writing) it
        returns that thread.  Configurations outside the pipeline's (resize paths, other input modes, sub-batching) fall through to
        predict()."""
        self._require_eval("predict_replay")
        simple = (self.use_unet and self.use_resnet and x.is_cuda and x.shape[0] > 0 and x.shape[0] <= self._max_frames(x)
                  and not self._needs_resize(x) and self.resnet_input == Input.IMG_AND_MASK
                  and x.dtype == torch.float32 and x.is_contiguous() and E.PackedConv.timer is None)
        if not simple:
            return self.predict(x, consistency=consistency, project_poi=project_poi)
        cache = self.__dict__.setdefault("_replay", {})
        key = self._replay_key(x, consistency, project_poi)
        ent = cache.get(key)
        if ent is None:
            # first batch of this shape: eagerly (engines, workspaces, the template check and the exponents settle here) ...
            out = self.predict(x, consistency=consistency, project_poi=project_poi)
            key = self._replay_key(x, consistency, project_poi)          # ... under the exponents the eager pass left
            if key in cache:
                return out
            self._pipe_drain()
            dev = x.device
            cur = torch.cuda.current_stream(dev)
            with torch.cuda.device(dev):
                xs = torch.empty_like(x)
                xs.copy_(x)
                cs = torch.cuda.Stream(dev)
                cs.wait_stream(cur)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=cs, capture_error_mode="thread_local"):
                    run_unet, run_stn, tail = self._predict_phases(xs, 0, consistency, project_poi)
                    r = run_unet()
                    static = tail(r, run_stn(r))
                cur.wait_stream(cs)
            while len(cache) >= self.REPLAY_CACHE:
                cache.pop(next(iter(cache)))
            cache[key] = {"graph": g, "x": xs, "out": static}
            return out
        self._pipe_wait_reads()
        ent["x"].copy_(x)
        ent["graph"].replay()
        rg = self._h2_ranges
        if (self._forced_precision or self.precision) == "f16x3" and rg is not None and self.range_guard:
            bits = rg.read()                 # the one synchronisation of the call, as in predict()
            bad, _ = rg.saturated(bits)
            if bad or rg.quiet(bits):
                cache.pop(key, None)         # the exponents are about to move: this capture is void
                return self.predict(x, consistency=consistency, project_poi=project_poi)
        return {k: v.clone() for k, v in ent["out"].items()}     # fresh tensors, as the reference returns them

    def predict_async(self, x, consistency=True, project_poi=False):
        """predict() for callers that feed batch after batch (predict.py's loop): returns a handle at once; `handle.result()`
        gives predict()'s dict.  The UNet of this batch is enqueued on the caller's stream, everything behind it - ResNet-STN,
        warp, consistency CE, POI: 36 small launches that leave most CUs idle - on a SIDE stream behind an event, so that it
        runs UNDER THE UNET OF THE NEXT BATCH the caller submits (two STN-input buffers alternate; outputs are fresh
        tensors).  Same kernels, same bits as predict().  Keep at most two batches in flight: submit k + 1, then take
        result(k).  The fp16 range check of a batch happens in its result() (one read-back on a third stream, behind that
        batch only); if a tensor was saturated the pipeline is drained, the exponent lowered and the batches in flight are
        recomputed with predict()."""
        self._require_eval("predict_async")
        simple = (self.use_unet and self.use_resnet and x.shape[0] > 0 and x.shape[0] <= self._max_frames(x)
                  and not self._needs_resize(x) and self.resnet_input == Input.IMG_AND_MASK)
        if not simple:      # configurations the pipeline does not cover run synchronously
            return _Done(self.predict(x, consistency=consistency, project_poi=project_poi))
        p = self.__dict__.get("_pipe")
        dev = x.device
        if p is None or p["device"] != dev:
            with torch.cuda.device(dev):
                # experiment knob: HIP priority of the side stream (negative = higher than the caller's default stream)
                prio = int(os.environ.get("SFH_SIDE_PRIO", "0"))
                side, copy = _pipe_streams(dev, prio)
                p = self.__dict__["_pipe"] = {"device": dev, "side": side, "copy": copy,
                                              "slot": 0, "stem_read": [None, None], "inflight": []}
        cur = torch.cuda.current_stream(dev)
        slot = p["slot"]
        p["slot"] ^= 1
        if p["stem_read"][slot] is not None:          # the batch that used this STN-input buffer last has consumed it
            cur.wait_event(p["stem_read"][slot])
        off = 0
        run_unet, run_stn, tail = self._predict_phases(x, off, consistency, project_poi, stn_slot=slot, pipelined=True)
        r = run_unet()
        unet_done = torch.cuda.Event()
        unet_done.record(cur)
        side = p["side"]
        side.wait_event(unet_done)
        with torch.cuda.stream(side):
            r["logits"].record_stream(side)
            theta = run_stn(r)
            stem_read = torch.cuda.Event()
            stem_read.record(side)                    # (recorded behind the whole ResNet: coarser than needed, never early)
            ret = tail(r, theta)
            done = torch.cuda.Event()
            done.record(side)
        p["stem_read"][slot] = stem_read
        h = _Pending(self, x, consistency, project_poi, ret, done)
        p["inflight"].append(h)
        return h

    def _pipe_check(self, h):
        """range check of one pipelined batch (called by its result()): True = the outputs stand"""
        rg = self._h2_ranges
        p = self.__dict__["_pipe"]
        if h in p["inflight"]:
            p["inflight"].remove(h)
        if (self._forced_precision or self.precision) != "f16x3" or rg is None or not self.range_guard:
            return True
        with torch.cuda.stream(p["copy"]):
            p["copy"].wait_event(h.done)
            bits = rg.read()            # synchronises the copy stream only: the next batch's UNet keeps running
        bad, nonfinite = rg.saturated(bits)
        plan = None if bad else rg.quiet(bits)
        if not bad and not plan:
            return True
        # drain, fix the exponents, and let every batch in flight (this one included) be recomputed synchronously
        torch.cuda.synchronize(p["device"])
        rg.reset_words()
        if plan:
            rg.raise_(plan)
            self.range_raises += 1
        elif not nonfinite:
            try:
                rg.lower(bad, bits)
                self.range_rescales += 1
            except E.FP16RangeExhausted:
                pass                             # predict() of the recomputation takes the three-plane operands
        self._pipe_mark_stale()
        return False

    def _empty_outputs(self, x, predict, consistency=False, project_poi=False):
        """A batch of zero frames: the reference's torch ops return empty tensors of the usual trailing shapes."""
        if not x.is_cuda:
            raise RuntimeError("input frames are on the CPU: the HIP path has no CPU fallback")
        dev, f32 = x.device, torch.float32
        tw, th = self.target_size
        h, w = self._warp_hw
        ret = {}
        if self.use_unet:
            ret["logits"] = torch.empty((0, self.mask_classes, th, tw), dtype=f32, device=dev)
            if not predict and self.unet_uv:
                ret["uv"] = torch.empty((0, 2, th, tw), dtype=f32, device=dev)
        if self.use_resnet:
            ret["theta"] = torch.empty((0, 1, 3, 3), dtype=f32, device=dev)
            if self.warper:
                ret["warp_mask"] = torch.empty((0, h, w), dtype=torch.int32 if predict else f32, device=dev)
                if predict and consistency and self.use_unet:
                    ret["consist_score"] = torch.empty((0,), dtype=f32, device=dev)
            if project_poi or not predict:
                ret["poi"] = torch.empty((0,) + tuple(self.court_poi.shape[1:]), dtype=f32, device=dev)
        return ret

    def _predict_one(self, x, off, consistency, project_poi):
        return self._guarded(lambda xi, o: self._predict_phases(xi, o, consistency, project_poi), x, off)

    def _predict_phases(self, x, off, consistency, project_poi, stn_slot=0, pipelined=False):
        def run_unet(resume=None):
            if not self.use_unet:
                return None
            return self._run_unet(x, resume=resume, want_stn_in=self.resnet_input == Input.IMG_AND_MASK, stn_slot=stn_slot)

        def run_stn(r, resume=None):
            if not self.use_resnet:
                return None
            if self.resnet_input == Input.IMG_AND_MASK_AND_UV:
                raise NotImplementedError  # the reference's predict() has no uv branch either (:216)
            return self._stn(x, r, resume=resume, splitk=False if (pipelined and not self.pipeline_splitk) else None)

        def tail(r, theta):
            ret = {}
            if r is not None:
                ret['logits'] = r["logits"]
            if theta is not None:
                ret['theta'] = theta
                if self.warper:
                    bs = theta.shape[0]
                    h, w = self._warp_hw
                    tmpl = self.court_img[off:off + bs]
                    if tmpl.shape[0] < bs:
                        raise ValueError(f"batch {bs} exceeds the court template batch {self.court_img.shape[0]}")
                    shared = self._template_is_shared(self.court_img, bs)
                    lgt = ret.get('logits')
                    if (consistency and self.use_unet and self.warp_with_nearest and self.fuse_warp_ce
                            and self.mask_classes == 4 and tuple(lgt.shape[1:]) in ((4, h, w), (4, h // 2, w // 2))
                            and (h % 2 == 0 and w % 2 == 0 or tuple(lgt.shape[2:]) == (h, w))):
                        # the usual cases - the warp has the logits' size, or twice it (predict.py's default geometry) - warp and
                        # consistency score fused: every wave scores the pixels it warps while their class ids are in registers,
                        # the logits are streamed once (reference: :223-240)
                        wm, ret['consist_score'] = E.warp_consistency(theta, tmpl.contiguous(), lgt, float(self.mask_classes),
                                                                      shared_template=shared, warp_hw=(h, w))
                    else:
                        # warp * mask_classes -> int32, fused in the kernel (reference: :223,240)
                        _, wm = E.homography_warp(theta, tmpl.contiguous(), h, w, self.warp_with_nearest,
                                                  scale=float(self.mask_classes), want_f32=False, want_i32=True,
                                                  shared_template=shared)
                        if consistency and self.use_unet:
                            ret['consist_score'] = E.consistency_ce(lgt, wm)
                    ret['warp_mask'] = wm
                if project_poi:
                    ret['poi'] = self.transform_poi(theta, self.court_poi[off:])
            return ret
        return run_unet, run_stn, tail
