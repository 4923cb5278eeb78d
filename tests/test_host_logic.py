"""Host-side logic of the Reconstructor mirror that needs no GPU: sub-batching around the 32-bit buffer
descriptors, the fp16-range guard of the "f16x3" mode (per-tensor exponents, resume from the saturated layer, bf16x6
only for non-finite values), weight-exponent choice."""
import math
import os
import sys
import warnings

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from sfh_amd import synth  # noqa: E402
from sfh_amd.reconstructor import Reconstructor  # noqa: E402


class _Ranges:
    """stands in for engine.H2Ranges: successive read() results, what was lowered, how often the words were zeroed"""
    NONFINITE = 0x7F800000

    def __init__(self, reads):
        self.reads = [dict(r) for r in reads]
        self.lowered = []
        self.raised = []
        self.zeroed = 0

    def read(self):
        return self.reads.pop(0) if self.reads else {}

    def saturated(self, bits):
        bad = [n for n, b in bits.items() if b > 0x477FE000]
        return bad, any(bits[n] >= self.NONFINITE for n in bad)

    def lower(self, bad, bits):
        self.lowered.extend(bad)
        return set(bad)

    def quiet(self, bits):
        """names whose word is an ordinary float below 4.0 (the scripted reads use tiny integers for "fine")"""
        return {n: 24 for n, b in bits.items() if 0x30000000 <= b < 0x40800000}

    def raise_(self, plan):
        self.raised.extend(sorted(plan))
        return set(plan)

    def reset_words(self):
        self.zeroed += 1


class _Engine:
    def __init__(self, steps):
        self.steps = steps      # {key: index of the first launch that writes it}

    def first_step(self, keys):
        hit = [self.steps[k] for k in keys if k in self.steps]
        return min(hit) if hit else None


def _net(wh):
    net = Reconstructor(synth.load_court_template(batch_size=1), synth.load_court_poi(batch_size=1), target_size=wh,
                        unet_size=wh, warp_size=wh).eval()
    calls = []

    def fake_phases(x, off, consistency, project_poi):
        prec = net._forced_precision or net.precision

        def run_unet(resume=None):
            calls.append(("unet", x.shape[0], off, prec, resume))
            return {"logits": None}

        def run_stn(r, resume=None):
            calls.append(("stn", x.shape[0], off, prec, resume))
            return "theta"

        def tail(r, theta):
            calls.append(("tail", x.shape[0], off, prec, None))
            return {"theta": torch.full((x.shape[0], 1), float(off))}
        return run_unet, run_stn, tail
    net._predict_phases = fake_phases
    net._engines = (_Engine({"inc.out": 1, "down2.mid": 5}), _Engine({"rn.layer3.0.out": 7}))
    return net, calls


def _passes(calls):
    """(frames, offset, precision) of every full pass = every un-resumed unet phase"""
    return [(b, o, p) for what, b, o, p, resume in calls if what == "unet" and resume is None]


def test_sub_batches_follow_the_bytes_per_element_of_the_precision():
    x = torch.empty((16, 3, 720, 1280))
    net, calls = _net((1280, 720))
    for prec, want in (("f16x3", [(16, 0)]), ("fp32", [(16, 0)]), ("bf16x6", [(8, 0), (8, 8)])):
        net.precision = prec
        net.range_guard = False
        calls.clear()
        out = net.predict(x)
        assert [(b, o) for b, o, _ in _passes(calls)] == want, prec
        assert out["theta"].shape[0] == 16
    # 640x360: 64 frames of 6 B/element need two launches (48 fit), of 4 B/element one (72 fit)
    x = torch.empty((64, 3, 360, 640))
    net, calls = _net((640, 360))
    net.range_guard = False
    net.precision = "bf16x6"
    net.predict(x)
    assert [(b, o) for b, o, _ in _passes(calls)] == [(32, 0), (32, 32)]
    calls.clear()
    net.precision = "f16x3"
    net.predict(x)
    assert [(b, o) for b, o, _ in _passes(calls)] == [(64, 0)]


BIG, NAN = 0x47800000, 0x7FC00000     # bit patterns: 65536.f (beyond 65504), a NaN


def test_range_guard_lowers_the_exponent_and_resumes_from_the_layer():
    """A tensor beyond its fp16 range: its exponent goes down and the pass resumes at the launch that writes it -
    the UNet from that step (then the whole STN and the tail), or only the ResNet from its step; the model stays on
    the two-plane path (no bf16x6 re-run)."""
    x = torch.empty((16, 3, 720, 1280))
    net, calls = _net((1280, 720))
    net.precision = "f16x3"
    net._h2_ranges = rg = _Ranges([{"down2.mid": BIG, "inc.out": 5}, {"rn.layer3.0.out": BIG}, {}])
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        out = net.predict(x)
    assert [c[0] + ("" if c[4] is None else f"@{c[4]}") for c in calls] == [
        "unet", "stn", "tail",            # the pass
        "unet@5", "stn", "tail",          # down2.mid saturated: UNet from launch 5, everything behind it again
        "stn@7", "tail"]                  # then a ResNet tensor: only the ResNet from launch 7 and the tail
    assert rg.lowered == ["down2.mid", "rn.layer3.0.out"] and rg.zeroed == 2
    assert net.range_rescales == 2 and net.range_fallbacks == 0 and out["theta"].shape[0] == 16
    # a clean pass reads the words once and repeats nothing
    net._h2_ranges = _Ranges([{"inc.out": 5}])
    calls.clear()
    net.predict(x)
    assert [c[0] for c in calls] == ["unet", "stn", "tail"] and net.range_rescales == 2


QUIET = 0x3A800000                    # bit pattern of 2^-10: far below the 4.0 under which a tensor counts as quiet


def test_range_guard_raises_the_exponent_of_a_quiet_tensor_and_resumes_from_the_layer():
    """The other direction (round 5): a tensor whose largest stored element sits at the bottom of the fp16 range loses
    bits in its low plane without saturating anything.  Its exponent goes UP and the pass resumes at the launch that
    writes it, exactly like after a saturation; a saturated tensor in the same read-back wins (one decision per pass)."""
    x = torch.empty((16, 3, 720, 1280))
    net, calls = _net((1280, 720))
    net.precision = "f16x3"
    net._h2_ranges = rg = _Ranges([{"down2.mid": QUIET, "inc.out": 5}, {"rn.layer3.0.out": QUIET, "down2.mid": 0x45800000}, {}])
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        out = net.predict(x)
    assert [c[0] + ("" if c[4] is None else f"@{c[4]}") for c in calls] == [
        "unet", "stn", "tail", "unet@5", "stn", "tail", "stn@7", "tail"]
    assert rg.raised == ["down2.mid", "rn.layer3.0.out"] and rg.lowered == [] and rg.zeroed == 2
    assert net.range_raises == 2 and net.range_rescales == 0 and net.range_fallbacks == 0 and out["theta"].shape[0] == 16
    # saturated and quiet tensors in one read-back: the saturated one is handled first, the quiet one in the next round
    net._h2_ranges = rg = _Ranges([{"down2.mid": BIG, "inc.out": QUIET}, {"inc.out": QUIET}, {}])
    calls.clear()
    net.predict(x)
    assert rg.lowered == ["down2.mid"] and rg.raised == ["inc.out"]
    assert [c[0] + ("" if c[4] is None else f"@{c[4]}") for c in calls] == [
        "unet", "stn", "tail", "unet@5", "stn", "tail", "unet@1", "stn", "tail"]
    # guard off: range_overflowed() reports and fixes quiet tensors too
    net.range_guard = False
    net._h2_ranges = rg = _Ranges([{"inc.out": QUIET}, {}])
    calls.clear()
    net.predict(x)
    assert [c[0] for c in calls] == ["unet", "stn", "tail"]
    assert net.range_overflowed() is True and rg.raised == ["inc.out"] and rg.zeroed == 1 and net.range_overflowed() is False


def test_h2_ranges_raise_quiet_keys():
    """engine.H2Ranges.quiet / raise_: a key whose written tensors all peak below 4.0 in stored units goes to
    [2^12, 2^13); unwritten words say nothing; `lower`'s exponent is a ceiling until the weights change."""
    import numpy as np
    from sfh_amd import engine as E
    rg = E.H2Ranges(torch.device("cpu"), capacity=8)
    rg.register("inc.out")
    rg.register("inc.pool", key="inc.out", word_of="inc.out")
    rg.register("up4.up", key="inc.out")
    rg.register("down1.mid")
    rg.register("down1.out")
    rg.register("frame")
    w = rg.words.numpy().view("uint32")

    def put(name, u):
        w[rg.slot[name][1]] = np.float32(u).view("uint32")
    # |v| peaks at 2^-12 in down1.mid: stored 2^-10 at the default exponent 2 -> e = 24 puts it at 2^12
    put("down1.mid", 2.0 ** -10)
    # a key with two written tensors: the larger one decides, and it is not quiet (stored 40 >= 4)
    put("inc.out", 0.5)
    put("up4.up", 40.0)
    # the frame of an ordinary video: peak 1.0 = stored 4.0, exactly at the boundary: left alone
    put("frame", 4.0)
    bits = rg.read()
    assert rg.saturated(bits) == ([], False)
    plan = rg.quiet(bits)
    assert plan == {"down1.mid": 24}
    assert rg.raise_(plan) == {"down1.mid"} and rg.exp("down1.mid") == 24 and rg.exp("down1.out") == 2
    assert 2.0 ** 12 <= 2.0 ** -12 * 2.0 ** rg.exp("down1.mid") < 2.0 ** 13
    # the words of the repeated pass: nothing quiet any more (down1.out, never written, is not a reason to act)
    rg.reset_words()
    put("down1.mid", 2.0 ** 12)
    put("inc.out", 0.5)
    put("up4.up", 40.0)
    assert rg.quiet(rg.read()) == {}
    # both tensors of a key quiet: raised from the LARGER of the two, once
    rg.reset_words()
    put("inc.out", 2.0 ** -4)           # |v| = 2^-6
    put("up4.up", 2.0 ** -7)
    plan = rg.quiet(rg.read())
    assert plan == {"inc.out": 18} and 2.0 ** 12 <= 2.0 ** -6 * 2.0 ** 18 < 2.0 ** 13
    rg.raise_(plan)
    assert rg.exp("inc.pool") == rg.exp("up4.up") == 18
    # hysteresis: a key that a saturation brought down is not raised beyond that exponent in this weights generation
    rg.reset_words()
    put("inc.out", 1.0e5)               # |v| = 1e5 * 2^-18 = 0.38: saturated at e = 18
    bits = rg.read()
    bad, _ = rg.saturated(bits)
    rg.lower(bad, bits)
    e = rg.exp("inc.out")
    assert e == 14 and rg.ceiling["inc.out"] == 14
    rg.reset_words()
    put("inc.out", 2.0 ** -3)           # a very quiet batch afterwards
    assert rg.quiet(rg.read()) == {}    # would be 28, capped by the ceiling 14 = no change
    rg.new_generation()                 # new weights: the ceiling belongs to the old ones
    assert int(rg.words.abs().sum()) == 0 and rg.exp("inc.out") == 14
    put("inc.out", 2.0 ** -3)
    plan = rg.quiet(rg.read())
    assert plan == {"inc.out": 14 + 15}
    rg.raise_(plan)
    # tiny tensors stop at MAX_EXP; an all-zero tensor (word 0) is left alone
    rg.reset_words()
    put("down1.out", 2.0 ** -120)
    assert rg.quiet(rg.read()) == {"down1.out": rg.MAX_EXP}


def test_h2_ranges_quiet_judges_a_key_on_all_of_its_tensors_across_a_resume():
    """a key shared by an EARLY tensor (a skip tensor) and a LATE one (the decoder's up tensor).  After a
    decision the words are zeroed and the pass resumes behind the early tensor, so only the late one has a word: the key
    must not be raised on the late tensor alone - the early tensor's peak of this weights generation still counts."""
    import numpy as np
    from sfh_amd import engine as E
    rg = E.H2Ranges(torch.device("cpu"), capacity=8)
    rg.register("inc.out")
    rg.register("up4.up", key="inc.out")
    rg.register("down4.out")
    w = rg.words.numpy().view("uint32")

    def put(name, u):
        w[rg.slot[name][1]] = np.float32(u).view("uint32")
    # full pass: the skip tensor peaks at |v| = 25 (stored 100), the up tensor is quiet, down4.out is quiet too
    put("inc.out", 100.0)
    put("up4.up", 2.0 ** -6)
    put("down4.out", 2.0 ** -8)
    plan = rg.quiet(rg.read())
    assert plan == {"down4.out": 22}                     # inc.out's key is decided by its larger tensor: left alone
    rg.raise_(plan)
    # the words are zeroed and the pass resumes from down4 (behind inc.out): only the late tensors are written again
    rg.reset_words()
    put("up4.up", 2.0 ** -6)
    put("down4.out", 2.0 ** 12)
    assert rg.quiet(rg.read()) == {}                     # NOT {"inc.out": ...}: the un-inspected skip tensor would saturate
    # the same words without the earlier observation (a new weights generation): the late tensor alone decides
    rg.new_generation()
    put("up4.up", 2.0 ** -6)
    assert rg.quiet(rg.read()) == {"inc.out": 20}


def test_range_guard_reruns_a_non_finite_batch_in_bf16x6_and_rechunks():
    x = torch.empty((16, 3, 720, 1280))
    net, calls = _net((1280, 720))
    net.precision = "f16x3"
    net._h2_ranges = _Ranges([{"inc.mid": NAN}])
    with pytest.warns(UserWarning, match="non-finite"):
        out = net.predict(x)
    # one f16x3 pass of 16 frames, flagged; then 8 + 8 frames with the three-plane operands, offsets kept
    assert _passes(calls) == [(16, 0, "f16x3"), (8, 0, "bf16x6"), (8, 8, "bf16x6")]
    assert out["theta"][:, 0].tolist() == [0.0] * 8 + [8.0] * 8
    assert net.range_fallbacks == 1 and net._h2_ranges.zeroed == 1 and net._forced_precision is None
    assert net._h2_ranges.lowered == []
    # second hit: counted, no second warning
    net._h2_ranges = _Ranges([{}, {"inc.mid": NAN}])
    calls.clear()
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        net.predict(x)              # clean
        net.predict(x)              # flagged
    assert net.range_fallbacks == 2 and [p for _, _, p in _passes(calls)] == ["f16x3", "f16x3", "bf16x6", "bf16x6"]
    # guard switched off: the caller asks, and the exponents are lowered for the batches that follow
    net.range_guard = False
    net._h2_ranges = _Ranges([{"inc.out": BIG}, {}])
    calls.clear()
    net.predict(x)
    assert [p for _, _, p in _passes(calls)] == ["f16x3"]
    assert net.range_overflowed() is True and net._h2_ranges.lowered == ["inc.out"] and net.range_overflowed() is False
    # other precisions never read the words
    net.range_guard = True
    net.precision = "bf16x6"
    net._h2_ranges = _Ranges([{"inc.out": BIG}])
    net.predict(x)
    assert len(net._h2_ranges.reads) == 1


def test_h2_ranges_bookkeeping():
    """engine.H2Ranges on the host: shared exponent keys, shared words, the exponent a saturated tensor gets."""
    import numpy as np
    from sfh_amd import engine as E
    rg = E.H2Ranges(torch.device("cpu"), capacity=8)
    rg.register("inc.out")
    rg.register("inc.pool", key="inc.out", word_of="inc.out")
    rg.register("up4.up", key="inc.out")
    rg.register("down1.mid")
    assert rg.exp("inc.out") == rg.exp("nope") == 2 and rg.key("up4.up") == "inc.out"
    assert rg.word_ptr("inc.pool") == rg.word_ptr("inc.out") != rg.word_ptr("up4.up")
    with pytest.raises(ValueError):
        rg.register("up4.up", key="down1.mid")
    a = rg.args("inc.out", "down1.mid", None)
    assert a["exp_src"] == a["exp_dst"] == a["exp_res"] == 2 and a["range_word"] == rg.word_ptr("down1.mid")
    # the kernels saw |u| = 3e5 at exponent 2, i.e. |v| = 75000 in up4.up: new exponent puts it into [2^12, 2^13)
    w = rg.words.numpy().view("uint32")
    w[rg.slot["up4.up"][1]] = np.float32(3.0e5).view("uint32")
    w[rg.slot["down1.mid"][1]] = np.float32(100.0).view("uint32")
    bits = rg.read()
    bad, nonfinite = rg.saturated(bits)
    assert bad == ["up4.up"] and not nonfinite
    assert rg.lower(bad, bits) == {"inc.out"}
    e = rg.exp("inc.out")
    assert e == rg.exp("inc.pool") == rg.exp("up4.up") == -4 and 2.0 ** 12 <= 75000.0 * 2.0 ** e < 2.0 ** 13
    assert rg.exp("down1.mid") == 2 and abs(rg.peak["down1.mid"] - 25.0) < 1e-6
    assert abs(rg.headroom()["down1.mid"] - 65504.0 / 4 / 25.0) < 1e-3
    rg.reset_words()
    w[rg.slot["down1.mid"][1]] = 0x7F800000
    assert rg.saturated(rg.read()) == (["down1.mid"], True)


def test_h2_ranges_lower_decides_once_per_exponent_key():
    """A conv output and its pooled copy share word AND key, a skip tensor and its up tensor share a key: however many
    saturated names point at a key, its exponent is lowered ONCE, from the largest word, converted with the exponent
    the words were written under (a second conversion under the new exponent would give away 4 more bits for good)."""
    import numpy as np
    from sfh_amd import engine as E
    rg = E.H2Ranges(torch.device("cpu"), capacity=8)
    rg.register("inc.out")
    rg.register("inc.pool", key="inc.out", word_of="inc.out")
    rg.register("up4.up", key="inc.out")
    w = rg.words.numpy().view("uint32")
    w[rg.slot["inc.out"][1]] = np.float32(1.0e5).view("uint32")      # u = v * 2^2 = 1e5 in inc.out (and inc.pool)
    w[rg.slot["up4.up"][1]] = np.float32(7.0e4).view("uint32")       # also saturated, smaller
    bits = rg.read()
    bad, _ = rg.saturated(bits)
    assert sorted(bad) == ["inc.out", "inc.pool", "up4.up"]
    assert rg.lower(bad, bits) == {"inc.out"}
    # |v| = 25000 -> [2^12, 2^13) at e = -2 (twice-lowered it would end at -6)
    assert rg.exp("inc.out") == rg.exp("inc.pool") == rg.exp("up4.up") == -2
    assert 2.0 ** 12 <= 25000.0 * 2.0 ** -2 < 2.0 ** 13
    # at the lowest exponent a saturated tensor cannot be helped: the caller is told at once
    rg.exps["inc.out"] = rg.MIN_EXP
    with pytest.raises(E.FP16RangeExhausted):
        rg.lower(["inc.out"], {"inc.out": int(np.float32(1.0e5).view("uint32"))})


def test_weight_exponent_puts_the_largest_weight_below_2_to_14():
    """PackedConv._pack_split: wexp = 14 - e with max|w| = m * 2^e, 0.5 <= m < 1."""
    for wmax in (1.0, 0.999, 0.5, 0.03, 3.7e-4, 123.0, 2.0 ** -20):
        e = math.frexp(wmax)[1]
        wexp = 14 - e
        assert 2.0 ** 13 <= wmax * 2.0 ** wexp < 2.0 ** 14


def test_precision_names():
    from sfh_amd import engine as E
    assert E.PRECISIONS == {"f16x3": "h2", "bf16x6": "s3", "fp32": None}
    assert E.split_shape("h2", 2, 5, 7, 64) == (2, 5, 2, 2, 4, 7, 8)
    assert E.split_shape("s3", 2, 5, 7, 64) == (2, 5, 2, 3, 4, 7, 8)
    with pytest.raises(ValueError):
        E.split_shape("h2", 1, 4, 4, 48)


def test_engine_stamp_notices_in_place_updates_and_replaced_parameters():
    """The packed-weight engines are rebuilt when the stamp moves: in-place writes move a tensor's torch _version; a
    Parameter object that is replaced (anywhere under the model) registers with torch's module hooks."""
    net = Reconstructor(synth.load_court_template(batch_size=1), synth.load_court_poi(batch_size=1)).eval()
    s1 = net._param_stamp()
    assert net._param_stamp() == s1                                   # stable while nothing changes
    with torch.no_grad():
        net.inc.double_conv[0].weight.mul_(1.0)
    s2 = net._param_stamp()
    assert s2 != s1
    net.inc.double_conv[0].weight = torch.nn.Parameter(torch.zeros_like(net.inc.double_conv[0].weight))
    s3 = net._param_stamp()
    assert s3 != s2
    assert any(t is net.inc.double_conv[0].weight for t in net.__dict__["_stamp_tensors"])
    net.load_state_dict(net.state_dict())
    assert net._param_stamp() != s3


def test_engine_stamp_ignores_unrelated_modules_and_notices_sub_module_casts():
    """Registrations elsewhere in the process (another model, a loss with a weight buffer) re-walk the tree but do
    not drop the engines; .to() / .double() on a SUB-module replaces its buffers and gives its parameters new
    storages without any registration hook - the stamp must move and the cached list must hold the live tensors."""
    net = Reconstructor(synth.load_court_template(batch_size=1), synth.load_court_poi(batch_size=1)).eval()
    s1 = net._param_stamp()
    torch.nn.CrossEntropyLoss(weight=torch.ones(4))           # registers a buffer somewhere else
    torch.nn.Linear(3, 3)
    assert net._param_stamp() == s1
    net.resnet_reg.double().float()                            # sub-module _apply: new buffer objects, new storages
    s2 = net._param_stamp()
    assert s2 != s1
    live = {id(t) for t in list(net.parameters()) + list(net.buffers())}
    assert {id(t) for t in net.__dict__["_stamp_tensors"]} == live
    assert net._param_stamp() == s2


def test_small_map_kernel_rule():
    """engine.choose_small_map: the finer 12x20 x 32-cout tiling only for launches whose standard grid is at most one workgroup
    per CU and that gain at least 1.2x the workgroups from it."""
    from sfh_amd import engine as E

    def rule(batch, ho, wo, cout, cin_unused=None):
        zr = 1 + ((ho + 1) & 1)
        tile = E.choose_tile_s3(batch, ho, wo, 1, zr, cout // 64)
        return E.choose_small_map(batch, ho, wo, zr, cout, tile)
    # 640x360, batch 16: ResNet layer4 (192 standard workgroups) and layer3 (240) take it, layer2 / layer1 and the UNet do not
    assert rule(16, 12, 20, 512) and rule(16, 23, 40, 256)
    assert not rule(16, 45, 80, 128) and not rule(16, 90, 160, 64)
    assert not rule(16, 22, 40, 1024) and not rule(16, 45, 80, 512) and not rule(16, 360, 640, 64)
    # 1280x720, batch 16: layer4 is 23x40 with 512 channels: 480 standard workgroups - stays
    assert not rule(16, 23, 40, 512)
    # one frame: most of the net is under-filled
    assert rule(1, 45, 80, 512) and rule(1, 22, 40, 1024) and rule(1, 90, 160, 256)
    # never when the finer tiling does not add workgroups: four 12x20 frames x 64 couts = 8 standard workgroups (16x16 tiles) and 8 fine ones
    assert E.choose_small_map(4, 12, 20, 2, 64, E._lib.TILE_16x16) is False
    # the threshold is a parameter (experiments): with 224, layer3's 240 standard workgroups stay on the standard kernel
    assert E.choose_small_map(16, 23, 40, 1, 256, E._lib.TILE_32x8, max_wgs=224) is False


def _trained_like_template():
    from sfh_amd import synth
    from sfh_amd.reconstructor import Reconstructor
    court = synth.load_court_template("ncaa_nc4_640x360", 4, 1)
    poi = synth.load_court_poi("pitch", 1)
    return Reconstructor(court, poi, target_size=(640, 360), unet_size=(640, 360), warp_size=(640, 360), warp_with_nearest=True)


def test_trained_like_family_has_the_stated_statistics():
    """(CPU part of the GPU file: cheap) the family is what its docstring says"""
    from sfh_amd import synth
    net = _trained_like_template()
    sd, info = synth.trained_like_state_dict(net.state_dict(), 0, return_info=True)
    assert info["conv_bn_pairs"] == 54 and info["scaled_layer_exp"] == 12
    rv = sd["down2.maxpool_conv.1.double_conv.4.running_var"]
    assert float(rv.min()) < 1e-2 and float(rv.max()) > 1e2
    g = torch.cat([sd[k].flatten() for k in sd if k.endswith("double_conv.1.weight") or k.endswith("double_conv.4.weight")])
    assert 0.02 < float((g == 0).float().mean()) < 0.09 and float((g < 0).float().mean()) > 0.4
    w = sd["down3.maxpool_conv.1.double_conv.0.weight"]
    per_ch = w.flatten(1)
    z = (per_ch.abs() / per_ch.std(1, keepdim=True))
    assert float((z > 10).float().mean()) > 0.003          # the outliers survive the per-channel rescale
    net.load_state_dict(sd)                                 # and the layout is the checkpoint's (strict)


