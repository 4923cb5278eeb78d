"""Host-side logic of the Reconstructor mirror that needs no GPU: sub-batching around the 32-bit buffer
descriptors, the fp16-range guard of the "f16x3" mode and its re-run in "bf16x6", weight-exponent choice."""
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


class _Flag:
    """stands in for the int32 device word the H2 kernels raise"""

    def __init__(self, hits):
        self.hits = list(hits)     # value returned by successive item() calls
        self.zeroed = 0

    def item(self):
        return self.hits.pop(0) if self.hits else 0

    def zero_(self):
        self.zeroed += 1


def _net(wh):
    net = Reconstructor(synth.load_court_template(batch_size=1), synth.load_court_poi(batch_size=1), target_size=wh,
                        unet_size=wh, warp_size=wh).eval()
    calls = []

    def fake(x, off, consistency, project_poi):
        calls.append((x.shape[0], off, net._forced_precision or net.precision))
        return {"theta": torch.full((x.shape[0], 1), float(off))}
    net._predict_one_unguarded = fake
    return net, calls


def test_sub_batches_follow_the_bytes_per_element_of_the_precision():
    x = torch.empty((16, 3, 720, 1280))
    net, calls = _net((1280, 720))
    for prec, want in (("f16x3", [(16, 0)]), ("fp32", [(16, 0)]), ("bf16x6", [(8, 0), (8, 8)])):
        net.precision = prec
        net.range_guard = False
        calls.clear()
        out = net.predict(x)
        assert [(b, o) for b, o, _ in calls] == want, prec
        assert out["theta"].shape[0] == 16
    # 640x360: 64 frames of 6 B/element need two launches (48 fit), of 4 B/element one (72 fit)
    x = torch.empty((64, 3, 360, 640))
    net, calls = _net((640, 360))
    net.range_guard = False
    net.precision = "bf16x6"
    net.predict(x)
    assert [(b, o) for b, o, _ in calls] == [(32, 0), (32, 32)]
    calls.clear()
    net.precision = "f16x3"
    net.predict(x)
    assert [(b, o) for b, o, _ in calls] == [(64, 0)]


def test_range_guard_reruns_in_bf16x6_and_rechunks():
    x = torch.empty((16, 3, 720, 1280))
    net, calls = _net((1280, 720))
    net.precision = "f16x3"
    net._h2_overflow = _Flag([1])
    with pytest.warns(UserWarning, match="fp16 range"):
        out = net.predict(x)
    # one f16x3 launch of 16 frames, flagged; then 8 + 8 frames with the three-plane operands, offsets kept
    assert calls == [(16, 0, "f16x3"), (8, 0, "bf16x6"), (8, 8, "bf16x6")]
    assert out["theta"][:, 0].tolist() == [0.0] * 8 + [8.0] * 8
    assert net.range_fallbacks == 1 and net._h2_overflow.zeroed == 1 and net._forced_precision is None
    # second hit: counted, no second warning
    net._h2_overflow = _Flag([0, 1])
    calls.clear()
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        net.predict(x)              # clean
        net.predict(x)              # flagged
    assert net.range_fallbacks == 2 and [c[2] for c in calls] == ["f16x3", "f16x3", "bf16x6", "bf16x6"]
    # guard switched off: the caller asks
    net.range_guard = False
    net._h2_overflow = _Flag([1])
    calls.clear()
    net.predict(x)
    assert [c[2] for c in calls] == ["f16x3"] and net.range_overflowed() is True and net.range_overflowed() is False
    # other precisions never read the word
    net.range_guard = True
    net.precision = "bf16x6"
    net._h2_overflow = _Flag([1])
    net.predict(x)
    assert net._h2_overflow.hits == [1]


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
