"""CPU study for the two-plane fp16 operand format ("f16x3"): how much error does the operand representation
add to a whole forward pass, next to what fp32 accumulation already costs?

Every conv / transposed conv of the CPU restatement (oracle/torch_ref.py) is replaced by

    y = sum over the kept partial products of (planes of x) * (planes of w),  evaluated in fp64,

so the ONLY error of a mode is its operand representation and its dropped products (the accumulation is ideal);
the result is rounded to fp32 per layer like the GPU epilogue does.  Modes:

    fp64      : reference
    fp32      : stock fp32 CPU convs (MKL-DNN accumulation order) - the yardstick
    bf16x6    : three bf16 planes per operand, six products (the round-1 format)
    f16x3     : two fp16 planes per operand (x * 2^6, w * 2^s with max|w| 2^s in [2^13, 2^14)), three products
    f16x3u    : same without the power-of-two pre-scaling (shows why it is there)

Run:  python tests/probes/f16x3_error_probe.py [W H]      (test infrastructure; imports the oracle)
"""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import torch_ref  # noqa: E402
from sfh_amd import synth  # noqa: E402

_real_conv2d = F.conv2d
_real_convT = F.conv_transpose2d
MODE = "fp32"
ACT_EXP = int(os.environ.get("ACT_EXP", "6"))


def _split_bf16(v):
    p0 = v.to(torch.bfloat16).float()
    r = v - p0
    p1 = r.to(torch.bfloat16).float()
    p2 = (r - p1).to(torch.bfloat16).float()
    return [p0.double(), p1.double(), p2.double()]


def _split_f16(v):
    p0 = v.to(torch.float16).float()
    p1 = (v - p0).to(torch.float16).float()
    return [p0.double(), p1.double()]


def _emul(op, x, w, b, kw):
    if MODE == "fp64":
        return op(x.double(), w.double(), None if b is None else b.double(), **kw).float()
    if MODE == "fp32":
        return op(x, w, b, **kw)
    x = x.float()
    w = w.float()
    if MODE == "bf16x6":
        xs, ws = _split_bf16(x), _split_bf16(w)
        pairs = [(0, 2), (1, 1), (2, 0), (0, 1), (1, 0), (0, 0)]
        sc = 1.0
    else:
        if MODE == "f16x3":
            ws_exp = 13 - int(torch.floor(torch.log2(w.abs().max())).item())
            a = ACT_EXP
        else:
            ws_exp, a = 0, 0
        xs, ws = _split_f16(x * 2.0 ** a), _split_f16(w * 2.0 ** ws_exp)
        pairs = [(0, 1), (1, 0), (0, 0)]
        sc = 2.0 ** -(a + ws_exp)
    y = None
    for (pw, px) in pairs:
        t = op(xs[px], ws[pw], None, **kw)
        y = t if y is None else y + t
    y = y * sc
    if b is not None:
        y = y + b.double().view(1, -1, 1, 1)
    return y.float()


def conv2d(x, w, b=None, **kw):
    return _emul(_real_conv2d, x, w, b, kw)


def conv_transpose2d(x, w, b=None, **kw):
    return _emul(_real_convT, x, w, b, kw)


def run(W, H, modes, verbose=True):
    """-> {mode: (max |dlogits|, mean |dlogits|, max |dtheta|, argmax flips)} against the fp64 run"""
    global MODE
    B = 1
    torch.manual_seed(0)
    from sfh_amd.reconstructor import Reconstructor
    court = synth.load_court_template("ncaa_nc4_640x360", 4, B)[:, :, :H, :W].contiguous()
    poi = synth.load_court_poi("pitch", B)
    net = Reconstructor(court, poi, target_size=(W, H), unet_size=(W, H), warp_size=(W, H), warp_with_nearest=True)
    sd = synth.synth_state_dict(net.state_dict(), 19)
    x = synth.smooth_frames(B, H, W, seed=19)
    F.conv2d = conv2d
    F.conv_transpose2d = conv_transpose2d
    res = {}
    try:
        for m in ["fp64"] + [m for m in modes if m != "fp64"]:
            MODE = m
            with torch.no_grad():
                out = torch_ref.predict(x, sd, court, poi, warp_size=(W, H), unet_size=(W, H), target_size=(W, H),
                                        project_poi=True)
            res[m] = {k: out[k].double() for k in ("logits", "theta")}
            if verbose:
                print(m, "done", flush=True)
    finally:
        F.conv2d = _real_conv2d
        F.conv_transpose2d = _real_convT
    ref = res["fp64"]
    lg = ref["logits"]
    top2 = lg.topk(2, dim=1).values
    margin = (top2[:, 0] - top2[:, 1])
    out = {}
    if verbose:
        print(f"size {W}x{H}, B={B}; errors against the fp64 run")
    for m in [k for k in res if k != "fp64"]:
        dl = (res[m]["logits"] - lg).abs()
        dt = (res[m]["theta"] - ref["theta"]).abs().max().item()
        flips = (res[m]["logits"].argmax(1) != lg.argmax(1)).sum().item()
        out[m] = (dl.max().item(), dl.mean().item(), dt, flips)
        if verbose:
            print(f"  {m:8s} logits max {dl.max().item():.3e} mean {dl.mean().item():.3e}  theta max {dt:.3e}  argmax flips {flips} of {margin.numel()}")
    return out


if __name__ == "__main__":
    W_, H_ = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (320, 180)
    run(W_, H_, os.environ.get("MODES", "fp64,fp32,bf16x6,f16x3,f16x3u").split(","))
