"""CPU study (round 5, stage (a) of the Winograd experiment - no GPU minutes): would Winograd F(2x2, 3x3) on the two-plane fp16
operands keep the parity bounds on the nine long-K 3x3 stride-1 launches of the UNet (d2.3, d3.*, d4.*, u1.*, u2.*: 4.7 of
the 11.8 ms of DoubleConv time, MFMA- / power-bound)?  16 products per 2x2 outputs instead of 36 is the only lever that removes
MFMA work from them.

What is emulated for a "Winograd" layer (the design that would be built):
  * the PRODUCER's epilogue transforms its fp32 outputs: V = B^T d B per 4x4 input tile (stride 2), in fp32 (coefficients +-1),
    and re-splits V * 2^e into two fp16 planes (22 significand bits) - 16 positions per 2x2 tile;
  * the weights U = G g G^T are transformed in fp64 once, scaled so that max |U| * 2^s is in [2^13, 2^14), split in two planes;
  * 16 batched GEMMs over the channels with the three kept plane products, accumulated IN FP32 IN THE ORDER THE MFMA CHAIN USES
    (one rounding per 32-channel step and product - the accumulation order, not the operand format, dominates the error of the
    direct kernel, DESIGN.md section 2, so the probe must carry it for both variants);
  * Y = A^T M A in fp32, then scale / bias like the direct epilogue.
The direct f16x3 conv of the same layers is emulated with the same fp32 chain (32 channels x tap x product per step).  Every
other conv of the net is the f16x3 emulation of tests/probes/f16x3_error_probe.py (ideal accumulation).  Both nets are compared
with an fp64 run: max / mean |d logits|, |d theta|, arg-max flips and their fp64 top-2 margins, 8x8 block sums.

Kill criteria: mean logit error > 2x the direct f16x3's, or the C2 flip rule (a differing pixel needs a
golden top-2 margin < 2e-4) / the 8e-3 block-sum bound fails at 320x180.

Run:  python tests/probes/winograd_f16x3_probe.py [W H]   (test infrastructure; imports the oracle)
"""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import f16x3_error_probe as P  # noqa: E402
from oracle import torch_ref  # noqa: E402
from sfh_amd import synth  # noqa: E402

ACT_EXP = P.ACT_EXP
LONGK_MIN_CIN = 256          # the launches in question read >= 256 channels
VARIANT = "direct"           # "direct" | "wino": how the long-K 3x3 layers are evaluated
STATS = {}

BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float64)
G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float64)
AT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float64)
PAIRS = [(0, 1), (1, 0), (0, 0)]      # (weight plane, activation plane): w0*x1 + w1*x0 + w0*x0, in the kernel's order


def _split_f16_from64(v):
    p0 = v.to(torch.float16)
    p1 = (v - p0.double()).to(torch.float16)
    return [p0.float(), p1.float()]


def _wexp(w):
    return 13 - int(torch.floor(torch.log2(w.abs().max())).item())


def _chain_direct(x, w):
    """3x3 pad-1 conv of fp32 x (B,C,H,W) with fp32 w (K,C,3,3): two-plane operands, three products, fp32 accumulator updated
    once per (32-channel block, tap, product) like the MFMA chain of conv_s3_kernel (each 32-term partial is exact-ish: fp64)."""
    s = _wexp(w)
    xs = _split_f16_from64(x.double() * 2.0 ** ACT_EXP)
    ws = _split_f16_from64(w.double() * 2.0 ** s)
    B, C, H, W = x.shape
    acc = torch.zeros((B, w.shape[0], H, W), dtype=torch.float32)
    xp = [F.pad(t.double(), (1, 1, 1, 1)) for t in xs]
    for c0 in range(0, C, 32):
        for ky in range(3):
            for kx in range(3):
                for (pw, px) in PAIRS:
                    part = P._real_conv2d(xp[px][:, c0:c0 + 32, ky:ky + H, kx:kx + W], ws[pw][:, c0:c0 + 32, ky:ky + 1, kx:kx + 1].double())
                    acc = (acc.double() + part).float()
    return acc.double() * 2.0 ** -(ACT_EXP + s)


def _chain_wino(x, w):
    """the same conv as Winograd F(2x2,3x3) in the transformed domain (see the module docstring)"""
    B, C, H, W = x.shape
    K = w.shape[0]
    He, We = H + (H & 1), W + (W & 1)
    xp = F.pad(x, (1, 1 + We - W, 1, 1 + He - H))
    t = xp.unfold(2, 4, 2).unfold(3, 4, 2)                      # (B,C,th,tw,4,4) fp32
    bt = BT.float()
    V = torch.einsum("ij,bcxyjk,lk->bcxyil", bt, t, bt)         # fp32: sums of four fp32 values
    U = torch.einsum("ij,kcjl,ml->kcim", G, w.double(), G)      # fp64 (K,C,4,4)
    s = _wexp(U)
    Vs = _split_f16_from64(V.double() * 2.0 ** ACT_EXP)
    Us = _split_f16_from64(U * 2.0 ** s)
    th, tw = V.shape[2], V.shape[3]
    STATS.setdefault("v_over_x", []).append(float(V.abs().max() / x.abs().max()))
    acc = torch.zeros((16, B * th * tw, K), dtype=torch.float32)
    Vm = [v.permute(4, 5, 0, 2, 3, 1).reshape(16, B * th * tw, C).double() for v in Vs]   # (pos, tiles, C)
    Um = [u.permute(2, 3, 1, 0).reshape(16, C, K).double() for u in Us]                    # (pos, C, K)
    for c0 in range(0, C, 32):
        for (pw, px) in PAIRS:
            part = torch.bmm(Vm[px][:, :, c0:c0 + 32], Um[pw][:, c0:c0 + 32, :])
            acc = (acc.double() + part).float()
    M = (acc * 2.0 ** -(ACT_EXP + s)).reshape(4, 4, B, th, tw, K).permute(2, 5, 3, 4, 0, 1)   # (B,K,th,tw,4,4) fp32
    STATS.setdefault("m_max", []).append(float(M.abs().max()))
    at = AT.float()
    Y = torch.einsum("ij,bkxyjl,ml->bkxyim", at, M, at)         # fp32 (B,K,th,tw,2,2)
    Y = Y.permute(0, 1, 2, 4, 3, 5).reshape(B, K, He, We)[:, :, :H, :W]
    STATS.setdefault("y_max", []).append(float(Y.abs().max()))
    return Y.double()


def conv2d(x, w, b=None, **kw):
    long_k = (w.shape[2:] == (3, 3) and kw.get("stride", 1) == 1 and kw.get("padding", 0) == 1 and w.shape[1] >= LONGK_MIN_CIN
              and P.MODE == "f16x3")
    if not long_k:
        return P.conv2d(x, w, b, **kw)
    x, w = x.float(), w.float()
    fn = _chain_wino if VARIANT == "wino" else _chain_direct
    if w.shape[1] > w.shape[0]:
        # first conv of an Up block: conv(cat([skip, up])) - the product evaluates the u-half as a composed 2x2 conv over the
        # low-resolution tensor (never Winograd); only the skip half (the first cin/2 channels) is a 3x3 conv over a stored tensor
        c = w.shape[1] // 2
        y = fn(x[:, :c], w[:, :c]) + _chain_direct(x[:, c:], w[:, c:])
    else:
        y = fn(x, w)
    if b is not None:
        y = y + b.double().view(1, -1, 1, 1)
    return y.float()


def run(W, H, B=1, seed=19):
    global VARIANT
    from sfh_amd.reconstructor import Reconstructor
    court = synth.load_court_template("ncaa_nc4_640x360", 4, B)[:, :, :H, :W].contiguous()
    poi = synth.load_court_poi("pitch", B)
    net = Reconstructor(court, poi, target_size=(W, H), unet_size=(W, H), warp_size=(W, H), warp_with_nearest=True)
    sd = synth.synth_state_dict(net.state_dict(), seed)
    x = synth.smooth_frames(B, H, W, seed=seed) if os.environ.get("FRAMES", "smooth") == "smooth" else \
        synth.frames_to_float(synth.synth_frames_u8(B, H, W, seed=seed))
    res = {}
    F.conv2d = conv2d
    F.conv_transpose2d = P.conv_transpose2d
    try:
        for tag, mode, variant in (("fp64", "fp64", "direct"), ("fp32 (stock CPU convs)", "fp32", "direct"),
                                   ("f16x3 direct", "f16x3", "direct"), ("f16x3 winograd", "f16x3", "wino")):
            P.MODE, VARIANT = mode, variant
            with torch.no_grad():
                out = torch_ref.predict(x, sd, court, poi, warp_size=(W, H), unet_size=(W, H), target_size=(W, H), project_poi=True)
            res[tag] = {k: out[k].double() for k in ("logits", "theta")}
            print(tag, "done", flush=True)
    finally:
        F.conv2d = P._real_conv2d
        F.conv_transpose2d = P._real_convT
    ref = res.pop("fp64")
    lg = ref["logits"]
    top2 = lg.topk(2, dim=1).values
    margin = top2[:, 0] - top2[:, 1]
    Hc, Wc = H - H % 8, W - W % 8
    bs = lambda t: t[:, :, :Hc, :Wc].reshape(B, 4, Hc // 8, 8, Wc // 8, 8).sum(dim=(3, 5))
    print(f"size {W}x{H}, B={B}, seed {seed}; errors against the fp64 run; long-K layers: 3x3 stride 1 with >= {LONGK_MIN_CIN} input channels")
    out = {}
    for tag, r in res.items():
        dl = (r["logits"] - lg).abs()
        dt = (r["theta"] - ref["theta"]).abs().max().item()
        diff = r["logits"].argmax(1) != lg.argmax(1)
        mm = margin[diff]
        dbs = (bs(r["logits"]) - bs(lg)).abs().max().item()
        out[tag] = dict(max=dl.max().item(), mean=dl.mean().item(), theta=dt, flips=int(diff.sum()),
                        flip_margin=float(mm.max()) if mm.numel() else 0.0, blocksum=dbs)
        print(f"  {tag:24s} logits max {dl.max().item():.3e} mean {dl.mean().item():.3e}  theta max {dt:.3e}  "
              f"arg-max flips {int(diff.sum())} of {margin.numel()} (largest fp64 margin among them {out[tag]['flip_margin']:.2e})  "
              f"8x8 block sums max {dbs:.3e}")
    d, w_ = out["f16x3 direct"], out["f16x3 winograd"]
    print(f"  winograd / direct: mean logit error x{w_['mean'] / d['mean']:.2f}, max x{w_['max'] / d['max']:.2f}, theta x{w_['theta'] / max(d['theta'], 1e-30):.2f}")
    print(f"  transformed-domain magnitudes: max|V|/max|x| per layer {[round(v, 2) for v in STATS.get('v_over_x', [])]}")
    print(f"  max|M| / max|Y| per layer {[round(m / max(y, 1e-30), 2) for m, y in zip(STATS.get('m_max', []), STATS.get('y_max', []))]}")
    kill = w_["mean"] > 2.0 * d["mean"] or w_["flip_margin"] >= 2e-4 or w_["blocksum"] >= 8e-3
    print("  KILL" if kill else "  survives stage (a)")
    return out


if __name__ == "__main__":
    W_, H_ = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (320, 180)
    torch.set_num_threads(os.cpu_count() or 1)
    run(W_, H_, seed=int(os.environ.get("SEED", "19")))
