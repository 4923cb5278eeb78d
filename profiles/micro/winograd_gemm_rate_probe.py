"""Stage (b) of the Winograd F(2x2, 3x3) experiment, the measurement that decides it: what would the
transformed-domain GEMMs cost on this chip with this kernel family?

F(2x2,3x3) replaces the 3x3 conv of a layer (18 * P * cin * cout FLOP for P pixels) by 16 independent GEMMs over the channels,
one per position of the 4x4 transformed tile, each over P / 4 tiles: 8 * P * cin * cout FLOP, 0.444 of the direct work.  A GEMM
over channels only IS a 1x1 convolution: the 16 GEMMs together are a 1x1 conv cin -> cout over 16 * P / 4 = 4 P "pixels".  This
probe times exactly that with the product's own split-fp16 kernel (conv_s3_kernel<KS = 1>, every workgroup shape the launcher
offers) next to the direct 3x3 launch of the same layer.  It is an UPPER bound for a Winograd consumer: one weight set instead of
sixteen, no input transform (V = B^T d B, re-split), no output transform (A^T M A), the 4x larger V tensor already in HBM.

A GEMM over K = cin has 1/9 of the direct conv's reuse of a staged activation tile: per MFMA the workgroup must bring 9x the
activation bytes into LDS (and 16/9 the weight bytes into registers), so the 1x1 instance is bound by its LDS-DMA / L2 streams,
not by the matrix cores - the measured rate of that instance is what the experiment stands or falls with.
usage (GPU box): python profiles/micro/winograd_gemm_rate_probe.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from sfh_amd import engine as E  # noqa: E402


def bench(fn, reps=8):
    fn()
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def layer(name, cin, cout, h, w, B=16):
    torch.manual_seed(0)
    bn = torch.nn.BatchNorm2d(cout).cuda().eval()
    bias = torch.randn(cout, device="cuda") * 0.1
    # (a) the direct 3x3 launch, as the engine would issue it (launcher's choice of workgroup shape)
    w3 = torch.randn(cout, cin, 3, 3, device="cuda") * (2.0 / (9 * cin)) ** 0.5
    c3 = E.PackedConv(w3, bias, bn, 3, cin, fmt="h2", tag="probe")
    x = E.f32_to_h2(torch.relu(torch.randn(B, h, w, cin, device="cuda")))
    y = E.split_empty("h2", B, h, w, cout, "cuda")
    t3 = bench(lambda: c3.run(x, B, h, w, y))
    f3 = 2.0 * B * h * w * cout * 9 * cin
    # (b) the sixteen transformed-domain GEMMs as ONE 1x1 conv over 4x the pixels (16 positions x P/4 tiles), H2 in, H2 out
    he, we = h + (h & 1), w + (w & 1)
    w1 = torch.randn(cout, cin, 1, 1, device="cuda") * (2.0 / cin) ** 0.5
    c1 = E.PackedConv(w1, bias, bn, 1, cin, fmt="h2", tag="probe")
    xv = E.f32_to_h2(torch.relu(torch.randn(B, 2 * he, 2 * we, cin, device="cuda")))      # stands for V: 4 values per pixel
    yv = E.split_empty("h2", B, 2 * he, 2 * we, cout, "cuda")
    best = None
    for wg in (0, 64, 128):
        for tile in (None, 0, 1, 2):
            try:
                t = bench(lambda: c1.run(xv, B, 2 * he, 2 * we, yv, wg_couts=wg, tile=tile), reps=4)
            except (RuntimeError, ValueError):
                continue
            if best is None or t < best[0]:
                best = (t, wg, tile)
    t1 = best[0]
    f1 = 2.0 * B * (2 * he) * (2 * we) * cout * cin
    elems_in, elems_out = B * h * w * cin, B * h * w * cout
    # HBM bytes the Winograd form adds: V is written by the producer and read by the consumer at 16 B per element instead of 4
    extra_gb = 2 * 12 * elems_in / 1e9
    print(f"{name:8s} {cin:5d}->{cout:<5d} {h:3d}x{w:<3d}  direct 3x3: {t3:6.3f} ms {f3 / t3 / 1e9:6.1f} TFLOP/s | 16 GEMMs as a 1x1 conv over 4x the "
          f"pixels: {t1:6.3f} ms {f1 / t1 / 1e9:6.1f} TFLOP/s (wg_couts {best[1]}, tile {best[2]}) = {t1 / t3:4.2f} of the direct launch "
          f"(MFMA work 0.444) | + {extra_gb:4.2f} GB of V traffic = {extra_gb / 4.5:5.3f} ms at 4.5 TB/s "
          f"-> Winograd >= {(t1 + extra_gb / 4.5) / t3:4.2f} x direct", flush=True)
    del x, y, xv, yv
    return t3, t1, extra_gb


if __name__ == "__main__":
    print("F(2x2,3x3) on the nine long-K launches: direct 3x3 against an UPPER bound of the transformed-domain GEMMs (no transforms, one weight set)")
    tot3 = tot1 = totx = 0.0
    for args in (("d2.3", 256, 256, 90, 160), ("d3.0", 256, 512, 45, 80), ("d3.3", 512, 512, 45, 80), ("d4.0", 512, 1024, 22, 40),
                 ("d4.3", 1024, 1024, 22, 40), ("u1.skip", 512, 512, 45, 80), ("u1.3", 512, 512, 45, 80), ("u2.skip", 256, 256, 90, 160),
                 ("u2.3", 256, 256, 90, 160)):
        a, b, c = layer(*args)
        tot3, tot1, totx = tot3 + a, tot1 + b, totx + c
    print(f"sum over the nine launches: direct {tot3:.3f} ms; transformed-domain GEMMs alone {tot1:.3f} ms; + V traffic {totx / 4.5:.3f} ms "
          f"= {tot1 + totx / 4.5:.3f} ms -> x{(tot1 + totx / 4.5) / tot3:.2f} of the direct launches BEFORE any transform work")
