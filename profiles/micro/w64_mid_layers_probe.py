"""The 64-couts-per-wave plain conv (profiles/removed/conv_w64.hip.txt, built into a variant library) against the product kernel on the
UNet's layer shapes at 640x360 x 16: same descriptor, the launcher swapped.  usage: SFH_AMD_LIB=<variant with sfh_conv_w64_fwd> python ..."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import sfh_amd  # noqa
from sfh_amd import engine as E, _lib
import ctypes
lib = _lib.load()
w64 = lib.sfh_conv_w64_fwd
w64.restype = ctypes.c_int
w64.argtypes = [ctypes.POINTER(_lib.ConvDesc), ctypes.c_void_p]
def bench(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
B = 16
real = lib.sfh_conv_s3_fwd
for cin, cout, h, w in ((64, 64, 360, 640), (64, 128, 180, 320), (128, 128, 180, 320), (256, 256, 90, 160), (512, 512, 45, 80)):
    torch.manual_seed(0)
    wt = torch.randn(cout, cin, 3, 3, device="cuda") * (2.0 / (9 * cin)) ** 0.5
    bn = torch.nn.BatchNorm2d(cout).cuda().eval()
    conv = E.PackedConv(wt, torch.zeros(cout, device="cuda"), bn, 3, cin, fmt="h2", tag="probe")
    x = E.f32_to_h2(torch.relu(torch.randn(B, h, w, cin, device="cuda")))
    y0, y1 = E.split_empty("h2", B, h, w, cout, "cuda"), E.split_empty("h2", B, h, w, cout, "cuda")
    t0 = bench(lambda: conv.run(x, B, h, w, y0, small=False))
    lib.sfh_conv_s3_fwd = w64
    try:
        t1 = bench(lambda: conv.run(x, B, h, w, y1, small=False, wg_couts=0))
    finally:
        lib.sfh_conv_s3_fwd = real
    same = torch.equal(y0, y1)
    gf = 2.0 * B * h * w * cout * 9 * cin / 1e9
    print(f"{cin:4d}->{cout:<4d} {h}x{w}: product {t0:6.3f} ms ({gf / t0:5.1f} TFLOP/s)   64 couts per wave {t1:6.3f} ms ({gf / t1:5.1f})   {100 * (t1 / t0 - 1):+5.1f} %   identical {same}", flush=True)
