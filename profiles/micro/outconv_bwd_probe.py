"""sfh_outconv_bwd at 640x360 x 16 (64 channels, 4 classes).  usage: [SFH_AMD_LIB=<variant>] python profiles/micro/outconv_bwd_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import sfh_amd  # noqa
from sfh_amd import _lib
from sfh_amd.engine import _ptr, _stream
lib = _lib.load()
B, H, W, C, NC = 16, 360, 640, 64, 4
y = torch.randn(B, H, W, C, device="cuda"); w = torch.randn(NC, C, device="cuda"); dl = torch.randn(B, NC, H, W, device="cuda")
dy = torch.empty_like(y); aw = torch.zeros(NC * C, dtype=torch.float64, device="cuda"); ab = torch.zeros(NC, dtype=torch.float64, device="cuda")
fn = lambda: _lib.check(lib.sfh_outconv_bwd(_ptr(y), C, _ptr(w), _ptr(dl), NC, B, H, W, _ptr(dy), _ptr(aw), _ptr(ab), _stream()), "outconv_bwd")
for _ in range(3): fn()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): fn()
e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) / 20
gb = (2 * y.numel() + dl.numel()) * 4 / 1e9
print(f"outconv_bwd {t * 1e3:.1f} us = {gb / t:.2f} TB/s")
