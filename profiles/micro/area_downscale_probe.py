"""Time of the uint8 HWC -> float NCHW preprocessing kernels on device-resident frames (batch 16): plain /255, 3x3 area (1920x1080 ->
640x360), 2x2 (1280x720 -> 640x360), generic (1600x900 -> 640x360).  usage: python profiles/micro/area_downscale_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import sfh_amd  # noqa
from sfh_amd import engine as E, synth
def bench(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
B = 16
for (sh, sw) in ((360, 640), (720, 1280), (1080, 1920), (900, 1600)):
    u8 = torch.from_numpy(synth.synth_frames_u8(B, sh, sw, seed=1)).cuda()
    t = bench(lambda: E.frames_u8_to_input(u8, (640, 360)) if (sh, sw) != (360, 640) else E.frames_u8_to_input(u8))
    gb = (u8.numel() + B * 3 * 360 * 640 * 4) / 1e9
    print(f"{sw}x{sh} -> 640x360: {t * 1e3:7.1f} us  ({gb / t:5.2f} TB/s of {gb * 1e3:.0f} MB)", flush=True)
