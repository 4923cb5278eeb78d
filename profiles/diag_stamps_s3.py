"""Where does a split-operand conv workgroup (conv_s3_kernel, single-buffer variant) spend its wave time?
Format from argv[1]: s3 (three bf16 planes, default) or h2 (two fp16 planes).
Runs single layers through the DIAGNOSTIC build (in-kernel s_memtime stamps, `python -m sfh_amd.build --diag`):
    SFH_AMD_LIB=sports-field-homography_amd/libsfh_amd_diag.so python profiles/diag_stamps_s3.py
Shares of wave time per phase are meaningful, not run times (the stamps fence the schedule)."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("SFH_AMD_LIB", os.path.join(ROOT, "sports-field-homography_amd", "libsfh_amd_diag.so"))
import torch  # noqa: E402
from sfh_amd import _lib, engine as E  # noqa: E402

lib = _lib.load()
rd = lib.sfh_debug_read_stamps
rd.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
FMT = sys.argv[1] if len(sys.argv) > 1 else "s3"
SEG = ["prologue", "dma wait+barrier", "mfma stage", "free barrier+dma issue", "epilogue"]
# launches that take the straight-line H2 epilogue (no residual): the epilogue is stamped in parts
SEG_FAST = ["prologue", "dma wait+barrier", "mfma stage", "free barrier+dma issue", "epi: addresses + scale/shift wait",
            "epi: pass 1", "epi: pooled output", "epi: report"]


def run(name, cin, cout, h, w, B=16, pool=False, residual=False, reps=3):
    torch.manual_seed(0)
    wt = torch.randn(cout, cin, 3, 3, device="cuda") * 0.05
    bn = torch.nn.BatchNorm2d(cout).cuda().eval()
    pc = E.PackedConv(wt, torch.zeros(cout, device="cuda"), bn, 3, cin, fmt=FMT)
    x = E.f32_to_split(torch.randn(B, h, w, cin, device="cuda"), FMT)
    y = E.split_empty(FMT, B, h, w, cout, "cuda")
    yp = E.split_empty(FMT, B, h // 2, w // 2, cout, "cuda") if pool else None
    res = torch.randn(B, h, w, cout, device="cuda") if residual else None
    pc.run(x, B, h, w, y, dst_pool=yp, residual=res)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 16)()
    rd(buf, 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        pc.run(x, B, h, w, y, dst_pool=yp, residual=res)
    e1.record()
    torch.cuda.synchronize()
    rd(buf, 1)
    fast = False
    names = SEG
    seg = [buf[8 + i] for i in range(len(names))]
    tot = sum(seg) or 1   # the clock-only build (libsfh_amd_clock.so) stamps no phases
    ms = e0.elapsed_time(e1) / reps
    tf = 2.0 * B * h * w * cout * 9 * cin / (ms * 1e-3) / 1e12
    ghz = 0.1 * buf[14] / buf[15] if (buf[15] and not fast) else float("nan")   # shader cycles per 100 MHz tick, over the stamped waves
    print(f"{name:34s} {ms:7.3f} ms {tf:6.1f} TF(diag) in-kernel clock {ghz:4.2f} GHz | "
          + "  ".join(f"{names[i]} {100.0 * seg[i] / tot:5.1f}%" for i in range(len(names))), flush=True)


if __name__ == "__main__":
    run("inc.3   64->64  360x640 +pool", 64, 64, 360, 640, pool=True)
    run("u4.skip 64->64  360x640 +res f32", 64, 64, 360, 640, residual=True)
    run("u4.3    64->64  360x640", 64, 64, 360, 640)
    run("d1.3   128->128 180x320 +pool", 128, 128, 180, 320, pool=True)
    run("d2.3   256->256  90x160 +pool", 256, 256, 90, 160, pool=True)
    run("d3.3   512->512  45x80", 512, 512, 45, 80)
    run("d4.3 1024->1024  22x40", 1024, 1024, 22, 40)
    run("     512->512   360x80 (14 rounds)", 512, 512, 360, 80)
