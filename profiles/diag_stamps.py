"""Where does a conv workgroup spend its cycles?  Runs single conv layers through the
DIAGNOSTIC build (in-kernel s_memtime stamps, `python -m sfh_amd.build --diag`) and prints
the share of wave time per segment of the stage loop.  GPU box only:
    SFH_AMD_LIB=sports-field-homography_amd/libsfh_amd_diag.so python profiles/diag_stamps.py
Shares, not run times, are meaningful (the stamps fence the schedule)."""
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
SEG = ["prologue", "barrier1", "vmcnt+lds_write", "barrier2", "prefetch_issue", "mfma_block", "epilogue"]


def run(name, cin, cout, h, w, B=16, pool=False, reps=3):
    torch.manual_seed(0)
    wt = torch.randn(cout, cin, 3, 3, device="cuda") * 0.05
    bn = torch.nn.BatchNorm2d(cout).cuda().eval()
    pc = E.PackedConv(wt, torch.zeros(cout, device="cuda"), bn, 3, cin)
    hs, ws = (2 * h, 2 * w) if pool else (h, w)
    x = torch.randn(B, hs, ws, max(4, cin), device="cuda")
    y = torch.empty(B, h, w, cout, device="cuda")
    pc.run(x, B, h, w, y, pool0=pool)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 16)()
    rd(buf, 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        pc.run(x, B, h, w, y, pool0=pool)
    e1.record()
    torch.cuda.synchronize()
    rd(buf, 1)
    tot = sum(buf[i] for i in range(7))
    ms = e0.elapsed_time(e1) / reps
    tf = 2.0 * B * h * w * cout * 9 * cin / (ms * 1e-3) / 1e12
    print(f"{name:28s} {ms:7.3f} ms {tf:6.1f} TF(diag) | " + " ".join(f"{SEG[i]}={100.0*buf[i]/tot:5.1f}%" for i in range(7)))


if __name__ == "__main__":
    run("inc.3   64->64  360x640", 64, 64, 360, 640)
    run("d1.0p   64->128 180x320", 64, 128, 180, 320, pool=True)
    run("d1.3   128->128 180x320", 128, 128, 180, 320)
    run("d2.3   256->256  90x160", 256, 256, 90, 160)
    run("d3.3   512->512  45x80", 512, 512, 45, 80)
    run("d4.3 1024->1024  22x40", 1024, 1024, 22, 40)
