"""Summarise the passes of profiles/warpce_pmc.sh (profiles/warpce_sweep.py --iters 5 under rocprofv3: kernel trace, SQ counters,
FETCH_SIZE, WRITE_SIZE in separate passes) into profiles/<tag>_warpce_pmc.txt: per kernel and grid - i.e. per (size, batch) -
median kernel duration, TB/s of the fused launch's algorithmic bytes, vector instructions per pixel, HBM bytes by the counters
(FETCH_SIZE doubled as the microarchitecture guide prescribes for gfx950, WRITE_SIZE as is; both in KiB).
usage: python profiles/summarize_warpce.py r05 [gpurun_out]"""
import collections
import csv
import glob
import os
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r05"
src = sys.argv[2] if len(sys.argv) > 2 else "gpurun_out"
CFG = [((640, 360), 16), ((640, 360), 128), ((1280, 720), 16), ((1280, 720), 128)]
KERNELS = ("warpce_kernel", "warpce_final_kernel", "warp2_kernel", "ce_partial_kernel", "ce_final_kernel")


def kname(n):
    for k in KERNELS:
        if k + "<" in n or k + "(" in n:
            return k
    return None


def by_cfg(rows, key):
    """dispatches of each kernel in order: the sweep runs its four configurations one after the other, the same number of
    launches of a kernel in each -> {(kernel, cfg index): [rows]}"""
    per = collections.defaultdict(list)
    for r in rows:
        k = kname(r[key])
        if k:
            per[k].append(r)
    out = {}
    for k, lst in per.items():
        n = len(lst) // len(CFG)
        for i in range(len(CFG)):
            out[(k, i)] = lst[i * n:(i + 1) * n]
    return out


trace = max(glob.glob(os.path.join(src, f"{tag}_wce_trace", "*", "*_kernel_trace.csv")), key=os.path.getmtime)
tr = by_cfg(sorted(csv.DictReader(open(trace)), key=lambda r: int(r["Start_Timestamp"])), "Kernel_Name")
cnt = {}
for kind in ("sq", "fetch", "write"):
    f = max(glob.glob(os.path.join(src, f"{tag}_wce_{kind}", "*", "*_counter_collection.csv")), key=os.path.getmtime)
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Dispatch_Id"]))
    for cname in sorted({r["Counter_Name"] for r in rows}):
        cnt[cname] = by_cfg([r for r in rows if r["Counter_Name"] == cname], "Kernel_Name")

lines = ["nearest warp + consistency CE: fused (warpce_kernel + warpce_final_kernel) against separate (warp2_kernel + ce_partial_kernel + ce_final_kernel)",
         "rocprofv3 passes of `python profiles/warpce_sweep.py --iters 5` (profiles/warpce_pmc.sh); median over the launches of a configuration", ""]
for i, ((w, h), B) in enumerate(CFG):
    px = B * h * w
    nbytes = px * 4 * 5 + h * w * 4 + B * 36
    lines.append(f"{w}x{h} batch {B}: {px} pixels, fused algorithmic bytes {nbytes / 1e6:.1f} MB (logits once + mask once + template)")
    tot = {}
    for k in KERNELS:
        rows = tr.get((k, i))
        if not rows:
            continue
        d = sorted((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows)
        med = d[len(d) // 2]
        tot[k] = med
        extra = ""
        c = cnt.get("SQ_INSTS_VALU", {}).get((k, i))
        if c and k in ("warpce_kernel", "warp2_kernel", "ce_partial_kernel"):
            v = sorted(float(r["Counter_Value"]) for r in c)[len(c) // 2]
            extra += f"  {v * 64 / px:6.1f} vector instructions per pixel"
        f_, w_ = cnt.get("FETCH_SIZE", {}).get((k, i)), cnt.get("WRITE_SIZE", {}).get((k, i))
        if f_ and w_ and k in ("warpce_kernel", "warp2_kernel", "ce_partial_kernel"):
            fv = sorted(float(r["Counter_Value"]) for r in f_)[len(f_) // 2] * 1024
            wv = sorted(float(r["Counter_Value"]) for r in w_)[len(w_) // 2] * 1024
            extra += f"  HBM by counters: FETCH {fv / 1e6:7.1f} MB (x2 = {2 * fv / 1e6:7.1f}) + WRITE {wv / 1e6:7.1f} MB"
        lines.append(f"    {k:22s} {med:9.2f} us" + extra)
    if "warpce_kernel" in tot:
        f1 = tot["warpce_kernel"]
        f2 = f1 + tot.get("warpce_final_kernel", 0.0)
        sep = sum(tot.get(k, 0.0) for k in ("warp2_kernel", "ce_partial_kernel", "ce_final_kernel"))
        lines.append(f"    fused kernel alone: {nbytes / f1 / 1e6:5.2f} TB/s = {nbytes / f1 / 8e6:5.3f} of 8 TB/s;  with its final launch: "
                     f"{nbytes / f2 / 1e6:5.2f} TB/s = {nbytes / f2 / 8e6:5.3f};  separate kernels: {sep:8.2f} us -> x{sep / f2:4.2f}")
    lines.append("")
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), f"{tag}_warpce_pmc.txt")
open(out, "w").write("\n".join(lines))
print("\n".join(lines))
