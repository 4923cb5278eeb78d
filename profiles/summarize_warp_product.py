"""Summarise the passes of profiles/warp_pmc_product.sh (the product warp kernel through the C-ABI, profiles/warp_sweep.py)
into profiles/<tag>_warp_pmc.txt: per (kernel instantiation, grid) = (mode, size, batch): average duration, TB/s of
algorithmic bytes, vector instructions per pixel, share of SIMD cycles issuing vector work, HBM bytes against algorithmic.
usage: python profiles/summarize_warp_product.py r04 [gpurun_out]"""
import collections
import csv
import glob
import os
import re
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
src = sys.argv[2] if len(sys.argv) > 2 else "gpurun_out"
here = os.path.dirname(os.path.abspath(__file__))
# profiles/warp_sweep.py --iters 5 launches, in this order, 3 warm-up + 5 timed launches per configuration
ORDER = [(mode, size, B) for size in ((640, 360), (1280, 720)) for B in (16, 128, 1024) for mode in ("nearest", "bilinear")]
PER = 8


def in_order(path, name_col):
    """warp2_kernel dispatches of one pass in dispatch order -> [(config, row)]"""
    rows = [r for r in csv.DictReader(open(path)) if "warp2_kernel" in r[name_col]]
    return rows


def counters(kind):
    f = glob.glob(os.path.join(src, f"{tag}_wp_{kind}", "*", "*_counter_collection.csv"))
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    if f:
        rows = [r for r in csv.DictReader(open(max(f, key=os.path.getmtime))) if "warp2_kernel" in r["Kernel_Name"]]
        ids = sorted({int(r["Dispatch_Id"]) for r in rows})
        rank = {d: i for i, d in enumerate(ids)}
        assert len(ids) == PER * len(ORDER), (len(ids), PER * len(ORDER))
        for r in rows:
            i = rank[int(r["Dispatch_Id"])]
            if i % PER >= 3:      # skip the warm-up launches
                acc[ORDER[i // PER]][r["Counter_Name"]].append((float(r["Counter_Value"]), 0))
    return acc


def durations():
    f = glob.glob(os.path.join(src, f"{tag}_wp_trace", "*", "*_kernel_trace.csv"))
    acc = collections.defaultdict(list)
    if f:
        rows = [r for r in csv.DictReader(open(max(f, key=os.path.getmtime))) if "warp2_kernel" in r["Kernel_Name"]]
        rows.sort(key=lambda r: int(r["Start_Timestamp"]))
        assert len(rows) == PER * len(ORDER), (len(rows), PER * len(ORDER))
        for i, r in enumerate(rows):
            if i % PER >= 3:
                acc[ORDER[i // PER]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    return {k: sum(v) / len(v) / 1e3 for k, v in acc.items()}


sq, fe, wr, dur = counters("sq"), counters("fetch"), counters("write"), durations()
mean = lambda lst: sum(v for v, _ in lst) / len(lst) if lst else float("nan")
lines = [f"# {tag}: rocprofv3 passes of the PRODUCT warp kernel (warp2_kernel<mode, J, rows per thread, out>: the round-3 kernel) through the C-ABI: profiles/warp_sweep.py",
         "# algorithmic bytes = B*h*w*4 + ht*wt*4 + 36*B; VALU/px = SQ_INSTS_VALU * 64 / pixels; valu_busy = 4 * SQ_ACTIVE_INST_VALU / (1024 SIMDs * GRBM_GUI_ACTIVE / 8)",
         "# hbm = (FETCH_SIZE + WRITE_SIZE) KiB (dword gathers: no doubling); durations from the kernel-trace pass (5 launches behind 3 warm-up launches)",
         "%-9s %-9s %6s %8s %7s %9s %10s %9s %8s" % ("mode", "size", "batch", "avg us", "TB/s", "VALU/px", "valu_busy", "hbm MB", "vs alg")]
for k in ORDER:
    if k not in dur:
        continue
    mode, (w, h), B = k
    pix = B * h * w
    alg = pix * 4 + h * w * 4 + 36 * B
    a = sq.get(k, {})
    us = dur[k]
    cyc = mean(a.get("GRBM_GUI_ACTIVE", [])) / 8.0
    busy = 4.0 * mean(a.get("SQ_ACTIVE_INST_VALU", [])) / (1024.0 * cyc) if a else float("nan")
    hbm = (mean(fe.get(k, {}).get("FETCH_SIZE", [])) + mean(wr.get(k, {}).get("WRITE_SIZE", []))) * 1024.0
    lines.append("%-9s %-9s %6d %8.1f %7.2f %9.1f %10.2f %9.1f %8.2f" % (
        mode, f"{w}x{h}", B, us, alg / us / 1e6, mean(a.get("SQ_INSTS_VALU", [])) * 64.0 / pix, busy, hbm / 1e6, hbm / alg))
open(os.path.join(here, f"{tag}_warp_pmc.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
