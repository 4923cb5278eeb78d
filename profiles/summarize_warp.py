"""Summarise the passes of profiles/warp_pmc.sh (homography warp micro-benchmark in its pmc mode: 1280x720,
batch 128, three launches per variant) into profiles/<tag>_warp_pmc.txt.
usage: python profiles/summarize_warp.py r02 [gpurun_out]"""
import collections
import csv
import glob
import os
import re
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
src = sys.argv[2] if len(sys.argv) > 2 else "gpurun_out"
here = os.path.dirname(os.path.abspath(__file__))
PIX = 128 * 720 * 1280
ALG = PIX * 4 + 720 * 1280 * 4 + 128 * 36


def short(n):
    m = re.search(r"(warp2?_kernel<[^>]*>|store_only)", n)
    return m.group(1) if m else None


def counters(kind):
    f = glob.glob(os.path.join(src, f"{tag}_warp_{kind}", "*", "*_counter_collection.csv"))
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    if f:
        for r in csv.DictReader(open(f[0])):
            k = short(r["Kernel_Name"])
            if k:
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items()}


def durations():
    f = glob.glob(os.path.join(src, f"{tag}_warp_trace", "*", "*_kernel_stats.csv"))
    out = {}
    if f:
        for r in csv.DictReader(open(f[0])):
            k = short(r["Name"])
            if k:
                out[k] = float(r["AverageNs"]) / 1e3
    return out


sq, sq2, fe, wr, tcp, dur = counters("sq"), counters("sq2"), counters("fetch"), counters("write"), counters("tcp"), durations()
lines = [f"# {tag}: rocprofv3 passes of profiles/micro/warp_variants (pmc mode): 1280x720, batch 128, nearest -> int32 unless noted",
         "# algorithmic bytes per launch: %d; SQ_ACTIVE_INST_* / SQ_WAVE_CYCLES / SQ_WAIT_* count quad-cycles" % ALG,
         "# valu_busy = 4 * SQ_ACTIVE_INST_VALU / (1024 SIMDs * GRBM_GUI_ACTIVE / 8): share of SIMD cycles issuing vector ALU work",
         "# hbm = FETCH_SIZE + WRITE_SIZE (KiB counters, dword gathers: no doubling); l2_hit = TCC_HIT / (TCC_HIT + TCC_MISS)",
         "%-36s %8s %7s %9s %10s %9s %9s %8s %8s" % ("kernel", "avg us", "TB/s", "VALU/px", "valu_busy", "wait_inst", "hbm MB", "vs alg", "l2 hit")]
for k in sorted(sq, key=lambda k: ("bilinear" if "<1" in k else "") + k):
    a, b = sq[k], sq2.get(k, {})
    us = dur.get(k, float("nan"))
    cyc = b.get("GRBM_GUI_ACTIVE", float("nan")) / 8.0
    busy = 4.0 * a["SQ_ACTIVE_INST_VALU"] / (1024.0 * cyc)
    hbm = (fe.get(k, {}).get("FETCH_SIZE", 0) + wr.get(k, {}).get("WRITE_SIZE", 0)) * 1024.0
    t = tcp.get(k, {})
    hit = t.get("TCC_HIT_sum", 0) / max(1.0, t.get("TCC_HIT_sum", 0) + t.get("TCC_MISS_sum", 0))
    lines.append("%-36s %8.1f %7.2f %9.1f %10.2f %9.2f %9.1f %8.2f %8.2f" % (
        k, us, ALG / us / 1e6, a["SQ_INSTS_VALU"] * 64.0 / PIX, busy, a["SQ_WAIT_INST_ANY"] / a["SQ_WAVE_CYCLES"],
        hbm / 1e6, hbm / ALG, hit))
open(os.path.join(here, f"{tag}_warp_pmc.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
