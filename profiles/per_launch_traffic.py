"""Per-launch-shape HBM-side traffic of the conv launches from the separate --pmc passes of profiles/collect_pmc.sh
(FETCH_SIZE and WRITE_SIZE, KiB; raw counter values - the microarchitecture guide's doubling of FETCH_SIZE for wide
reads is NOT applied here, the algorithmic bytes beside them say how to read each row).
usage: python profiles/per_launch_traffic.py r04 [gpurun_out]  ->  profiles/<tag>_pmc_per_launch.txt"""
import collections
import csv
import glob
import os
import re
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
src = sys.argv[2] if len(sys.argv) > 2 else "gpurun_out"
here = os.path.dirname(os.path.abspath(__file__))


def load(kind, counter):
    f = glob.glob(os.path.join(src, f"{tag}_{kind}", "*", "*counter_collection.csv"))
    rows = [r for r in csv.DictReader(open(max(f, key=os.path.getmtime))) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    return rows


def label(r):
    m = re.search(r"S3Cfg<([^>]*)>, (true|false)", r["Kernel_Name"])
    if m:
        return f"S3Cfg<{m.group(1)}> DB={m.group(2)}", r["Grid_Size"]
    if "conv3x3_c4h2" in r["Kernel_Name"]:
        return "conv3x3_c4h2 (first layer, fp16 cores)", r["Grid_Size"]
    if "conv_upfused_kernel" in r["Kernel_Name"]:
        return "conv_upfused_kernel (single-kernel Up block: u3 / u4)", r["Grid_Size"]
    m = re.search(r"conv_small_kernel<(true|false)>", r["Kernel_Name"])
    if m:
        return f"conv_small_kernel<DB={m.group(1)}>", r["Grid_Size"]
    return None


agg = collections.OrderedDict()
for rows, name in ((load("fetch", "FETCH_SIZE"), "F"), (load("write", "WRITE_SIZE"), "W")):
    for r in rows:
        k = label(r)
        if k is None:
            continue
        a = agg.setdefault(k, {"F": [], "W": [], "ns": []})
        a[name].append(float(r["Counter_Value"]))
        if name == "F":
            a["ns"].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
lines = [f"# {tag}: raw FETCH_SIZE / WRITE_SIZE per launch (MiB), averaged over the launches of one (kernel instance, grid) = one layer shape of the",
         "# UNet / ResNet at 640x360 batch 16 (bench.py --no-pipeline under rocprofv3 --pmc, separate passes); us = launch duration in the PMC run",
         "%-52s %10s %4s %12s %12s %10s" % ("kernel instance <KS, stride, SH, SW, TH, TW, planes, NWN, NWM, stats>", "grid", "n", "FETCH MiB", "WRITE MiB", "us")]
for (k, grid), a in agg.items():
    if len(a["F"]) < 2:
        continue
    F = sum(a["F"]) / len(a["F"]) / 1024
    W = sum(a["W"]) / max(1, len(a["W"])) / 1024
    ns = sum(a["ns"]) / len(a["ns"])
    lines.append("%-52s %10s %4d %12.1f %12.1f %10.1f" % (k, grid, len(a["F"]), F, W, ns / 1e3))
open(os.path.join(here, f"{tag}_pmc_per_launch.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
