"""Ordered launch table of the LAST training step in a rocprofv3 kernel trace of `bench.py --train`: start offset, duration,
workgroups, short kernel name - to see which backward-filter / backward-data launches sit far from their layer's forward rate.
    python profiles/train_step_table.py gpurun_out/r06_train_trace [min_us]"""
import csv
import glob
import re
import sys

rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 100.0


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    return n[:100]


opt = [i for i, r in enumerate(rows) if re.search(r"rmsprop_kernel|sgd_kernel|adam_kernel", r["Kernel_Name"])]
a, b = (opt[-2] + 1, opt[-1] + 1) if len(opt) >= 2 else (0, len(rows))
t0 = int(rows[a]["Start_Timestamp"])
tot = 0
for i, r in enumerate(rows[a:b]):
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot += d
    if d >= min_us:
        wg = int(r.get("Workgroup_Size_X", 256) or 256)
        grid = int(r.get("Grid_Size_X", 0) or 0) * int(r.get("Grid_Size_Y", 1) or 1) * int(r.get("Grid_Size_Z", 1) or 1)
        print(f"{i:4d} +{(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} us {d:8.1f} us  wgs {grid // max(wg, 1):6d}  {short(r['Kernel_Name'])}")
span = (int(rows[b - 1]["End_Timestamp"]) - t0) / 1e3
print(f"step: {b - a} launches, kernel time {tot / 1e3:.2f} ms, span {span / 1e3:.2f} ms")
