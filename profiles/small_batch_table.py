"""Per-kernel table of ONE steady-state batch from a rocprofv3 kernel trace of profiles/small_batch_trace.py (the last batch:
launches between the last two frame_to_h2 launches), in launch order: duration, grid size, short kernel name.
    python profiles/small_batch_table.py gpurun_out/r06_sb1"""
import csv
import glob
import re
import sys

rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "frame_to_h2" in r["Kernel_Name"]]
b = rows[starts[-2]:starts[-1]]
t0 = int(b[0]["Start_Timestamp"])
tot = 0
for i, r in enumerate(b):
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    tot += d
    name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
    name = re.sub(r"^void ", "", name)[:86]
    wg = int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 256)) or 256)
    grid = int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0) * int(r.get("Grid_Size_Y", 1) or 1) * int(r.get("Grid_Size_Z", 1) or 1)
    print(f"{i:3d} +{(int(r['Start_Timestamp']) - t0) / 1e3:8.1f} us  {d / 1e3:7.1f} us  wgs {grid // max(wg, 1):6d}  {name}")
print(f"total kernel time {tot / 1e3:.1f} us over {len(b)} launches")
