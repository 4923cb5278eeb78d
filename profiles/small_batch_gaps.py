"""Reads a rocprofv3 kernel trace of profiles/small_batch_trace.py: for the last N batches (a batch = the launches between two
frame_to_h2 launches) the sum of kernel durations, the span first-start -> last-end, and the idle time inside the span.
    python profiles/small_batch_gaps.py gpurun_out/r06_sb1"""
import csv
import glob
import sys

rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "frame_to_h2" in r["Kernel_Name"]]
batches = [rows[a:b] for a, b in zip(starts[:-1], starts[1:])][-10:]
for k, b in enumerate(batches):
    own = [r for r in b if "rocclr" not in r["Kernel_Name"]]
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in own)
    span = int(own[-1]["End_Timestamp"]) - int(own[0]["Start_Timestamp"])
    print(f"batch {k}: {len(own)} launches, kernel time {busy / 1e3:.1f} us, span {span / 1e3:.1f} us, idle inside {100 * (1 - busy / span):.1f} %, "
          f"mean launch {busy / len(own) / 1e3:.1f} us")
