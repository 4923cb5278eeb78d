"""Summarise a rocprofv3 kernel trace: launches of kernels that are NOT this library's (at::native::*, rocm copy / fill kernels),
in launch order with their index, so that cold-start and steady-state launches can be told apart.
    python profiles/foreign_kernels.py gpurun_out/r06_cold > profiles/r06_cold_start_foreign_kernels.txt"""
import csv
import glob
import sys
from collections import Counter, OrderedDict

root = sys.argv[1]
files = glob.glob(root + "/**/*kernel_trace.csv", recursive=True)
rows = []
for f in files:
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# this library's kernels live in the global namespace; everything torch / the HIP runtime / RCCL launches is namespaced or prefixed
FOREIGN = ("at::native", "at::cuda", "__amd_rocclr", "rccl", "nccl", "c10::", "hipcub", "rocprim", "Cijk_", "thrust")
total = len(rows)
foreign = [(i, r["Kernel_Name"]) for i, r in enumerate(rows) if any(o in r["Kernel_Name"] for o in FOREIGN)]
print(f"{total} kernel launches in the trace, {len(foreign)} not from libsfh_amd.so")
cnt = Counter(n for _, n in foreign)
first = OrderedDict()
last = {}
for i, n in foreign:
    first.setdefault(n, i)
    last[n] = i
for n, c in cnt.most_common():
    print(f"{c:6d}  first launch #{first[n]:<6d} last #{last[n]:<6d} {n[:200]}")
own = Counter(r["Kernel_Name"].split("(")[0][:70] for r in rows if not any(o in r["Kernel_Name"] for o in FOREIGN))
print(f"\n{sum(own.values())} launches of {len(own)} kernels of libsfh_amd.so")
