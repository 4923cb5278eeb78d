"""Summarise profiles/train_trace.sh output: per-kernel launches and time per training step, and the matrix-pipe
occupancy of the conv / backward-filter kernels.   python profiles/summarize_train.py <tag> [steps traced = 3]"""
import csv
import glob
import collections
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r02h"
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
import os
f = max(glob.glob(f"gpurun_out/{tag}_train_trace/*/*_kernel_trace.csv"), key=os.path.getmtime)   # newest (gpurun merges)
rows = list(csv.DictReader(open(f)))
agg = collections.OrderedDict()
for r in rows:
    k = r["Kernel_Name"]
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    a = agg.setdefault(k, [0, 0.0])
    a[0] += 1
    a[1] += d
tot = sum(v[1] for v in agg.values())
out = [f"# rocprofv3 --kernel-trace --stats -- python3 bench.py --train --steps 2 --warmup 1   ({nsteps} steps, B=16, 640x360; profiles/train_trace.sh)",
       f"# per-step figures = totals / {nsteps}; GPU kernel time per step: {tot / nsteps:.1f} ms; launches per step: {len(rows) // nsteps}"]
for k, (n, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:32]:
    out.append(f"{k[:118]:118s} calls/step={n / nsteps:6.1f}  ms/step={ms / nsteps:7.2f}  avg_us={ms / n * 1e3:8.1f}  {100 * ms / tot:5.1f}%")
g = glob.glob(f"gpurun_out/{tag}_train_mfma/*/*_counter_collection.csv")
if g:
    c = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(max(g, key=os.path.getmtime))):
        n = r["Kernel_Name"]
        key = "conv_s3_kernel" if "conv_s3_kernel" in n else "wgrad_s3_kernel" if "wgrad_s3_kernel" in n else "wgrad_kernel (fp32)" if "wgrad_kernel" in n else "conv_mfma_kernel (fp32)" if "conv_mfma" in n else None
        if key:
            c[key][r["Counter_Name"]] += float(r["Counter_Value"])
    out.append("# matrix-pipe occupancy (separate --pmc pass, one step): SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs)")
    for k, v in c.items():
        if v.get("GRBM_GUI_ACTIVE"):
            out.append(f"{k:28s} mfma busy frac {v['SQ_VALU_MFMA_BUSY_CYCLES'] / (v['GRBM_GUI_ACTIVE'] / 8 * 1024):.3f}")
open(f"profiles/{tag}_train_kernel_stats.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out[:28]))
