"""How busy is the GPU inside the pipelined (predict_async) region?  From the kernel trace of
`bench.py --no-alone-pass` (profiles/collect_pmc.sh: <tag>_trace_pipe): span of the last three steps, union of the kernel
intervals (= time at least one kernel is running), sum of the kernel durations (> span: launches of the two batches overlap),
and the largest gaps.   usage: python profiles/pipelined_gaps.py r04 [gpurun_out]  ->  profiles/<tag>_pipelined_gaps.txt"""
import csv
import glob
import os
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
src = sys.argv[2] if len(sys.argv) > 2 else "gpurun_out"
here = os.path.dirname(os.path.abspath(__file__))
f = max(glob.glob(os.path.join(src, f"{tag}_trace_pipe", "*", "*kernel_trace.csv")), key=os.path.getmtime)
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
first = [i for i, r in enumerate(rows) if "frame_to_h2" in r["Kernel_Name"] or "nchw_to_nhwc" in r["Kernel_Name"]]
a, b = first[-4], first[-1]
seg = rows[a:b]
t0, t1 = int(seg[0]["Start_Timestamp"]), int(rows[b]["Start_Timestamp"])
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in seg)
busy, (cs, ce), gaps = 0, iv[0], []
for s, e in iv[1:]:
    if s > ce:
        busy += ce - cs
        gaps.append(s - ce)
        cs, ce = s, e
    else:
        ce = max(ce, e)
busy += ce - cs
n = 3.0
lines = [f"# {tag}: kernel trace of the pipelined region (rocprofv3 --kernel-trace -- python3 bench.py --steps 5 --warmup 1 --no-alone-pass ...), last three steps",
         f"step span                         {(t1 - t0) / n / 1e6:8.3f} ms",
         f"at least one kernel running       {busy / n / 1e6:8.3f} ms   ({100.0 * busy / (t1 - t0):.1f} % of the span)",
         f"sum of kernel durations           {sum(e - s for s, e in iv) / n / 1e6:8.3f} ms   (launches of two batches overlap)",
         f"idle gaps                         {len(gaps) / n:8.1f} per step, {sum(gaps) / n / 1e3:.1f} us per step in total; largest (us): "
         + ", ".join(f"{g / 1e3:.1f}" for g in sorted(gaps, reverse=True)[:8])]
open(os.path.join(here, f"{tag}_pipelined_gaps.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
