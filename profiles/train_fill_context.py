"""Which launches of a training step are stock torch fills / copies, and between which of the library's kernels do they sit?
Reads the rocprofv3 kernel trace of `bench.py --train --steps 2 --warmup 1`; prints, for the LAST step, every foreign launch
pattern (previous own kernel -> foreign kernel -> next own kernel) with its count.
    python profiles/train_fill_context.py gpurun_out/r06_train_trace"""
import csv
import glob
import re
import sys
from collections import Counter

rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
FOREIGN = ("at::native", "__amd_rocclr", "at::cuda")


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    return re.split(r"[<(]", n)[0][:40] if not any(f in n for f in FOREIGN) else re.sub(r"std::array.*", "", n)[:90]


names = [short(r["Kernel_Name"]) for r in rows]
opt = [i for i, n in enumerate(names) if n.startswith(("rmsprop", "sgd_", "adam_"))]
a, b = (opt[-2] + 1, opt[-1] + 1) if len(opt) >= 2 else (0, len(rows))
step = names[a:b]
own = [i for i, n in enumerate(step) if not any(f in n for f in FOREIGN)]
print(f"last step: {len(step)} launches, {len(step) - len(own)} foreign")
ctx = Counter()
for i, n in enumerate(step):
    if any(f in n for f in FOREIGN):
        prev = next((step[j] for j in range(i - 1, -1, -1) if j in set(own)), "-")
        nxt = next((step[j] for j in range(i + 1, len(step)) if not any(f in step[j] for f in FOREIGN)), "-")
        ctx[(prev, n, nxt)] += 1
for (p, n, x), c in ctx.most_common(40):
    print(f"{c:4d}  {p:40s} -> {n:90s} -> {x}")
cnt = Counter(step)
print("\nlaunches per kernel (last step):")
for n, c in cnt.most_common(60):
    print(f"{c:5d}  {n}")
