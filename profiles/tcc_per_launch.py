"""L2 (TCC) hits / misses per launch shape from `rocprofv3 --pmc TCC_HIT TCC_MISS` over `bench.py --no-pipeline` (review item 6: where
do the composed 2x2 and the single-kernel Up launches get their bytes from?).  A request = one 128-byte line; misses go to the
fabric (Infinity Cache / HBM).
    python profiles/tcc_per_launch.py gpurun_out/r06_tcc > profiles/r06_tcc_per_launch.txt"""
import collections
import csv
import glob
import re
import sys

rows = []
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))


ALL = "--all" in sys.argv


def label(n):
    if ALL:      # every kernel of the library, by its (shortened) name: training steps
        if "at::native" in n or "rocclr" in n:
            return None
        n = re.sub(r"\(anonymous namespace\)::", "", n)
        n = re.sub(r"^void ", "", n)
        return re.split(r"\(", n)[0][:64]
    m = re.search(r"S3Cfg<([^>]*)>, (true|false)", n)
    if m:
        return f"conv_s3 S3Cfg<{m.group(1)}> DB={m.group(2)}"
    for k in ("conv_upfused_kernel", "conv_small_kernel", "conv3x3_c4h2_kernel", "stem7x7_kernel", "outconv_kernel", "warp"):
        if k in n:
            return k
    return None


agg = collections.OrderedDict()
for r in rows:
    k = label(r["Kernel_Name"])
    if k is None:
        continue
    key = (k, r["Grid_Size"])
    a = agg.setdefault(key, collections.defaultdict(list))
    a[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("# TCC_HIT / TCC_MISS per launch (requests of 128 B), averaged over the launches of one (kernel instance, grid) = one layer shape;")
print("# miss MB = TCC_MISS x 128 B: what the launch pulls over the fabric (Infinity Cache or HBM)")
print("%-64s %10s %4s %14s %14s %8s %10s" % ("kernel instance", "grid", "n", "TCC_HIT", "TCC_MISS", "hit %", "miss MB"))
for (k, grid), a in agg.items():
    h = a.get("TCC_HIT", a.get("TCC_HIT_sum", [0]))
    m = a.get("TCC_MISS", a.get("TCC_MISS_sum", [0]))
    H, M = sum(h) / max(len(h), 1), sum(m) / max(len(m), 1)
    if H + M == 0:
        continue
    if ALL and M * 128 / 1e6 * len(h) < 50:      # training table: launches that matter
        continue
    print("%-64s %10s %4d %14.0f %14.0f %8.1f %10.1f" % (k[:64], grid, len(h), H, M, 100 * H / (H + M), M * 128 / 1e6))
