#!/bin/bash
# A/B of alternative builds of the library on ONE device (one gpurun call): alternating bench runs.
#   bash profiles/ab_bench.sh <rounds> <name>=<lib.so|-> ...      ("-" = the product library)
# prints value / ms per step / roofline.frac / per-group ms for every run
R=$1; shift
mkdir -p gpurun_out
for i in $(seq 1 $R); do
  for spec in "$@"; do
    name=${spec%%=*}; lib=${spec#*=}
    if [ "$lib" = "-" ]; then
      python bench.py --no-cpu-baseline --no-extra-configs --steps 20 > gpurun_out/ab_${name}_$i.json 2>> gpurun_out/ab.err || exit 1
    else
      SFH_AMD_LIB=$PWD/$lib python bench.py --no-cpu-baseline --no-extra-configs --steps 20 > gpurun_out/ab_${name}_$i.json 2>> gpurun_out/ab.err || exit 1
    fi
  done
done
python - "$@" <<'PY'
import json, glob, sys
for spec in sys.argv[1:]:
    name = spec.split("=")[0]
    for f in sorted(glob.glob(f"gpurun_out/ab_{name}_*.json")):
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f"{name:10s} {d['value']:8.2f} fps {d['ms_per_step']:7.3f} ms frac {d['roofline']['frac']:.4f}", {k: v["ms_per_step"] for k, v in d["kernel_groups"].items()})
PY
