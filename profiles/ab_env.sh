#!/bin/bash
# A/B of ENVIRONMENT variants of the product library on ONE device (one gpurun call): alternating bench runs.
#   bash profiles/ab_env.sh <rounds> <name>=<ENV=VAL[,ENV=VAL...]|-> ...      ("-" = no extra environment)
# prints value / ms per step / unpipelined ms / per-group ms for every run
R=$1; shift
mkdir -p gpurun_out
for i in $(seq 1 $R); do
  for spec in "$@"; do
    name=${spec%%=*}; envs=${spec#*=}
    if [ "$envs" = "-" ]; then envs=""; fi
    ( for kv in ${envs//,/ }; do export "$kv"; done
      python bench.py --no-cpu-baseline --no-extra-configs --steps 20 > gpurun_out/abe_${name}_$i.json 2>> gpurun_out/abe.err ) || exit 1
  done
done
python - "$@" <<'PY'
import json, glob, sys
for spec in sys.argv[1:]:
    name = spec.split("=")[0]
    for f in sorted(glob.glob(f"gpurun_out/abe_{name}_*.json")):
        d = json.loads(open(f).read().strip().splitlines()[-1])
        u = d["roofline"].get("unpipelined", {})
        print(f"{name:12s} {d['value']:8.2f} fps {d['ms_per_step']:7.3f} ms | predict() {u.get('ms_per_step')} ms", {k: v["ms_per_step"] for k, v in d["kernel_groups"].items()})
PY
