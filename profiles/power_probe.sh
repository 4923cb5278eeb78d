#!/bin/bash
# Power / clock samples while bench.py runs (one gpurun call):  bash profiles/power_probe.sh [precision]
export SFH_PRECISION=${1:-f16x3}
python bench.py --no-cpu-baseline --no-extra-configs --steps 1200 --warmup 5 > gpurun_out/power_bench_$SFH_PRECISION.json 2> gpurun_out/power_bench.err &
BP=$!
n=0
while kill -0 $BP 2>/dev/null && [ $n -lt 400 ]; do
  L=$(rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Power \(W\)|sclk|junction" | sed 's/.*: //' | tr '\n' ' ')
  echo "$n $L"
  n=$((n+1))
  sleep 0.5
done | awk '{ if ($NF+0 > 400) print }' | head -40
wait $BP
python -c "import json;d=json.loads(open('gpurun_out/power_bench_$SFH_PRECISION.json').read().strip().splitlines()[-1]);print(d['value'], d['ms_per_step'], d['roofline']['frac'])"
