#!/bin/bash
# kernel trace of the ResNet-STN part with and without split-K (one gpurun call): bash profiles/trace_resnet.sh <tag>
TAG=${1:-r03}
ROOTD=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOTD/gpurun_out
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-extra-configs"
rm -rf $OUT/${TAG}_rn_split $OUT/${TAG}_rn_nosplit
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/${TAG}_rn_split -- python3 $ROOTD/bench.py $ARGS > $OUT/${TAG}_rn_split.log 2>&1
export SFH_SPLITK=0
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/${TAG}_rn_nosplit -- python3 $ROOTD/bench.py $ARGS > $OUT/${TAG}_rn_nosplit.log 2>&1
cd $ROOTD
for v in split nosplit; do
  f=$(ls $OUT/${TAG}_rn_$v/*/*kernel_trace.csv | head -1)
  echo "== $v"; python3 profiles/resnet_table.py $f > $OUT/${TAG}_resnet_table_$v.txt; tail -80 $OUT/${TAG}_resnet_table_$v.txt
done
