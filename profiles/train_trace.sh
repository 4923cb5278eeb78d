#!/bin/bash
# Kernel trace of the training step (BASELINE config 3) on the GPU box: bash profiles/train_trace.sh r02
TAG=${1:-r02}
ROOTD=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOTD/gpurun_out
mkdir -p $OUT
rm -rf $OUT/${TAG}_train_trace
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_train_trace -- python3 $ROOTD/bench.py --train --steps 2 --warmup 1 > $OUT/${TAG}_train_trace.log 2>&1
echo "trace done"
ls $OUT/${TAG}_train_trace/*/ | head
# matrix-pipe occupancy of the training kernels (separate pass)
rm -rf $OUT/${TAG}_train_mfma
timeout -k 10 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/${TAG}_train_mfma -- python3 $ROOTD/bench.py --train --steps 1 --warmup 1 > $OUT/${TAG}_train_mfma.log 2>&1
echo "mfma pass done"
