#!/bin/bash
# SQ / TCP / TCC counter passes over bench.py (3 steps), one pass per counter set (no tracing beside --pmc):
#   bash profiles/sq_counters.sh <tag>      -> gpurun_out/<tag>_sqA|sqB|tc/
set -e
TAG=${1:-r02h}
ROOTD=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOTD/gpurun_out
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-extra-configs"
rm -rf $OUT/${TAG}_sqA $OUT/${TAG}_sqB $OUT/${TAG}_tc
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $OUT/${TAG}_sqA -- python3 $ROOTD/bench.py $ARGS > $OUT/${TAG}_sqA.log 2>&1
echo A
timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_INSTS_VALU --output-format csv -d $OUT/${TAG}_sqB -- python3 $ROOTD/bench.py $ARGS > $OUT/${TAG}_sqB.log 2>&1
echo B
timeout -k 10 300 rocprofv3 --pmc TCC_HIT TCC_MISS TCP_TCC_READ_REQ TCP_TCC_READ_REQ_LATENCY TCP_PENDING_STALL_CYCLES --output-format csv -d $OUT/${TAG}_tc -- python3 $ROOTD/bench.py $ARGS > $OUT/${TAG}_tc.log 2>&1
echo C
