#!/bin/bash
# Collect the rocprofv3 evidence for one round on the GPU box (run through gpurun from the
# repo root):  bash profiles/collect_pmc.sh r01
# Separate passes, as MI355X_MICROARCH.md prescribes: kernel-trace+stats alone; FETCH_SIZE
# alone (3 TCC slots); WRITE_SIZE alone (2 slots).  Outputs land in gpurun_out/<tag>_*/ and
# are summarised into profiles/ by profiles/summarize_pmc.py (run back in the build container).
set -e
TAG=${1:-r01}
ROOTD=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOTD/gpurun_out
mkdir -p $OUT
rm -rf $OUT/${TAG}_trace $OUT/${TAG}_trace_pipe $OUT/${TAG}_trace_np $OUT/${TAG}_fetch $OUT/${TAG}_write $OUT/${TAG}_mfma
cd /tmp && export TMPDIR=/tmp
echo "start $(date +%T)"
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-extra-configs"
# the default command (two batches in flight): kernel stats that bench.py's roofline.avg_launch_ms must agree with
timeout -k 10 420 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_trace -- python3 $ROOTD/bench.py $ARGS > $OUT/${TAG}_trace.log 2>&1
echo "trace done $(date +%T)"
# the pipelined region ONLY (no unpipelined pass behind it): every launch counted ran with two batches in flight
timeout -k 10 420 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_trace_pipe -- python3 $ROOTD/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extra-configs --no-alone-pass > $OUT/${TAG}_trace_pipe.log 2>&1
echo "trace (pipelined region only) done $(date +%T)"
# one predict() per step: launches alone on the chip, in order (per-layer table; the counter passes below serialise anyway)
ARGS="$ARGS --no-pipeline"
timeout -k 10 420 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_trace_np -- python3 $ROOTD/bench.py $ARGS > $OUT/${TAG}_trace_np.log 2>&1
echo "trace (no pipeline) done $(date +%T)"
timeout -k 10 420 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_fetch -- python3 $ROOTD/bench.py $ARGS > $OUT/${TAG}_fetch.log 2>&1
echo "fetch done $(date +%T)"
timeout -k 10 420 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_write -- python3 $ROOTD/bench.py $ARGS > $OUT/${TAG}_write.log 2>&1
echo "write done $(date +%T)"
timeout -k 10 420 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $OUT/${TAG}_mfma -- python3 $ROOTD/bench.py $ARGS > $OUT/${TAG}_mfma.log 2>&1
echo "mfma done $(date +%T)"
ls $OUT/${TAG}_*/*/ | head -40
