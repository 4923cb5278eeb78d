#!/bin/bash
# Kernel trace + PMC passes (separate, as the microarchitecture guide prescribes) over profiles/warpce_sweep.py:
#   bash profiles/warpce_pmc.sh r05
TAG=${1:-r05}
ROOTD=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOTD/gpurun_out
mkdir -p $OUT
rm -rf $OUT/${TAG}_wce_*
cd /tmp && export TMPDIR=/tmp
CMD="python3 $ROOTD/profiles/warpce_sweep.py --iters 5"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_wce_trace -- $CMD > $OUT/${TAG}_wce_trace.log 2>&1 && echo trace done
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $OUT/${TAG}_wce_sq -- $CMD > $OUT/${TAG}_wce_sq.log 2>&1 && echo sq done
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_wce_fetch -- $CMD > $OUT/${TAG}_wce_fetch.log 2>&1 && echo fetch done
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_wce_write -- $CMD > $OUT/${TAG}_wce_write.log 2>&1 && echo write done
