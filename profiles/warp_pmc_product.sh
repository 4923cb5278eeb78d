#!/bin/bash
# PMC evidence for the PRODUCT homography warp kernel through the C-ABI (run through gpurun from the repo root):
#   bash profiles/warp_pmc_product.sh r04
# Separate passes (kernel-trace alone; SQ counters; FETCH_SIZE; WRITE_SIZE) over profiles/warp_sweep.py (both sizes,
# batches 16 / 128 / 1024, nearest -> int32 and bilinear -> f32, the launcher's own rows per wave).
TAG=${1:-r04}
ROOTD=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOTD/gpurun_out
mkdir -p $OUT
rm -rf $OUT/${TAG}_wp_*
cd /tmp && export TMPDIR=/tmp
CMD="python3 $ROOTD/profiles/warp_sweep.py --iters 5"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_wp_trace -- $CMD > $OUT/${TAG}_wp_trace.log 2>&1 && echo trace done
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $OUT/${TAG}_wp_sq -- $CMD > $OUT/${TAG}_wp_sq.log 2>&1 && echo sq done
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_wp_fetch -- $CMD > $OUT/${TAG}_wp_fetch.log 2>&1 && echo fetch done
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_wp_write -- $CMD > $OUT/${TAG}_wp_write.log 2>&1 && echo write done
