#!/bin/bash
# PMC evidence for the homography warp kernel (run through gpurun from the repo root):
#   bash profiles/warp_pmc.sh r02
# Separate passes (kernel-trace alone; SQ counters; TCC FETCH_SIZE; TCC WRITE_SIZE), on the stand-alone
# micro-benchmark in its "pmc" mode (1280x720, batch 128, three launches per variant).
TAG=${1:-r02}
ROOTD=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOTD/gpurun_out
BIN=$ROOTD/profiles/micro/warp_variants
mkdir -p $OUT
rm -rf $OUT/${TAG}_warp_*
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_warp_trace -- $BIN 30 pmc > $OUT/${TAG}_warp_trace.log 2>&1 && echo trace done
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_WAVES --output-format csv -d $OUT/${TAG}_warp_sq -- $BIN 30 pmc > $OUT/${TAG}_warp_sq.log 2>&1 && echo sq done
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INSTS_SALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/${TAG}_warp_sq2 -- $BIN 30 pmc > $OUT/${TAG}_warp_sq2.log 2>&1 && echo sq2 done
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_warp_fetch -- $BIN 30 pmc > $OUT/${TAG}_warp_fetch.log 2>&1 && echo fetch done
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_warp_write -- $BIN 30 pmc > $OUT/${TAG}_warp_write.log 2>&1 && echo write done
timeout -k 10 200 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/${TAG}_warp_tcp -- $BIN 30 pmc > $OUT/${TAG}_warp_tcp.log 2>&1 && echo tcp done
ls $OUT/${TAG}_warp_*/*/ | head -40
