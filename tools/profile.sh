#!/bin/bash
# tools/profile.sh — run on the GPU box via gpurun: kernel-trace stats + PMC passes of bench.py.
# Usage: tools/profile.sh <tag> [bench args...]; results under gpurun_out/prof_<tag>/
set -u
TAG=${1:-r01}; shift || true
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench.py --cpu-sample 0 $*"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace --output-format csv -- $BENCH > $OUT/bench_trace.log 2>&1
# PMC passes: separate runs, counters only (no trace domains besides kernel-trace)
i=0
for PMC in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE GRBM_COUNT" \
           "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum" \
           "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_DRAM_sum TCC_EA0_WRREQ_64B_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $PMC -d $OUT/pmc$i -o pmc --output-format csv -- $BENCH > $OUT/bench_pmc$i.log 2>&1
done
ls -R $OUT | head -50
