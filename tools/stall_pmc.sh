#!/bin/bash
# tools/stall_pmc.sh — GPU box: latency / stall counters of the bench workload's list kernel (average VMEM / LDS / SMEM latency =
# SQ_INST_LEVEL_x / SQ_INSTS_x, scalar-cache misses, FIFO-full stalls). usage: [POLAR_AMD_LIB=...] tools/stall_pmc.sh <tag>
R=$PWD; T=${1:-x}; cd /tmp && export TMPDIR=/tmp
i=0
for PMC in "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_INST_LEVEL_SMEM SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_DCACHE_MISSES_DUPLICATE SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" \
           "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" \
           "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INSTS_FLAT SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $PMC -d $R/gpurun_out/st_$T/p$i -o pmc --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --cpu-sample 0 --no-other-configs --mc-trials 0 > $R/gpurun_out/st_$T/log$i.txt 2>&1
done
python - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$R/gpurun_out/st_$T/p*/pmc_counter_collection.csv")):
    acc = collections.defaultdict(float); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        if "0, true" in r["Kernel_Name"] and "<32" in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    print({c: f"{v / n[c]:.4g}" for c, v in acc.items()})
PY
