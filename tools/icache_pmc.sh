#!/bin/bash
# tools/icache_pmc.sh — GPU box: instruction-cache counters of the bench workload's decode kernel
R=$PWD; cd /tmp && export TMPDIR=/tmp
i=0
for PMC in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_IFETCH_LEVEL" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_BRANCH SQ_INSTS_CBRANCH_TAKEN SQ_INSTS_CBRANCH_NOT_TAKEN"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $PMC -d $R/gpurun_out/ic$i -o pmc --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --cpu-sample 0 --no-other-configs --mc-trials 0 > $R/gpurun_out/ic$i.log 2>&1
  tail -1 $R/gpurun_out/ic$i.log | cut -c1-100
done
python - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$R/gpurun_out/ic*/pmc_counter_collection.csv")):
    acc = collections.defaultdict(float); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        if "0, true" in r["Kernel_Name"] and "<32" in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    print({c: f"{v / n[c]:.4g}" for c, v in acc.items()})
PY
