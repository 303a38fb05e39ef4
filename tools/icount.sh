#!/bin/bash
# tools/icount.sh — GPU box: deterministic instruction counts of the dominant kernel for several dev builds
# usage: tools/icount.sh <lib-tag> [...]
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
for T in "$@"; do
  rm -rf /tmp/ic_$T
  POLAR_AMD_LIB=$REPO/polar_amd/libpolar_amd_$T.so rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d /tmp/ic_$T -o pmc --output-format csv -- python $REPO/bench.py --steps 2 --warmup 1 --cpu-sample 0 --no-other-configs --mc-trials 0 > /tmp/ic_$T.log 2>&1
  python - "$T" <<'PY'
import csv, glob, collections, sys
T = sys.argv[1]
agg = collections.defaultdict(list)
for f in glob.glob(f"/tmp/ic_{T}/**/pmc_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "scl_decode" in k and ("true" in k or "Lb1" in k or "false" not in k):
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
print(T, " ".join(f"{k}={sum(v)/len(v)/32768:.0f}" for k, v in sorted(agg.items())), "(per wave-decode)")
PY
done
