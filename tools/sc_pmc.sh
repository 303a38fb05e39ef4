#!/bin/bash
# tools/sc_pmc.sh — GPU box: counters of the list-size-1 kernels (separate passes, counters only)
R=$PWD; cd /tmp && export TMPDIR=/tmp
i=0
for PMC in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM" \
           "GRBM_GUI_ACTIVE GRBM_COUNT SQ_INSTS_SMEM SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $PMC -d $R/gpurun_out/scpmc$i -o pmc --output-format csv -- python $R/tools/sc_time.py 262144 2 > $R/gpurun_out/scpmc$i.log 2>&1
done
python - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$R/gpurun_out/scpmc*/pmc_counter_collection.csv")):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:40]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    for k in acc:
        if k.startswith("sc8"):
            print(k, {c: f"{v:.4g}" for c, v in acc[k].items()})
PY
