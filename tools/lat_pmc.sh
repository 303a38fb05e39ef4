#!/bin/bash
# tools/lat_pmc.sh — GPU box (via gpurun): instruction counts and stall cycles of the one-codeword-per-wave list kernels at B = 1
# (is a leaf step issue-bound or latency-bound?). Separate rocprofv3 --pmc passes of tools/lat_kernel_time.py, kernel trace only;
# dispatches of ONE wave (grid 64) are summarised per kernel instantiation. usage: tools/lat_pmc.sh <out.txt> [L ...]
set -u
REPO=$(pwd); OUT=$REPO/$1; shift
LS="${*:-2 4 8}"
cd /tmp && export TMPDIR=/tmp
W=/tmp/latpmc; rm -rf $W; mkdir -p $W
i=0
for PMC in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM" \
           "SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH SQ_IFETCH" \
           "SQ_INST_LEVEL_LDS SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_THREAD_CYCLES_VALU" \
           "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $PMC -d $W/pmc$i -o pmc --output-format csv -- python $REPO/tools/lat_kernel_time.py $LS > $W/log$i.txt 2>&1
done
python - "$W" > $OUT <<'PY'
import collections, csv, glob, sys
w = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
def grid(r):
    for k in ("Grid_Size", "Grid_Size_X"):
        if k in r: return int(r[k]) * int(r.get("Grid_Size_Y", 1) or 1) * int(r.get("Grid_Size_Z", 1) or 1)
    raise KeyError(str(list(r.keys())))
for f in glob.glob(w + "/pmc*/**/pmc_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if ("decode" not in r["Kernel_Name"] and "lat_kernel" not in r["Kernel_Name"]) or grid(r) != 64: continue
        agg[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob(w + "/pmc1/**/pmc_kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if ("decode" not in r["Kernel_Name"] and "lat_kernel" not in r["Kernel_Name"]) or grid(r) != 64: continue
        dur[r["Kernel_Name"].split("(")[0]].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
for k in sorted(agg):
    c = {n: sum(v) / len(v) for n, v in agg[k].items()}
    ns = sum(dur[k]) / max(1, len(dur[k]))
    print("kernel", k, "one wave; avg_us %.1f" % (ns / 1e3), "launches", len(dur[k]))
    for n in sorted(c): print("  %-24s %.5g" % (n, c[n]))
PY
cat $OUT
