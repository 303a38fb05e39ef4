#!/bin/bash
# tools/config_pmc.sh — GPU box (via gpurun): where a BASELINE configuration's decode kernel spends its time — VALU occupancy,
# instruction mix, stalls, HBM bytes — from separate rocprofv3 --pmc passes of `bench.py --only-config <name>` (counters in their own
# runs, kernel trace only: MI355X_MICROARCH.md). usage: tools/config_pmc.sh <out-dir> <name>...   -> <out-dir>/<name>.txt
set -u
REPO=$(pwd); OUTD=$REPO/$1; shift; mkdir -p $OUTD
cd /tmp && export TMPDIR=/tmp
for NAME in "$@"; do
  W=/tmp/cpmc_$NAME; rm -rf $W; mkdir -p $W
  BENCH="python $REPO/bench.py --only-config $NAME --steps 2"
  i=0
  for PMC in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
             "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM" \
             "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
    i=$((i+1))
    rocprofv3 --kernel-trace --pmc $PMC -d $W/pmc$i -o pmc --output-format csv -- $BENCH > $W/log$i.txt 2>&1
  done
  python - "$NAME" "$W" > $OUTD/$NAME.txt <<'PY'
import collections, csv, glob, sys
name, w = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob(w + "/pmc*/**/pmc_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob(w + "/pmc1/**/pmc_kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Kernel_Name"].split("(")[0]].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
dom = max((k for k in agg if "decode" in k), key=lambda k: sum(dur.get(k, [0])))
c = {k: sum(v) / len(v) for k, v in agg[dom].items()}
ns = sum(dur[dom]) / len(dur[dom])
print("config", name, "kernel", dom, "avg_ms %.3f" % (ns / 1e6), "launches", len(dur[dom]))
for k in sorted(c): print("  %-24s %.4g" % (k, c[k]))
if "GRBM_GUI_ACTIVE" in c:
    cyc = c["GRBM_GUI_ACTIVE"] / 8.0
    print("  shader clock GHz %.3f" % (cyc / ns))
    if "SQ_ACTIVE_INST_VALU" in c: print("  VALU busy frac %.3f" % (c["SQ_ACTIVE_INST_VALU"] * 4 / (1024 * cyc)))
    if "SQ_ACTIVE_INST_SCA" in c: print("  SALU busy frac (per SIMD-quad) %.3f" % (c["SQ_ACTIVE_INST_SCA"] * 4 / (1024 * cyc)))
if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
    b = c["FETCH_SIZE"] * 2048 + c["WRITE_SIZE"] * 1024
    print("  HBM-side bytes %.4g (%.2f TB/s)" % (b, b / ns / 1e3))
if "SQ_WAVES" in c and c["SQ_WAVES"]:
    for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SMEM"):
        if k in c: print("  %s per wave %.0f" % (k, c[k] / c["SQ_WAVES"]))
PY
  cat $OUTD/$NAME.txt
done
