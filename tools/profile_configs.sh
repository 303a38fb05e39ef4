#!/bin/bash
# tools/profile_configs.sh — GPU box (via gpurun): kernel-trace stats + FETCH_SIZE / WRITE_SIZE passes of every OTHER
# BASELINE.json configuration on its own workload (`bench.py --only-config <name>`), counters in their own runs as
# MI355X_MICROARCH.md prescribes. Results under gpurun_out/prof_cfg_<name>/; summarise with tools/update_traffic_configs.py.
set -u
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
for NAME in "$@"; do
  OUT=$REPO/gpurun_out/prof_cfg_$NAME
  rm -rf $OUT; mkdir -p $OUT
  BENCH="python $REPO/bench.py --only-config $NAME --steps 2"
  rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace --output-format csv -- $BENCH > $OUT/bench_trace.log 2>&1
  i=0
  for PMC in "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    rocprofv3 --kernel-trace --pmc $PMC -d $OUT/pmc$i -o pmc --output-format csv -- $BENCH > $OUT/bench_pmc$i.log 2>&1
  done
  tail -1 $OUT/bench_trace.log | cut -c1-300
done
