#!/bin/bash
# tools/sweep.sh — tuning sweep on the GPU box: "waves/CU lds_log" pairs for the bench workload
CFGS=${1:-"8:4 12:3 14:3 10:4"}
for cfg in $CFGS; do
  w=${cfg%%:*}; l=${cfg##*:}
  python bench.py --steps 2 --warmup 1 --batch ${BATCH:-8192} --cpu-sample 0 --mc-trials 0 --no-other-configs --waves-per-cu $w --lds-log $l 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('wpc=$w lds_log=$l', round(d['value']), 'cw/s', round(d['roofline']['kernel_ms_avg'],2),'ms')"
done
