#!/bin/bash
# tools/tune_sweep.sh — GPU box: headline bench over (waves_per_cu, lds_log) settings
for cfg in "0 0" "16 4" "16 3" "12 4" "8 4" "8 3" "8 5" "16 5"; do
  set -- $cfg
  python bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-other-configs --mc-trials 0 --waves-per-cu $1 --lds-log $2 2>&1 | tail -1 | python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read()); print('wpc $1 lds_log $2:', round(d['value']), round(d['roofline']['kernel_ms_avg'],3))
except Exception as e: print('wpc $1 lds_log $2: failed', e)"
done
