#!/bin/bash
# tools/sweep.sh — tuning sweep on the GPU box: waves/CU x lds_log for the bench workload
for cfg in "8 4" "6 4" "4 4" "12 3" "14 3" "10 3" "4 5" "3 5"; do
  set -- $cfg
  python bench.py --steps 2 --warmup 1 --batch 8192 --cpu-sample 0 --waves-per-cu $1 --lds-log $2 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('wpc=$1 lds_log=$2', round(d['value']), 'cw/s', round(d['roofline']['kernel_ms_avg'],2),'ms')"
done
