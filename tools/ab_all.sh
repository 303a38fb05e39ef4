#!/bin/bash
# tools/ab_all.sh — interleaved A/B of library builds in ONE gpurun call: headline + configs 2, 3, 5 (decode kernels' HIP-event times)
# usage: tools/ab_all.sh <rounds> <lib.so> <lib.so> ...
R=$1; shift
for i in $(seq $R); do
  for L in "$@"; do
    POLAR_AMD_LIB=$PWD/$L python bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-other-configs --mc-trials 0 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('headline', '$L'[-16:], round(d['value']), round(d['roofline']['kernel_ms_avg'],3))"
    for c in config2 config3 config5; do
      POLAR_AMD_LIB=$PWD/$L python bench.py --only-config $c --steps 3 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$c', '$L'[-16:], round(d['value']), round(d['roofline']['kernel_ms_avg'],3))"
    done
  done
done
