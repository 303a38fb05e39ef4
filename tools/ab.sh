#!/bin/bash
# tools/ab.sh — interleaved A/B of two library builds in ONE gpurun call (same box, same clocks)
# usage: tools/ab.sh <libA.so> <libB.so> [rounds] [bench args...]
# (bench.py does not rebuild when POLAR_AMD_LIB is set: build both libraries BEFORE the call, e.g. the
#  baseline from a stashed tree into its own POLAR_BUILD_TAG — never compare against the default name)
A=$1; B=$2; R=${3:-3}; shift 3
for i in $(seq $R); do
  for L in $A $B; do
    POLAR_AMD_LIB=$PWD/$L python bench.py --steps 3 --warmup 1 --cpu-sample 0 "$@" 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$L', round(d['value']), round(d['roofline']['kernel_ms_avg'],3))"
  done
done
