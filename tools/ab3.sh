#!/bin/bash
# tools/ab3.sh — GPU box: interleaved timing only (no parity check) of several dev builds, for measurement-only variants
# usage: tools/ab3.sh <rounds> <lib-tag> [<lib-tag> ...]
R=$1; shift
for i in $(seq $R); do
  for T in "$@"; do
    POLAR_AMD_LIB=$PWD/polar_amd/libpolar_amd_$T.so python bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-other-configs --mc-trials 0 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$T', round(d['value']), round(d['roofline']['kernel_ms_avg'],3), d['bler'])"
  done
done
