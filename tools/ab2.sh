#!/bin/bash
# tools/ab2.sh — GPU box: parity (exp-domain vs LLR-domain kernel, headline code) + interleaved timing of several dev builds
# usage: tools/ab2.sh <rounds> <lib-tag> [<lib-tag> ...]      (polar_amd/libpolar_amd_<tag>.so, built beforehand)
R=$1; shift
for T in "$@"; do
  POLAR_AMD_LIB=$PWD/polar_amd/libpolar_amd_$T.so ED_ONLY32=1 python tools/ed_check.py 4096 0 2>&1 | grep "TOTAL" | sed "s/^/$T parity: /"
done
for i in $(seq $R); do
  for T in "$@"; do
    POLAR_AMD_LIB=$PWD/polar_amd/libpolar_amd_$T.so python bench.py --steps 4 --warmup 1 --cpu-sample 0 --no-other-configs --mc-trials 0 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$T', round(d['value']), round(d['roofline']['kernel_ms_avg'],3))"
  done
done
