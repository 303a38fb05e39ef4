#!/bin/bash
# tools/ab_sc.sh — GPU box: list-size-1 configurations (1, 2, 2 at batch 262144) of several dev builds, interleaved
# usage: tools/ab_sc.sh <rounds> <lib-tag|default> ...      (tag "x+nofold" = library x with POLAR_SC_NO_FOLD=1)
R=$1; shift
for i in $(seq $R); do for T in "$@"; do
  LT=${T%%+*}; ENVX=""; [ "$T" != "$LT" ] && ENVX="POLAR_SC_NO_FOLD=1"
  if [ $LT = default ]; then L=$PWD/polar_amd/libpolar_amd.so; else L=$PWD/polar_amd/libpolar_amd_$LT.so; fi
  for C in config1 config2 config2_b262144; do
    env $ENVX POLAR_AMD_LIB=$L python bench.py --only-config $C --steps 4 --warmup 1 --cpu-sample 0 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$T $C', round(d['value']), d.get('gpu_vs_cpu_mismatching_codewords', d.get('cpu_baseline',{}).get('gpu_vs_cpu_mismatching_codewords')))"
  done
done; done
