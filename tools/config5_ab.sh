#!/bin/bash
# tools/config5_ab.sh — GPU box: configuration 5 (and 3) on the round-3, round-4 and current libraries, ONE box, five interleaved
# rounds (round-4 verdict item 5: the driver-timed 7.13 -> 6.51 M cw/s between r03 and r04). Libraries built from the round-end
# commits into polar_amd/_ab/ (git worktree + python -m polar_amd.build there; not committed).
for r in 1 2 3 4 5; do
  for l in ${LIBS:-r03 r04 cur}; do
    L=polar_amd/_ab/libpolar_$l.so; [ $l = cur ] && L=polar_amd/libpolar_amd.so
    for c in config5 config3; do
      POLAR_AMD_LIB=$L python bench.py --only-config $c --steps 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('round $r', '$l', d['config'], '%.3f M cw/s' % (d['value']/1e6), 'kernel %.3f ms' % d['roofline']['kernel_ms_avg'])"
    done
  done
done
