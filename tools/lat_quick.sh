#!/bin/bash
# tools/lat_quick.sh — GPU box: the latency-kernel tests, B = 1 call times and the lone-wave counters in one go. usage: tools/lat_quick.sh <tag>
O=gpurun_out/$1; mkdir -p $O
(timeout 900 python -m pytest tests -x -q -m gpu -k "one_codeword or latency or lat or small" 2>&1 | grep -E "passed|failed|error" | tail -3) > $O/tests.txt
python tools/lat_kernel_time.py 2 4 8 > $O/lat_time.txt 2>&1
bash tools/lat_pmc.sh $O/lat_pmc.txt 2 4 8 > /dev/null 2>&1
cat $O/tests.txt; grep "B=1:" $O/lat_time.txt
grep -E "kernel|SQ_WAVE_CYCLES|SQ_WAIT_ANY|SQ_INSTS_VALU |SQ_INSTS_SALU|SQ_INSTS_LDS|SQ_INSTS_BRANCH" $O/lat_pmc.txt
