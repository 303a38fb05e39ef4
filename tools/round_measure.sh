#!/bin/bash
# tools/round_measure.sh — GPU box (via gpurun): the full measurement set of a round on the library in the tree, results under
# gpurun_out/<tag>/ (copy what is to be judged into profiles/<round>/ with tools/pmc_summary.py, update_traffic*.py).
# usage: tools/round_measure.sh <tag>
T=${1:-r06}; O=gpurun_out/$T; mkdir -p $O
sha256sum polar_amd/libpolar_amd.so > $O/lib_sha256.txt
(timeout 1800 python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -3) > $O/gpu_tests.txt
python bench.py > $O/bench.json 2> $O/bench.err
python bench.py --batch 524288 --cpu-sample 0 --no-other-configs --mc-trials 0 > $O/bench_b524288.json 2>> $O/bench.err
bash tools/profile.sh ${T} --steps 2 --warmup 1 --no-other-configs --mc-trials 0 > /dev/null 2>&1
bash tools/profile_configs.sh config1 config2 config2_b262144 config3 config5 config3_b262144 config5_b262144 > $O/profile_configs.log 2>&1
python tools/host_path.py --out $O/host_path_65536.json > $O/host_path_65536.txt 2>&1
python tools/host_path.py --batch 262144 --out $O/host_path_262144.json > $O/host_path_262144.txt 2>&1
(cd /tmp && export TMPDIR=/tmp && cd $OLDPWD && rocprofv3 --kernel-trace --output-format csv -d gpurun_out/host_trace_$T -- python tools/host_trace.py config3 0 > $O/host_trace_config3.log 2>&1; python tools/host_trace.py --report gpurun_out/host_trace_$T > $O/host_trace_config3.txt 2>&1)
rm -rf gpurun_out/ic1 gpurun_out/ic2; bash tools/icache_pmc.sh 2>&1 | tail -2 > $O/icache_pmc.txt
mkdir -p gpurun_out/st_$T; bash tools/stall_pmc.sh $T 2>&1 | tail -4 > $O/stall_pmc.txt
python tools/bler_sweeps.py $O/bler_sweeps.json > $O/bler_sweeps.txt 2>&1
python tools/latency_table.py $O/latency_table.json > $O/latency_table.txt 2>&1
python tools/stress_parity.py 10 > $O/stress_parity.txt 2>&1
STRESS_SET=2 python tools/stress_parity.py 4 >> $O/stress_parity.txt 2>&1
FUZZ_SANE=1 python tools/fuzz_parity.py 300 11 > $O/fuzz_sane.txt 2>&1
python tools/fuzz_parity.py 150 12 > $O/fuzz_any.txt 2>&1
hipcc --offload-arch=gfx950 -O3 tools/mall_microbench.hip -o /tmp/mb 2>/dev/null && /tmp/mb > $O/cache_footprint_microbench.txt
hipcc --offload-arch=gfx950 -O3 tools/traffic_replay.hip -o /tmp/tr 2>/dev/null && /tmp/tr 8 > $O/traffic_replay_raw.txt
python tools/fresh_out_probe.py --out $O/fresh_out_probe.json > $O/fresh_out_probe.txt 2>&1
bash tools/config_pmc.sh $O/pmc config3 config5 > /dev/null 2>&1
python tools/sc_rounds.py $O/sc_rounds.json > $O/sc_rounds.txt 2>&1
python tools/config4_record.py 100 8388608 > $O/config4_record.json 2>> $O/bench.err
python tools/fuzz_parity_p1.py 60 > $O/fuzz_p1.txt 2>&1
bash tools/lat_pmc.sh $O/lat_pmc.txt 2 4 8 > /dev/null 2>&1
python tools/sc_p1_time.py > $O/sc_p1_time.txt 2>&1
python tools/stress_parity_lat.py 8192 > $O/stress_parity_lat.txt 2>&1
hipcc --offload-arch=gfx950 -O3 tools/lone_wave_microbench.hip -o /tmp/lw 2>/dev/null && /tmp/lw > $O/lone_wave_microbench.txt
cat $O/gpu_tests.txt; tail -c 400 $O/bench_b524288.json; tail -n 2 $O/stress_parity.txt $O/fuzz_sane.txt $O/fuzz_any.txt
