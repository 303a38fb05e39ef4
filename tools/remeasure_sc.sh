T=r03j; O=gpurun_out/$T; mkdir -p $O
sha256sum polar_amd/libpolar_amd.so > $O/lib_sha256.txt
(timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -3) > $O/gpu_tests.txt
bash tools/profile.sh ${T} --steps 2 --warmup 1 --no-other-configs > /dev/null 2>&1
bash tools/profile_configs.sh config1 config2 config2_b262144 config3 config5 > $O/profile_configs.log 2>&1
python tools/sc_rounds.py $O/sc_rounds.json > $O/sc_rounds.txt 2>&1
python tools/latency_table.py $O/latency_table.json > $O/latency_table.txt 2>&1
FUZZ_SANE=1 python tools/fuzz_parity.py 300 21 > $O/fuzz_sane.txt 2>&1
python tools/stress_parity.py 40 > $O/stress_parity_big.txt 2>&1
cat $O/gpu_tests.txt; tail -1 $O/fuzz_sane.txt; tail -2 $O/stress_parity_big.txt
