#!/bin/bash
# tools/round_collect.sh — HERE (after `gpurun -- bash tools/round_measure.sh <tag>` has merged gpurun_out/ back): summarise the session
# into profiles/<round>/ and refresh profiles/traffic.json / traffic_configs.json (keyed by the sha256 of the library in the tree, which
# must be the one the session measured). usage: tools/round_collect.sh <tag> <round>     e.g. r05c r05
set -e
T=$1; R=$2; O=gpurun_out/$T; P=profiles/$R
test "$(cut -d' ' -f1 $O/lib_sha256.txt)" = "$(sha256sum polar_amd/libpolar_amd.so | cut -d' ' -f1)" || { echo "the library in the tree is not the one session $T measured"; exit 1; }
mkdir -p $P
python tools/pmc_summary.py gpurun_out/prof_$T $P/final_headline "bench.py --cpu-sample 0 --steps 2 --warmup 1 --no-other-configs --mc-trials 0 (tools/profile.sh $T)" > /dev/null
python tools/update_traffic.py $P/final_headline_pmc.json 11 1024 16 32 262144 > /dev/null
python tools/update_traffic_configs.py $P config1 config2 config2_b262144 config3 config5 config3_b262144 config5_b262144 > /dev/null
for f in bench.json bench_b524288.json bler_sweeps.json config4_record.json lib_sha256.txt gpu_tests.txt stress_parity.txt fuzz_sane.txt fuzz_any.txt fuzz_p1.txt \
         fresh_out_probe.json traffic_replay_raw.txt sc_rounds.json icache_pmc.txt stall_pmc.txt cache_footprint_microbench.txt latency_table.json host_path_65536.json host_path_262144.json \
         host_trace_config3.txt lat_pmc.txt sc_p1_time.txt lone_wave_microbench.txt stress_parity_lat.txt; do cp $O/$f $P/ 2>/dev/null || echo "missing $f"; done
cp gpurun_out/fuzz_slice.json $P/ 2>/dev/null || true
for c in config3 config5; do cp $O/pmc/$c.txt $P/final_pmc_$c.txt 2>/dev/null || true; done
python - <<EOF
import json
d = json.loads(open("$P/bench.json").read().strip().splitlines()[-1])
mc = d["monte_carlo"]
print("headline %.4f M cw/s, %.1f ms/step, kernel %.1f ms" % (d["value"] / 1e6, d["ms_per_step"], d["roofline"]["kernel_ms_avg"]))
print("monte_carlo %.4f M trials/s (weak %.4f, native %.4f), counters_equal %s" % (mc["mc_trials_per_s"] / 1e6, mc["mc_weak"]["mc_trials_per_s"] / 1e6, mc["native_multi"].get("mc_trials_per_s", 0) / 1e6, mc["counters_equal_single_gpu"]))
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["gpu_vs_cpu_mismatching_codewords"], d["cpu_baseline"]["all_cores"]["value"], d["cpu_baseline"]["all_cores"]["gpu_vs_cpu_mismatching_codewords"])
for c in d["other_configs"]: print(c["config"], round(c["value"] / 1e6, 3), round(c["roofline"]["kernel_ms_avg"], 3), c.get("cpu_baseline", {}).get("gpu_vs_cpu_mismatching_codewords"))
for c in d["host_batch"]["configs"]:
    for r in c["rows"]: print(c["config"], r["llr"], round(r["value"] / 1e6, 3), round(r["frac_of_bound"], 2), r["bound_by"], r["bits_equal_device_resident"])
t = json.load(open("profiles/traffic.json")); print("traffic", t["traffic_bytes_per_launch"], t["kernel_avg_ns_in_profile"], t["valu_busy_frac_in_profile"], t["lib_sha256"][:8])
EOF
cat $P/gpu_tests.txt; tail -n 1 $P/stress_parity.txt $P/fuzz_sane.txt $P/fuzz_any.txt $P/fuzz_p1.txt
