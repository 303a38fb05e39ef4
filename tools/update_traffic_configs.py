#!/usr/bin/env python3
"""Summarise tools/profile_configs.sh results (gpurun_out/prof_cfg_<name>/) into profiles/traffic_configs.json, keyed by the
sha256 of the library the profile was taken with (bench.py refuses the entry when the hash differs), and copy the
kernel-trace stats next to it. FETCH_SIZE x2 / WRITE_SIZE x1: profiles/r02/fetch_calibration.txt (8-B-per-lane rows).
usage: tools/update_traffic_configs.py profiles/r03 [name ...]"""
import collections, csv, glob, hashlib, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import importlib.util
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
dst = sys.argv[1]
names = sys.argv[2:]
KERN = {"config1": "sc8_decode_kernel", "config2": "sc8_decode_kernel", "config2_b262144": "sc8_decode_kernel",
        "config3": "scl_decode_llr_kernel<4, 3, 0, true", "config5": "scl_decode_llr_kernel<8, 3, 0, true",
        "config3_b262144": "scl_decode_llr_kernel<4, 3, 0, true", "config5_b262144": "scl_decode_llr_kernel<8, 3, 0, true"}
h = hashlib.sha256(open(os.path.join(ROOT, "polar_amd", "libpolar_amd.so"), "rb").read()).hexdigest()
path = os.path.join(ROOT, "profiles", "traffic_configs.json")
out = {"lib_sha256": h, "configs": {}}
if os.path.exists(path):
    old = json.load(open(path))
    if old.get("lib_sha256") == h:
        out = old
os.makedirs(dst, exist_ok=True)
for name in names:
    src = os.path.join(ROOT, "gpurun_out", "prof_cfg_" + name)
    kern = KERN[name]
    vals = {}
    for f in sorted(glob.glob(f"{src}/pmc*/pmc_counter_collection.csv")):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if kern in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            vals[k] = sum(v) / len(v)
    dur = None
    for r in csv.DictReader(open(f"{src}/trace/trace_kernel_stats.csv")):
        if kern in r["Name"]:
            dur = float(r["AverageNs"])
    shutil.copy(f"{src}/trace/trace_kernel_stats.csv", os.path.join(dst, f"cfg_{name}_kernel_stats.csv"))
    rd, wr = vals["FETCH_SIZE"] * 1024 * 2, vals["WRITE_SIZE"] * 1024
    out["configs"][name] = {"kernel": kern, "kernel_avg_ns": dur, "read_bytes_corrected": rd, "write_bytes": wr,
                            "traffic_bytes_per_launch": rd + wr,
                            "command": f"python bench.py --only-config {name} --steps 2 (tools/profile_configs.sh)"}
    print(name, out["configs"][name])
json.dump(out, open(path, "w"), indent=1)
