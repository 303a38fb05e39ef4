#!/usr/bin/env python3
"""Record the HBM traffic per launch (rocprofv3 PMC passes, corrected as MI355X_MICROARCH.md §HBM
prescribes — see tools/pmc_summary.py) for the bench.py default workload, so that bench.py can
report roofline.traffic next to the live-measured kernel time.
usage: tools/update_traffic.py profiles/r01/<name>_pmc.json n K crc L batch"""
import hashlib
import json
import os
import sys

src, n, K, crc, L, batch = sys.argv[1], *map(int, sys.argv[2:7])
d = json.load(open(src))
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = {"config": {"n": n, "K": K, "crc": crc, "L": L, "batch": batch},
       # the library this profile was taken with: bench.py reports the counters only when it runs the same file
       "lib_sha256": hashlib.sha256(open(os.path.join(ROOT, "polar_amd", "libpolar_amd.so"), "rb").read()).hexdigest(),
       "traffic_bytes_per_launch": d["hbm_traffic_per_launch"]["total_bytes"],
       "read_bytes_corrected": d["hbm_traffic_per_launch"]["read_bytes_corrected"],
       "write_bytes": d["hbm_traffic_per_launch"]["write_bytes"],
       "kernel_avg_ns_in_profile": d["kernel_time"]["avg_ns"],
       "source": src, "command": d["command"]}
c = d.get("counters", {})
if "GRBM_GUI_ACTIVE" in c and "SQ_ACTIVE_INST_VALU" in c:
    # GRBM_GUI_ACTIVE is summed over the 8 XCDs; SQ_* VALU activity is in quad-cycles over 1024 SIMDs
    cyc = c["GRBM_GUI_ACTIVE"]["mean_per_launch"] / 8.0
    out["shader_clock_ghz_in_profile"] = cyc / d["kernel_time"]["avg_ns"]
    out["valu_busy_frac_in_profile"] = c["SQ_ACTIVE_INST_VALU"]["mean_per_launch"] * 4.0 / (1024.0 * cyc)
if "SQ_INSTS_VALU" in c:
    out["valu_insts_per_wave_decode"] = c["SQ_INSTS_VALU"]["mean_per_launch"] / (batch / 2.0)     # L = 32: two codewords per wave
if "SQ_INSTS_LDS" in c:
    out["lds_insts_per_launch"] = c["SQ_INSTS_LDS"]["mean_per_launch"]
if "SQ_LDS_BANK_CONFLICT" in c and "SQ_ACTIVE_INST_LDS" in c and c["SQ_ACTIVE_INST_LDS"]["mean_per_launch"] > 0:
    out["lds_bank_conflict_frac"] = c["SQ_LDS_BANK_CONFLICT"]["mean_per_launch"] / c["SQ_ACTIVE_INST_LDS"]["mean_per_launch"]
json.dump(out, open("profiles/traffic.json", "w"), indent=1)
print(out)
