#!/usr/bin/env python3
"""Summarise a tools/profile.sh result directory (gpurun_out/prof_<tag>) into profiles/<round>/<name>_*.
usage: tools/pmc_summary.py gpurun_out/prof_<tag> profiles/r01/<name> "<command note>" [kernel substring]"""
import collections
import csv
import glob
import json
import shutil
import sys

src, dst, note = sys.argv[1], sys.argv[2], sys.argv[3]
kern = sys.argv[4] if len(sys.argv) > 4 else None
if kern is None:
    # the dominant decode kernel of the trace (the fallback pass launches a second, tiny, instantiation whose name also matches)
    rows = [r for r in csv.DictReader(open(f"{src}/trace/trace_kernel_stats.csv")) if "scl_decode" in r["Name"] or "sc8_decode" in r["Name"]]
    kern = max(rows, key=lambda r: float(r["TotalDurationNs"]))["Name"].split("(")[0]
shutil.copy(f"{src}/trace/trace_kernel_stats.csv", dst + "_kernel_stats.csv")
out = {}
for f in sorted(glob.glob(f"{src}/pmc*/pmc_counter_collection.csv")):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if kern in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        out[k] = {"launches": len(v), "mean_per_launch": sum(v) / len(v)}
dur = None
for r in csv.DictReader(open(f"{src}/trace/trace_kernel_stats.csv")):
    if kern in r["Name"]:
        dur = {"calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"])}
hbm = None
if "FETCH_SIZE" in out and "WRITE_SIZE" in out:
    # KiB units; FETCH_SIZE counts 128-B requests as 64 B on gfx950 -> x2 (MI355X_MICROARCH.md §HBM)
    rd = out["FETCH_SIZE"]["mean_per_launch"] * 1024 * 2
    wr = out["WRITE_SIZE"]["mean_per_launch"] * 1024
    hbm = {"read_bytes_corrected": rd, "write_bytes": wr, "total_bytes": rd + wr}
json.dump({"command": note, "kernel_filter": kern, "kernel_time": dur, "hbm_traffic_per_launch": hbm, "counters": out,
           "notes": "SQ_* cycle counters are quad-cycles; FETCH_SIZE/WRITE_SIZE in KiB; FETCH_SIZE x2 correction "
                    "for wide coalesced reads on gfx950 (MI355X_MICROARCH.md §HBM)"},
          open(dst + "_pmc.json", "w"), indent=1)
print(open(dst + "_pmc.json").read()[:1500])
