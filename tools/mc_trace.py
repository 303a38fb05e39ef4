#!/usr/bin/env python3
"""tools/mc_trace.py — a short pipelined get_bler_quick sweep (configuration 4's grid, 6 rounds of 262144 trials) under
`rocprofv3 --kernel-trace`: per step, what the device did and how long it sat idle between kernels.

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/mc_trace -- python tools/mc_trace.py
    python tools/mc_trace.py --report gpurun_out/mc_trace
"""
import csv, glob, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
if len(sys.argv) > 1 and sys.argv[1] == "--report":
    f = sorted(glob.glob(os.path.join(sys.argv[2], "**", "*kernel_trace.csv"), recursive=True))[-1]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    big = [i for i, r in enumerate(rows) if "scl_decode_llr_kernel<32" in r["Kernel_Name"] and int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) > 50e6]
    big = big[-5:]                                    # the last five full steps
    for a, b in zip(big[:-1], big[1:]):
        t0 = int(rows[a]["End_Timestamp"])
        print(f"--- between two big decode kernels: {(int(rows[b]['Start_Timestamp']) - t0) / 1e6:.3f} ms; the big kernel itself {(int(rows[b]['End_Timestamp']) - int(rows[b]['Start_Timestamp'])) / 1e6:.2f} ms")
        busy = 0
        for r in rows[a + 1:b]:
            s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
            busy += e - s
            if e - s > 30000:
                print(f"   {s / 1e6:8.3f} .. {e / 1e6:8.3f} ms ({(e - s) / 1e6:6.3f})  {r['Kernel_Name'][:80]}")
        print(f"   kernels busy {busy / 1e6:.3f} ms of the gap")
    sys.exit(0)
import ctypes as C
import numpy as np, torch, polar_amd
C.CDLL(None).srand(C.c_uint(1))
g = polar_amd.PolarCode(11, 1024, 0.32, 16)
grid = [1.0, 1.25, 1.5, 1.75, 2.0]
g.get_bler_quick(grid, [32], max_runs=2 * 262144, max_err=10**12, seed=5, batch=262144)
b, c = g.get_bler_quick(grid, [32], max_runs=8 * 262144, max_err=10**12, seed=5, batch=262144, return_counters=True)
print(b, c["rounds"], {k: g.debug_get("round_us_" + k) for k in ("first", "min", "median", "max", "count")})
