#!/usr/bin/env python3
"""BASELINE configuration 4 on ONE GPU, as a record: N=2048 K=1024 CRC16 L=32, Eb/N0 1:0.25:2 dB, through
PolarCode::get_bler_quick's replacement (polar_get_bler_quick_multi_ex; PolarCode.cpp:658-785) — not per-point decode calls.
(a) the whole grid in ONE call with the reference's early stop (max_err block errors per point, ascending Eb/N0 with
"decoded at a lower Eb/N0 => counted, not simulated"): every point ends with > max_err block errors or max_runs trials;
(b) every point alone, for its own trials/s. Anchors: BASELINE.md §1 (0.2 / 8e-3 / 3e-5 at 1 / 1.5 / 2 dB).
usage: tools/config4_record.py [max_err] [max_runs]   -> JSON on stdout"""
import ctypes as C, hashlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import polar_amd
max_err = int(sys.argv[1]) if len(sys.argv) > 1 else 100
max_runs = int(sys.argv[2]) if len(sys.argv) > 2 else 8 * 1048576
GRID = [1.0, 1.25, 1.5, 1.75, 2.0]
C.CDLL(None).srand(1)
g = polar_amd.PolarCode(11, 1024, 0.32, 16)
g.get_bler_quick(GRID, [32], max_runs=262144, max_err=10**9, seed=5, batch=262144)      # warm-up (allocations)
torch.cuda.synchronize()


def wilson(k, n, z=1.96):
    if n == 0:
        return [0.0, 1.0]
    p = k / n
    d = 1 + z * z / n
    c = p + z * z / (2 * n)
    h = z * np.sqrt(p * (1 - p) / n + z * z / (4 * n * n))
    return [float((c - h) / d), float((c + h) / d)]


t0 = time.perf_counter()
bler, cnt = g.get_bler_quick(GRID, [32], max_runs=max_runs, max_err=max_err, seed=2026, return_counters=True)
dt = time.perf_counter() - t0
rec = {"workload": "N=2048 K=1024 crc16 L=32 LLR-SCL, BPSK/AWGN, Eb/N0 1:0.25:2 dB (BASELINE config 4, one GPU)",
       "lib_sha256": hashlib.sha256(open(polar_amd.LIB_PATH, "rb").read()).hexdigest(),
       "max_err": max_err, "max_runs": max_runs,
       "sweep_one_call": {"seconds": dt, "rounds": cnt["rounds"], "automatic_rounds": "geometric, up to 262144 trials per round",
                          "points": [{"ebno_db": e, "block_errors": int(cnt["err"][0, i]), "runs": int(cnt["run"][0, i]), "bler": float(bler[0, i]),
                                      "wilson95": wilson(int(cnt["err"][0, i]), int(cnt["run"][0, i]))} for i, e in enumerate(GRID)],
                          "trials_per_s_of_the_longest_point": float(cnt["run"].max()) / dt}}
pts = []
for e in GRID:
    t0 = time.perf_counter()
    b, c = g.get_bler_quick([e], [32], max_runs=max_runs, max_err=max_err, seed=2026, return_counters=True)
    dt = time.perf_counter() - t0
    k, n = int(c["err"][0, 0]), int(c["run"][0, 0])
    pts.append({"ebno_db": e, "block_errors": k, "runs": n, "bler": float(b[0, 0]), "wilson95": wilson(k, n), "seconds": dt,
                "rounds": c["rounds"], "trials_per_s": n / dt})
rec["points_alone"] = pts
print(json.dumps(rec, indent=1))
