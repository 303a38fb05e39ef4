#!/usr/bin/env python3
"""Round-3 BLER records on the GPU engine, from the library in the tree (sha256 recorded), with 95 % Wilson intervals, next to
the anchors read off the reference's figures:
  (a) results/polar_performance.jpeg: N=2048 K=1024, L in {1, 2, 4, 8, 32} without CRC and L=32 + CRC16, Eb/N0 1 .. 3 dB
  (b) BASELINE config 3: L=4 + CRC16, Eb/N0 1:0.25:2.5
  (c) BASELINE config 5 / results/monte_carlo.png: (1024, 512) on the reference's shipped Monte-Carlo construction table,
      16-ASK Gray BICM, SNR = 13 + (-3:0.25:3) dB (PolarM/main_MC_CC_Comparison.m:15,28,42,121), L = 1 (SC, the figure) and L = 8
Each (L, point) is simulated independently (no 'decoded at a lower Eb/N0' shortcut) until `min_err` block errors or `max_trials`.
usage: tools/bler_sweeps.py out.json [quick]"""
import ctypes as C
import hashlib
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import polar_amd

out_path = sys.argv[1] if len(sys.argv) > 1 else "bler_sweeps.json"
quick = len(sys.argv) > 2
MIN_ERR = 100 if quick else 300


def wilson(k, n, z=1.96):
    if n == 0:
        return (0.0, 1.0)
    p = k / n
    d = 1 + z * z / n
    c = (p + z * z / (2 * n)) / d
    h = z * math.sqrt(p * (1 - p) / n + z * z / (4 * n * n)) / d
    return (max(0.0, c - h), min(1.0, c + h))


def point(g, L, x, max_trials, bicm=None):
    err = np.zeros((1, 1), np.uint64); run = np.zeros((1, 1), np.uint64)
    t0 = time.time(); base = 0
    T = 262144 if L <= 8 else 131072
    while err[0, 0] < MIN_ERR and base < max_trials:
        if bicm:
            g.mc_batch_bicm(bicm, 4242, base, T, 1, [x], [L], np.ones((1, 1), np.uint8), err, run)
        else:
            g.mc_batch(4242, base, T, 1, [x], [L], np.ones((1, 1), np.uint8), err, run)
        base += T
    k, n = int(err[0, 0]), int(run[0, 0])
    lo, hi = wilson(k, n)
    return {"errors": k, "trials": n, "bler": k / n, "ci95": [lo, hi], "seconds": round(time.time() - t0, 2)}


res = {"lib_sha256": hashlib.sha256(open(os.path.join(ROOT, "polar_amd", "libpolar_amd.so"), "rb").read()).hexdigest(),
       "note": "GPU Monte-Carlo (polar_mc_batch / polar_mc_batch_bicm), seed 4242, independent points, stop at %d block errors" % MIN_ERR,
       "anchors_polar_performance_jpeg": {"L=1": {"1.0": 0.7, "1.5": 0.3, "2.0": 0.05, "2.5": 3e-3, "3.0": 2e-4},
                                          "L=32": {"1.0": 0.12, "1.5": 8e-3, "2.0": 1.5e-3, "2.5": 3e-4, "3.0": 3e-5},
                                          "L=32+CRC16": {"1.0": 0.2, "1.5": 8e-3, "2.0": 3e-5}},
       "anchors_monte_carlo_png_16ask_bicm_sc": {"9.25": 0.2, "10.5": 1e-2, "11.25": 1e-3},
       "reference_main_cpp_table_1000_runs": {"Eb/N0": [1.0, 1.25, 1.5, 1.75, 2.0],
                                              "L=1": [0.711268, 0.505, 0.300595, 0.145954, 0.052], "L=2": [0.474178, 0.271505, 0.091, 0.033, 0.007],
                                              "L=4": [0.306991, 0.128827, 0.031, 0.013, 0.002], "L=8": [0.216738, 0.068, 0.017, 0.007, 0.002],
                                              "L=32": [0.120669, 0.023, 0.009, 0.005, 0.002]},
       "curves": []}
scale = 0.25 if quick else 1.0
grid_a = [1.0, 1.25, 1.5, 1.75, 2.0, 2.25, 2.5, 2.75, 3.0]
jobs = [("polar_performance.jpeg", 0, L, grid_a, int((4e6 if L < 32 else 6e6) * scale)) for L in (1, 2, 4, 8, 32)]
jobs.append(("polar_performance.jpeg", 16, 32, [1.0, 1.25, 1.5, 1.75, 2.0], int(12e6 * scale)))
jobs.append(("config 3", 16, 4, [1.0, 1.25, 1.5, 1.75, 2.0, 2.25, 2.5], int(8e6 * scale)))
if os.environ.get("BLER_ONLY") == "cfg5":
    jobs = []
for (what, crc, L, xs, max_trials) in jobs:
    C.CDLL(None).srand(1)
    g = polar_amd.PolarCode(11, 1024, 0.32, crc)
    cur = {"figure": what, "N": 2048, "K": 1024, "crc": crc, "L": L, "constellation": "bpsk", "axis": "Eb/N0 dB", "points": []}
    for x in xs:
        r = point(g, L, x, max_trials); r["x"] = x
        cur["points"].append(r); print(what, crc, L, r, flush=True)
    res["curves"].append(cur)
    g.close()
gold = np.load(os.path.join(ROOT, "tests", "golden", "polar_golden.npz"))
for L in (1, 8):
    g = polar_amd.PolarCode.from_counts(gold["cfg5_n10_k512_ask16/counts"], 512)
    cur = {"figure": "config 5 / monte_carlo.png", "N": 1024, "K": 512, "crc": 0, "L": L, "constellation": "ask16-gray BICM",
           "axis": "SNR dB (Eb/N0 = SNR - 3.01 dB)", "code": "reference's MC table ..._13_ask16-gray_bicm_250000.txt", "points": []}
    for x in [13 + 0.25 * i for i in range(-12, 13)]:
        r = point(g, L, x, int(4e6 * scale), bicm="ask16-gray"); r["x"] = x
        cur["points"].append(r); print("cfg5", L, r, flush=True)
        if r["errors"] == 0:
            break
    res["curves"].append(cur)
    g.close()
json.dump(res, open(out_path, "w"), indent=1)
