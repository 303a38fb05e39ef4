#!/usr/bin/env python3
"""BLER curve on the GPU engine for comparison with the reference's published figure
(results/polar_performance.jpeg; anchor points in BASELINE.md §1). Each (L, Eb/N0) point is simulated
independently (no 'decoded at a lower Eb/N0' shortcut) until `min_err` block errors or `max_trials`.
usage: tools/bler_curve.py out.json"""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import polar_amd

out_path = sys.argv[1] if len(sys.argv) > 1 else "bler_curve.json"
res = []
for (crc, L, ebnos, max_trials) in [(0, 1, [1.0, 1.5, 2.0, 2.5, 3.0], 2_000_000), (0, 32, [1.0, 1.5, 2.0], 1_000_000),
                                    (16, 32, [1.0, 1.25, 1.5, 1.75, 2.0], 6_000_000)]:
    C.CDLL(None).srand(1)
    g = polar_amd.PolarCode(11, 1024, 0.32, crc)
    for e in ebnos:
        err = np.zeros((1, 1), np.uint64); run = np.zeros((1, 1), np.uint64)
        t0 = time.time(); base = 0
        T = 131072 if L == 1 else 32768
        while err[0, 0] < 200 and base < max_trials:
            g.mc_batch(4242, base, T, 1, [e], [L], np.ones((1, 1), np.uint8), err, run)
            base += T
        r = {"N": 2048, "K": 1024, "crc": crc, "L": L, "ebno_db": e, "errors": int(err[0, 0]), "trials": int(run[0, 0]),
             "bler": float(err[0, 0]) / float(run[0, 0]), "seconds": round(time.time() - t0, 2)}
        res.append(r); print(r, flush=True)
json.dump({"note": "GPU Monte-Carlo (polar_mc_batch), seed 4242; compare BASELINE.md §1 anchors read off results/polar_performance.jpeg",
           "points": res}, open(out_path, "w"), indent=1)
