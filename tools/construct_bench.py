#!/usr/bin/env python3
"""Monte-Carlo code construction timing (run on the GPU box): python tools/construct_bench.py [n] [constellation] [snr_db] [runs]"""
import sys
import time

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import polar_amd

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
cons = sys.argv[2] if len(sys.argv) > 2 else "ask16-gray"
snr = float(sys.argv[3]) if len(sys.argv) > 3 else 13.0
runs = int(sys.argv[4]) if len(sys.argv) > 4 else 250000
polar_amd.mc_construction(n, snr, 1024, cons)            # warm-up (module load, allocations)
t = time.perf_counter()
c = polar_amd.mc_construction(n, snr, runs, cons)
dt = time.perf_counter() - t
print(f"N={1 << n} {cons} design {snr} dB: {runs} runs in {dt * 1e3:.1f} ms -> {runs / dt:.3e} runs/s; "
      f"counts sum {int(c.sum())}, K=N/2 boundary count {int(np.sort(c)[(1 << n) // 2 - 1])}")
