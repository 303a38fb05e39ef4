#!/usr/bin/env python3
"""GPU box: end-to-end Monte-Carlo trial rate of get_bler_quick (synth + decode + count, device-side rounds) next to the
decode-only rate of bench.py. usage: tools/mc_rate.py [L] [runs]"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, polar_amd
L = int(sys.argv[1]) if len(sys.argv) > 1 else 32
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 1048576
C.CDLL(None).srand(1)
g = polar_amd.PolarCode(11, 1024, 0.32, 16 if L > 1 else 0)
g.get_bler_quick([2.0], [L], max_runs=262144, max_err=10**9, seed=3, batch=262144)        # warm-up (allocations)
for ebno in ([2.0], [1.0, 1.5, 2.0, 2.5]):
    t0 = time.time()
    bler = g.get_bler_quick(ebno, [L], max_runs=runs, max_err=10**9, seed=4, batch=262144)
    dt = time.time() - t0
    print(f"L={L} Eb/N0 {ebno}: {runs} runs per point in {dt:.3f} s = {runs / dt / 1e6:.3f} M trials/s per sweep ({runs * len(ebno) / dt / 1e6:.3f} M decodes/s if every point were simulated), bler {np.asarray(bler).ravel()}")
