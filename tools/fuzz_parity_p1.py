#!/usr/bin/env python3
"""GPU box: random (sane) codes / list sizes / SNRs for the probability-domain decoder decode_scl_p1 vs the oracle."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, polar_amd
from oracle_lib import Oracle
rng = np.random.default_rng(5)
bad_total = 0; cw_total = 0
for it in range(60):
    n = int(rng.integers(3, 11)); N = 1 << n
    crc = int(rng.choice([0, 0, 4, 8, 16]))
    if crc >= N - 1: crc = 0
    K = int(rng.integers(1, max(2, int(0.6 * N) - crc)))
    L = int(rng.choice([1, 2, 3, 4, 5, 8, 16, 20, 32, 33, 64]))
    ebno = float(rng.uniform(-1.0, 4.5))
    B = int(min(256, max(8, 40000 * 64 // (N * n * L))))
    o = Oracle(n, K, 0.32, crc, srand=it + 1)
    C.CDLL(None).srand(C.c_uint(it + 1))
    g = polar_amd.PolarCode(n, K, 0.32, crc)
    llr, _ = o.synth_llr(2000 + it, 0, B, o.snr_sqrt_linear(ebno))
    p1 = 1.0 / (1.0 + np.exp(llr)); p0 = 1.0 - p1
    want = np.stack([o.decode_scl_p1(p1[i], p0[i], L) for i in range(B)])
    got = g.decode_scl_p1(p1, p0, L)
    bad = int((want != got).any(axis=1).sum()); bad_total += bad; cw_total += B
    if bad: print(f"[{it}] n={n} K={K} crc={crc} L={L} EbN0={ebno:.2f} B={B}: decode_scl_p1 mismatching codewords {bad}", flush=True)
print(f"p1 fuzz: 60 configurations, {cw_total} codewords, TOTAL MISMATCHES {bad_total}")
