#!/usr/bin/env python3
"""GPU box: random codes / list sizes / SNRs / design parameters against the oracle (C restatement of the reference).
usage: tools/fuzz_parity.py [configs] [seed]   (FUZZ_SANE=1: rates <= 0.6 and design parameters 0.32 .. 0.5 only)"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import polar_amd
from oracle_lib import Oracle

n_cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad_total = 0; cw_total = 0
t_start = time.time()
for it in range(n_cfg):
    n = int(rng.integers(3, 13)); N = 1 << n
    crc = int(rng.choice([0, 0, 4, 8, 11, 16, 24]))
    if crc >= N - 1: crc = 0
    K = int(rng.integers(1, N - crc + 1))
    if os.environ.get("FUZZ_SANE"):                              # codes a construction would produce for its channel
        K = int(rng.integers(1, max(2, int(0.6 * N) - crc)))
    L = int(rng.choice([1, 1, 2, 3, 4, 5, 8, 12, 16, 17, 24, 31, 32, 32, 33, 64]))
    eps = float(rng.choice([0.1, 0.32, 0.32, 0.5, 0.7]))
    if os.environ.get("FUZZ_SANE"):
        eps = float(rng.choice([0.32, 0.32, 0.4, 0.5]))
    ebno = float(rng.uniform(-1.0, 4.5))
    rate = 131.0 * (2048 * 11 * 32) / (N * n * L)                 # oracle codewords/s, single thread (rough)
    B = int(min(4096, max(16, 1.5 * rate)))
    o = Oracle(n, K, eps, crc, srand=it + 1)
    C.CDLL(None).srand(C.c_uint(it + 1))
    g = polar_amd.PolarCode(n, K, eps, crc)
    llr, _ = o.synth_llr(1000 + it, 0, B, o.snr_sqrt_linear(ebno))
    deg = rng.random() < 0.15
    if deg:                                                        # a few degenerate rows
        llr[0] = 0.0
        llr[1] = np.where(np.arange(N) % 2 == 0, 1e3, -1e3)
        llr[2] *= 1e-3
    want = o.decode_scl_llr(llr, L)
    got = g.decode_scl_llr(llr, L)
    bad = int((want != got).any(axis=1).sum())
    bad_total += bad; cw_total += B
    if bad:
        rows = np.nonzero((want != got).any(axis=1))[0]
        print(f"    rows {rows[:8]} (rows 0..2 are the degenerate ones when present: {bool(deg)})")
    if bad or it % 10 == 0:
        print(f"[{it}] n={n} K={K} crc={crc} L={L} eps={eps} EbN0={ebno:.2f} B={B}: mismatching codewords {bad}", flush=True)
print(f"fuzz: {n_cfg} configurations, {cw_total} codewords, TOTAL MISMATCHES {bad_total}  ({time.time() - t_start:.0f} s)")
