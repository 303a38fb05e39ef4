#!/usr/bin/env python3
"""GPU box: random codes / list sizes / SNRs / design parameters against the oracle (C restatement of the reference).
usage: tools/fuzz_parity.py [configs] [seed]   (FUZZ_SANE=1: rates <= 0.6 and design parameters 0.32 .. 0.5 only).
The configuration generator is tests/fuzz_util.py; a seeded slice of it runs in the -m gpu suite (tests/test_gpu_fuzz.py)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import fuzz_util

n_cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
sane = bool(os.environ.get("FUZZ_SANE"))
bad_total = 0; deg_total = 0; cw_total = 0
t_start = time.time()
for it in range(n_cfg):
    cfg = fuzz_util.draw(rng, sane)
    B, bad, bad_deg, deg, rows = fuzz_util.run_one(cfg, it, rng, 1.5)
    bad_total += bad; deg_total += bad_deg; cw_total += B
    if bad or bad_deg:
        print(f"    rows {rows[:8]} (rows 0..2 are the degenerate ones when present: {deg})")
    if bad or bad_deg or it % 10 == 0:
        print(f"[{it}] n={cfg['n']} K={cfg['K']} crc={cfg['crc']} L={cfg['L']} eps={cfg['eps']} EbN0={cfg['ebno']:.2f} B={B}: "
              f"mismatching codewords {bad} (+ {bad_deg} degenerate rows)", flush=True)
print(f"fuzz: {n_cfg} configurations, {cw_total} codewords, TOTAL MISMATCHES {bad_total} ordinary + {deg_total} degenerate rows  ({time.time() - t_start:.0f} s)")
