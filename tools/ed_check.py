#!/usr/bin/env python3
"""GPU box: exp-domain kernel (mode 2) vs LLR-domain kernel (mode 1) — bit parity on large batches and timing.
usage: tools/ed_check.py [B_parity] [B_time]"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import polar_amd

Bp = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
Bt = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
libc = C.CDLL(None)
CODES = [(11, 1024, 16, 32), (11, 1024, 16, 8), (10, 512, 0, 8), (9, 256, 8, 32), (11, 1024, 0, 16), (11, 1536, 11, 64)]
if os.environ.get("ED_ONLY32"):
    CODES = CODES[:1]
if os.environ.get("ED_SMALL"):       # the 4-lane groups (lists of 3 and 4: exp-domain kernel in automatic mode since round 3)
    CODES = [(11, 1024, 16, 4), (11, 1024, 0, 3), (10, 512, 0, 4), (9, 256, 8, 3), (12, 2048, 16, 4), (8, 100, 4, 4), (11, 1400, 0, 4)]
bad_total = 0
for (n, K, crc, L) in CODES:
    libc.srand(1)
    g = polar_amd.PolarCode(n, K, 0.32, crc)
    N = 1 << n
    d_llr = torch.empty((Bp, N), dtype=torch.float64, device="cuda")
    o1 = torch.empty((Bp, K), dtype=torch.uint8, device="cuda")
    o2 = torch.empty((Bp, K), dtype=torch.uint8, device="cuda")
    for ebno in (0.5, 1.5, 2.0, 3.0, 6.0):
        g.synth_llr_dev(4242, 0, Bp, g.snr_sqrt_linear(ebno), d_llr.data_ptr())
        g.set_mode(1)
        g.decode_scl_llr_dev(d_llr.data_ptr(), Bp, L, o1.data_ptr())
        g.set_mode(2)
        g.decode_scl_llr_dev(d_llr.data_ptr(), Bp, L, o2.data_ptr())
        torch.cuda.synchronize()
        bad = int((o1 != o2).any(dim=1).sum())
        bad_total += bad
        print(f"n={n} K={K} crc={crc} L={L} EbN0={ebno} B={Bp}: ED vs LLR mismatching codewords {bad}", flush=True)
print("TOTAL MISMATCHES", bad_total)

if Bt <= 0:
    sys.exit(0 if bad_total == 0 else 1)
# timing, headline config
libc.srand(1)
g = polar_amd.PolarCode(11, 1024, 0.32, 16)
d_llr = torch.empty((Bt, 2048), dtype=torch.float64, device="cuda")
o1 = torch.empty((Bt, 1024), dtype=torch.uint8, device="cuda")
g.synth_llr_dev(4242, 0, Bt, g.snr_sqrt_linear(2.0), d_llr.data_ptr())
for mode in ("1", "2", "1", "2"):
    g.set_mode(int(mode))
    g.decode_scl_llr_dev(d_llr.data_ptr(), Bt, 32, o1.data_ptr())
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(3):
        g.decode_scl_llr_dev(d_llr.data_ptr(), Bt, 32, o1.data_ptr())
    torch.cuda.synchronize()
    dt = (time.time() - t0) / 3
    print(f"mode {mode}: {dt*1e3:.2f} ms per {Bt} codewords = {Bt/dt:.0f} cw/s", flush=True)
