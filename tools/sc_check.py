#!/usr/bin/env python3
"""GPU box: list-size-1 kernel (pruned SC, polar_kernels_sc.hip) vs the general LLR-domain kernel — parity and timing."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import polar_amd

libc = C.CDLL(None)
bad_total = 0
for (n, K, crc, B) in [(11, 1024, 0, 65536), (11, 1024, 16, 16384), (9, 256, 0, 65536), (5, 16, 4, 4096), (10, 300, 8, 8192), (12, 3000, 0, 4096), (7, 100, 0, 8192)]:
    libc.srand(1)
    g = polar_amd.PolarCode(n, K, 0.32, crc)
    N = 1 << n
    d_llr = torch.empty((B, N), dtype=torch.float64, device="cuda")
    o1 = torch.empty((B, K), dtype=torch.uint8, device="cuda")
    o2 = torch.empty((B, K), dtype=torch.uint8, device="cuda")
    for ebno in (0.0, 2.0, 5.0):
        g.synth_llr_dev(99, 0, B, g.snr_sqrt_linear(ebno), d_llr.data_ptr())
        g.set_mode(1); g.decode_scl_llr_dev(d_llr.data_ptr(), B, 1, o1.data_ptr())
        g.set_mode(0); g.decode_scl_llr_dev(d_llr.data_ptr(), B, 1, o2.data_ptr())
        torch.cuda.synchronize()
        bad = int((o1 != o2).any(dim=1).sum())
        bad_total += bad
        print(f"n={n} K={K} crc={crc} EbN0={ebno} B={B}: SC kernel vs general kernel mismatching codewords {bad}", flush=True)
print("TOTAL MISMATCHES", bad_total)
libc.srand(1)
g = polar_amd.PolarCode(11, 1024, 0.32, 0)
for B in (65536, 65536 * 4):
  d_llr = torch.empty((B, 2048), dtype=torch.float64, device="cuda")
  o1 = torch.empty((B, 1024), dtype=torch.uint8, device="cuda")
  g.synth_llr_dev(4242, 0, B, g.snr_sqrt_linear(2.0), d_llr.data_ptr())
  for mode in (1, 0, 1, 0):
    g.set_mode(mode)
    g.decode_scl_llr_dev(d_llr.data_ptr(), B, 1, o1.data_ptr()); torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(3):
        g.decode_scl_llr_dev(d_llr.data_ptr(), B, 1, o1.data_ptr())
    torch.cuda.synchronize()
    dt = (time.time() - t0) / 3
    print(f"mode {mode}: {dt*1e3:.2f} ms per {B} codewords = {B/dt/1e6:.2f} M cw/s", flush=True)
