#!/usr/bin/env python3
"""One-off stress parity on the GPU box: large batches at low SNR against the unmodified reference
build (oracle/_ref) running on host threads. usage: tools/stress_parity.py [scale]"""
import ctypes as C
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import polar_amd
import oracle_lib

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
libc = C.CDLL(None)
total_bad = 0
SETS = {
    "1": [(11, 1024, 16, 32, 1.0, 6144), (11, 1024, 16, 32, 2.0, 6144), (11, 1024, 16, 4, 1.0, 16384),
          (11, 1024, 0, 1, 1.5, 65536), (10, 512, 0, 8, 1.0, 16384), (9, 256, 8, 32, 0.5, 8192)],
    # other rates / sizes / list sizes, high and very low SNR
    "2": [(11, 1024, 16, 32, 3.0, 4096), (11, 1024, 16, 32, -1.0, 3072), (12, 2048, 16, 8, 1.5, 4096), (10, 768, 8, 16, 3.0, 8192),
          (8, 64, 0, 32, 0.0, 16384), (11, 512, 24, 5, 0.5, 8192), (13, 4096, 0, 2, 2.0, 2048), (11, 1536, 11, 64, 3.5, 2048)],
}
for (n, K, crc, L, ebno, B) in SETS[os.environ.get("STRESS_SET", "1")]:
    B = int(B * scale)
    libc.srand(1)
    g = polar_amd.PolarCode(n, K, 0.32, crc)
    N = 1 << n
    d_llr = torch.empty((B, N), dtype=torch.float64, device="cuda")
    d_out = torch.empty((B, K), dtype=torch.uint8, device="cuda")
    g.synth_llr_dev(777, 0, B, g.snr_sqrt_linear(ebno), d_llr.data_ptr())
    g.decode_scl_llr_dev(d_llr.data_ptr(), B, L, d_out.data_ptr())
    torch.cuda.synchronize()
    llr = d_llr.cpu().numpy(); got = d_out.cpu().numpy()
    T = 32
    want = np.zeros_like(got)
    def work(t):
        libc.srand(1)   # (each thread builds its own reference object; CRC matrix pinned below)
        r = oracle_lib.Reference(n, K, 0.32, crc)
        r.set_crc_matrix(g.crc_matrix)
        sl = slice(t * B // T, (t + 1) * B // T)
        want[sl] = r.decode_scl_llr(llr[sl], L)
    t0 = time.time()
    th = [threading.Thread(target=work, args=(t,)) for t in range(T)]
    [x.start() for x in th]; [x.join() for x in th]
    bad = int((want != got).any(axis=1).sum())
    total_bad += bad
    print(f"n={n} K={K} crc={crc} L={L} EbN0={ebno} B={B}: mismatching codewords {bad}  (reference on {T} threads: {time.time()-t0:.1f}s)", flush=True)
print("TOTAL MISMATCHES", total_bad)
