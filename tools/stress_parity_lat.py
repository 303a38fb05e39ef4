#!/usr/bin/env python3
"""tools/stress_parity_lat.py — the one-codeword-per-wave kernels (list sizes 2 ... 8 and 1; decode_sc_p1's is covered by its own
tests) against the unmodified reference build (oracle/_ref) on host threads: the headline code and three others, three Eb/N0 each,
decoded in calls of 1 ... 512 codewords — every call below the dispatch threshold, i.e. through the LAT form — device-resident.
usage: tools/stress_parity_lat.py [codewords per point = 8192]"""
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

per_point = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
libc = C.CDLL(None)
total = total_bad = 0
CODES = [(11, 1024, 16), (10, 512, 0), (9, 300, 8), (7, 40, 0)]
for (n, K, crc) in CODES:
    for L in (1, 2, 3, 4, 5, 7, 8):
        for ebno in (0.5, 2.0, 3.5):
            libc.srand(1)
            g = polar_amd.PolarCode(n, K, 0.32, crc)
            N = 1 << n
            B = per_point if n >= 10 else 2 * per_point
            d_llr = torch.empty((B, N), dtype=torch.float64, device="cuda")
            d_out = torch.empty((B, K), dtype=torch.uint8, device="cuda")
            g.synth_llr_dev(4242 + L, 0, B, g.snr_sqrt_linear(ebno), d_llr.data_ptr())
            # calls of 1, 2, 3, 17, 64, 256, 512, ... codewords (all through the one-codeword-per-wave form: "lat_max_b" forced)
            g.debug_set("lat_max_b", 1 << 40)
            sizes = [1, 2, 3, 17, 64, 256, 512]
            off = k = 0
            while off < B:
                c = min(sizes[k % len(sizes)], B - off)
                g.decode_scl_llr_dev(d_llr.data_ptr() + off * N * 8, c, L, d_out.data_ptr() + off * K)
                off += c; k += 1
            torch.cuda.synchronize()
            g.debug_set("lat_max_b", 0)
            llr = d_llr.cpu().numpy(); got = d_out.cpu().numpy()
            T = 32
            want = np.zeros_like(got)
            def work(t):
                libc.srand(1)
                r = oracle_lib.Reference(n, K, 0.32, crc)
                r.set_crc_matrix(g.crc_matrix)
                sl = slice(t * B // T, (t + 1) * B // T)
                want[sl] = r.decode_scl_llr(llr[sl], L)
            t0 = time.time()
            th = [threading.Thread(target=work, args=(t,)) for t in range(T)]
            [x.start() for x in th]; [x.join() for x in th]
            bad = int((want != got).any(axis=1).sum())
            total += B; total_bad += bad
            print(f"n={n} K={K} crc={crc} L={L} EbN0={ebno} B={B}: mismatching codewords {bad}  (reference: {time.time()-t0:.1f}s)", flush=True)
            g.close()
print(f"LAT stress: {total} codewords, TOTAL MISMATCHES {total_bad}")
