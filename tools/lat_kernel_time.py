#!/usr/bin/env python3
"""B = 1 ... 64 decodes of the headline code at list size 1 (and 4, 32) through the host-pointer ABI: wall time per call; run under
`rocprofv3 --kernel-trace --stats` for the kernels' own durations. usage: tools/lat_kernel_time.py [L ...]"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import polar_amd, oracle_lib
Ls = [int(x) for x in sys.argv[1:]] or [1]
o = oracle_lib.Oracle(11, 1024, 0.32, 16, srand=1)
C.CDLL(None).srand(C.c_uint(1))
g = polar_amd.PolarCode(11, 1024, 0.32, 16)
llr, _ = o.synth_llr(99, 0, 64, o.snr_sqrt_linear(2.0))
for L in Ls:
    for B in (1, 8, 64):
        x = np.ascontiguousarray(llr[:B])
        g.decode_scl_llr(x, L)
        ts = []
        for _ in range(200 if L == 1 else 30):
            t = time.perf_counter(); g.decode_scl_llr(x, L); ts.append(time.perf_counter() - t)
        print(f"L={L} B={B}: median {np.median(ts) * 1e3:.3f} ms, min {min(ts) * 1e3:.3f} ms per call", flush=True)
