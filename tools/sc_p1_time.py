#!/usr/bin/env python3
"""tools/sc_p1_time.py — PolarM's decode_sc_p1 through the host entry point, B = 1 / 8 / 64 (the one-codeword-per-wave kernel) at the
headline block length: wall time per call, first rows checked against the C restatement."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import polar_amd, oracle_lib
o = oracle_lib.Oracle(11, 1024, 0.32, 0, srand=1)
C.CDLL(None).srand(C.c_uint(1))
g = polar_amd.PolarCode(11, 1024, 0.32, 0)
llr, _ = o.synth_llr(99, 0, 64, o.snr_sqrt_linear(2.0))
p1 = 1.0 / (1.0 + np.exp(llr))
for B in (1, 8, 64):
    x = np.ascontiguousarray(p1[:B])
    got = g.decode_sc_p1(x)
    ok = all((got[i] == o.decode_sc_p1(x[i])).all() for i in range(min(B, 4)))
    ts = []
    for _ in range(30):
        t = time.perf_counter(); g.decode_sc_p1(x); ts.append(time.perf_counter() - t)
    print(f"decode_sc_p1 B={B}: median {np.median(ts) * 1e3:.3f} ms, min {min(ts) * 1e3:.3f} ms per call, equal to the restatement: {ok}", flush=True)
