#!/usr/bin/env python3
"""tools/small_batch_geometry.py — small batches at list sizes 16 / 32 (no one-codeword-per-wave form: the state does not fit the LDS):
call time through the host ABI for the default geometry (16 waves per CU, layers <= 8 in LDS) against fewer, fatter waves
(polar_set_tuning: 8 or 4 waves per CU, layers <= 16 / 32 in LDS). A lone wave pays a memory round trip per dependent access of an
HBM-resident layer; the more layers its LDS holds, the fewer it makes. usage: tools/small_batch_geometry.py [out.json]"""
import ctypes as C, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import polar_amd, oracle_lib
o = oracle_lib.Oracle(11, 1024, 0.32, 16, srand=1)
C.CDLL(None).srand(C.c_uint(1))
g = polar_amd.PolarCode(11, 1024, 0.32, 16)
llr, _ = o.synth_llr(99, 0, 4096, o.snr_sqrt_linear(2.0))
rows = []
for L in (16, 32):
    ref = {}
    for wpc, ll in ((0, 0), (8, 4), (8, 5), (4, 5), (16, 4)):
        try:
            g.set_tuning(wpc, ll)
        except Exception as e:
            print("tuning", wpc, ll, "refused:", e); continue
        for B in (1, 8, 64, 256, 1024, 4096):
            x = np.ascontiguousarray(llr[:B])
            try:
                got = g.decode_scl_llr(x, L)
            except Exception as e:
                print("L", L, "tuning", wpc, ll, "B", B, "failed:", e); break
            if (wpc, ll) == (0, 0): ref[B] = got
            same = bool((got == ref[B]).all())
            ts = []
            for _ in range(12):
                t = time.perf_counter(); g.decode_scl_llr(x, L); ts.append(time.perf_counter() - t)
            rows.append(dict(L=L, waves_per_cu=wpc, lds_log=ll, B=B, call_ms=float(np.median(ts) * 1e3), same_bits=same))
            print(f"L={L} tuning ({wpc},{ll}) B={B}: {np.median(ts) * 1e3:.3f} ms same_bits={same}", flush=True)
g.set_tuning(0, 0)
if len(sys.argv) > 1: json.dump(rows, open(sys.argv[1], "w"), indent=1)
