#!/usr/bin/env python3
"""Small-batch latency at the boundary (round-2 verdict item 6): the reference's loops call the decoder one codeword at a time
(PolarCode.cpp:756, PolarM/main_MC_CC_Comparison.m:96). Host-pointer ABI (polar_decode_scl_llr_batch: H2D copy + kernels + D2H
copy + synchronisation) for B in {1, 8, 64, 512, 4096} and L in {1, 4, 32} on the headline code; median of `reps` calls; next to
the CPU side (unmodified reference when present, else the C restatement) on one core.
usage: tools/latency_table.py out.json"""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import polar_amd
import oracle_lib

out_path = sys.argv[1] if len(sys.argv) > 1 else "latency_table.json"
n, K, crc = 11, 1024, 16
kind = "reference" if oracle_lib.have_reference() else "port"
cpu = (oracle_lib.Reference if kind == "reference" else oracle_lib.Oracle)(n, K, 0.32, crc, srand=1)
o = oracle_lib.Oracle(n, K, 0.32, crc, srand=1)
C.CDLL(None).srand(C.c_uint(1))
g = polar_amd.PolarCode(n, K, 0.32, crc)
llr_all, _ = o.synth_llr(99, 0, 4096, o.snr_sqrt_linear(2.0))
rows = []
for L in [int(x) for x in os.environ.get("LAT_LS", "1,4,32").split(",")]:
    t = time.perf_counter()
    nc = 64 if L == 32 else 512
    want = cpu.decode_scl_llr(llr_all[:nc], L)
    cpu_per = (time.perf_counter() - t) / nc
    cross = None
    for B in (1, 8, 64, 128, 256, 512, 1024, 2048, 4096):
        llr = np.ascontiguousarray(llr_all[:B])
        # list sizes 1 ... 8 have two kernels (round 4): one codeword per wave with the state in LDS (small batches) and the batch
        # kernels; "auto" is what a caller gets, the other two rows force one of them (polar_debug_set "lat_max_b") to show the
        # crossover (list sizes 2 ... 8: the forced latency kernel still needs its state to fit the LDS, else the batch kernel runs)
        for variant, knob in ((("auto", 0), ("batch kernel", -1), ("one codeword per wave", 1 << 40)) if L <= 8 else (("auto", 0),)):
            g.debug_set("lat_max_b", knob)
            got = g.decode_scl_llr(llr, L)                       # warm-up (allocations) + check
            m = min(B, nc)
            assert (got[:m] == want[:m]).all(), (L, B)
            reps = 30 if B <= 64 else 8
            ts = []
            for _ in range(reps):
                t = time.perf_counter()
                g.decode_scl_llr(llr, L)
                ts.append(time.perf_counter() - t)
            med = float(np.median(ts))
            r = {"L": L, "B": B, "kernel": variant, "gpu_call_ms": med * 1e3, "gpu_ms_per_codeword": med * 1e3 / B, "gpu_codewords_per_s": B / med,
                 "cpu_ms_per_codeword_one_core": cpu_per * 1e3, "cpu_kind": kind, "gpu_faster_than_one_core": bool(med / B < cpu_per)}
            if variant == "auto" and cross is None and med / B < cpu_per:
                cross = B
            rows.append(r); print(r, flush=True)
        g.debug_set("lat_max_b", 0)
    rows.append({"L": L, "crossover_batch_vs_one_cpu_core": cross})
json.dump({"code": "N=2048 K=1024 crc16, Eb/N0 = 2 dB", "abi": "polar_decode_scl_llr_batch (host pointers: H2D + decode + D2H, synchronous)",
           "rows": rows}, open(out_path, "w"), indent=1)
