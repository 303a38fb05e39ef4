#!/usr/bin/env python3
"""Geometry sweep per configuration (round-3 verdict item 5): decode throughput of BASELINE configurations 3 and 5 (and the
headline) at their batches over (waves_per_cu, lds_log) — polar_set_tuning — three timed decodes each, best of two passes.
usage: tools/tune_configs.py [config ...]   -> table on stdout"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import bench
import polar_amd
names = sys.argv[1:] or ["config3", "config5", "config3_b262144", "config5_b262144"]
POINTS = [(0, 0), (16, 3), (16, 2), (12, 3), (12, 2), (8, 3), (8, 4), (8, 5), (6, 4), (4, 5), (16, 4), (16, 5)]
dev = torch.device("cuda", 0)
for name in names:
    n, K, crc, L, B, axis, const, kern, cpu_n = bench.OTHER_CONFIGS[name] if name != "headline" else (11, 1024, 16, 32, 262144, 2.0, "bpsk", "", 0)
    if name == "headline":
        C.CDLL(None).srand(1); code = polar_amd.PolarCode(n, K, 0.32, crc)
    else:
        code = bench.make_config(name)
    N = 1 << n
    llr = torch.empty((B, N), dtype=torch.float64, device=dev)
    out = torch.empty((B, K), dtype=torch.uint8, device=dev)
    ref = torch.empty((B, K), dtype=torch.uint8, device=dev)
    if const == "bpsk":
        code.synth_llr_dev(2024, 0, B, code.snr_sqrt_linear(axis), llr.data_ptr())
    else:
        code.synth_bicm_llr_dev(const, 2024, 0, B, axis, llr.data_ptr())
    code.decode_scl_llr_dev(llr.data_ptr(), B, L, ref.data_ptr())
    torch.cuda.synchronize()
    print(f"{name}: N={N} K={K} crc={crc} L={L} batch {B}")
    res = {}
    for rep in range(2):
        for (w, l) in POINTS:
            try:
                code.set_tuning(w, l)
            except polar_amd.PolarError as e:
                res[(w, l)] = None
                continue
            code.decode_scl_llr_dev(llr.data_ptr(), B, L, out.data_ptr())
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                code.decode_scl_llr_dev(llr.data_ptr(), B, L, out.data_ptr())
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 3
            same = bool(torch.equal(out, ref))
            prev = res.get((w, l))
            if prev is None or dt < prev[0]:
                res[(w, l)] = (dt, same)
    base = res[(0, 0)][0]
    for (w, l) in POINTS:
        r = res.get((w, l))
        if r is None:
            print(f"   waves/CU {w:2d} lds_log {l}: refused by polar_set_tuning")
        else:
            print(f"   waves/CU {w:2d} lds_log {l}: {B / r[0] / 1e6:7.3f} M cw/s  ({r[0] * 1e3:7.2f} ms, {100 * (base / r[0] - 1):+5.1f} % vs default, same bits: {r[1]})")
    code.set_tuning(0, 0)
    del llr, out, ref
