#!/usr/bin/env python3
"""tools/host_pipe_crossover.py — host-pointer batches between 32 MiB and 1 GiB of LLRs: the chunked pipeline (a launch of the list kernels
takes 4 .. 8 ms whatever it carries: n chunks on k lanes are n / k launch latencies) against one copy in, one launch, one copy out.
polar_debug_set "host_pipe_min_bytes" (-1: never pipeline). usage: tools/host_pipe_crossover.py [out.json]"""
import ctypes as C, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import polar_amd, oracle_lib
o = oracle_lib.Oracle(11, 1024, 0.32, 16, srand=1)
C.CDLL(None).srand(C.c_uint(1))
g = polar_amd.PolarCode(11, 1024, 0.32, 16)
base, _ = o.synth_llr(99, 0, 4096, o.snr_sqrt_linear(2.0))
rows = []
for L in (1, 2, 4, 8, 32):
    for B in (1024, 2048, 4096, 8192, 16384, 32768, 65536):
        x = np.ascontiguousarray(np.tile(base, (max(1, B // 4096), 1))[:B])
        res = {}
        for name, knob in (("pipelined (32 MiB rule)", 32 << 20), ("single copy", -1), ("default", 0)):
            g.debug_set("host_pipe_min_bytes", knob)
            got = g.decode_scl_llr(x, L)
            ts = []
            for _ in range(5):
                t = time.perf_counter(); g.decode_scl_llr(x, L); ts.append(time.perf_counter() - t)
            res[name] = float(np.median(ts) * 1e3)
            if name == "pipelined (32 MiB rule)": ref = got
            else: assert (got == ref).all()
        g.debug_set("host_pipe_min_bytes", 0)
        rows.append(dict(L=L, B=B, MiB=B * 2048 * 8 / 2**20, **res))
        print(f"L={L} B={B} ({B * 2048 * 8 >> 20} MiB): pipelined {res['pipelined (32 MiB rule)']:.2f} ms, single copy {res['single copy']:.2f} ms, default {res['default']:.2f} ms", flush=True)
if len(sys.argv) > 1: json.dump(rows, open(sys.argv[1], "w"), indent=1)
