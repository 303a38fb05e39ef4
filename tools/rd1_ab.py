#!/usr/bin/env python3
"""tools/rd1_ab.py — round 6: layer 1 re-derived from the channel row (rd1) against layer 1 stored per path (knob "no_rd1"), lane
groups of 4 / 8 / 16, on BASELINE configurations 3 and 5 and a list of 16: same library, interleaved rounds, dominant-kernel time
from HIP events, bits compared between the two forms and with the CPU side on a prefix.

    python tools/rd1_ab.py [--rounds 5] [--out gpurun_out/rd1_ab.json]
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np
import torch

import bench

CASES = [("config3", None), ("config5", None), ("config3_b262144", None), ("config5_b262144", None),
         ("n11_L16", (11, 1024, 16, 16, 65536, 2.0)), ("n10_L8_bpsk", (10, 512, 8, 8, 65536, 2.0))]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "rd1_ab.json"))
    a = ap.parse_args()
    import polar_amd
    import oracle_lib
    dev = torch.device("cuda", 0)
    res = []
    for name, spec in CASES:
        if spec is None:
            n, K, crc, L, B, axis, const, _, _ = bench.OTHER_CONFIGS[name]
            code = bench.make_config(name)
        else:
            n, K, crc, L, B, axis = spec
            const = "bpsk"
            C.CDLL(None).srand(C.c_uint(1))
            code = polar_amd.PolarCode(n, K, 0.32, crc)
        N = 1 << n
        llr = torch.empty((B, N), dtype=torch.float64, device=dev)
        out = torch.empty((B, K), dtype=torch.uint8, device=dev)
        if const == "bpsk":
            code.synth_llr_dev(5, 0, B, code.snr_sqrt_linear(axis), llr.data_ptr())
        else:
            code.synth_bicm_llr_dev(const, 5, 0, B, axis, llr.data_ptr())
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record(); ev[1].record()
        t = {0: [], 1: []}
        wall = {0: [], 1: []}
        bits = {}
        for r in range(a.rounds + 1):
            for off in (0, 1):
                code.debug_set("no_rd1", off)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                code.decode_scl_llr_dev(llr.data_ptr(), B, L, out.data_ptr(), ev_start=ev[0].cuda_event, ev_stop=ev[1].cuda_event)
                torch.cuda.synchronize()
                if r:
                    wall[off].append(time.perf_counter() - t0)
                    t[off].append(ev[0].elapsed_time(ev[1]))
                else:
                    bits[off] = out.cpu().numpy().copy()
        code.debug_set("no_rd1", 0)
        kind = "reference" if oracle_lib.have_reference() else "port"
        C.CDLL(None).srand(C.c_uint(1))
        cpu = (oracle_lib.Reference if kind == "reference" else oracle_lib.Oracle)(n, K, 0.32, crc, srand=1)
        if const != "bpsk":
            cpu.set_tables(code.frozen_bits, code.channel_order_descending)
        ncpu = 512 if L <= 8 else 128
        want = cpu.decode_scl_llr(llr[:ncpu].cpu().numpy(), L)
        rec = {"case": name, "N": N, "L": L, "batch": B, "kernel_ms_rd1": float(np.median(t[0])), "kernel_ms_stored": float(np.median(t[1])),
               "speedup": float(np.median(t[1]) / np.median(t[0])), "cw_per_s_rd1": B / float(np.median(wall[0])), "cw_per_s_stored": B / float(np.median(wall[1])),
               "bits_equal_between_forms": bool((bits[0] == bits[1]).all()),
               "mismatching_codewords_vs_cpu_%s_first_%d" % (kind, ncpu): int((bits[0][:ncpu] != want).any(axis=1).sum())}
        print(json.dumps(rec), flush=True)
        res.append(rec)
        del llr, out
        code.close()
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(res, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
