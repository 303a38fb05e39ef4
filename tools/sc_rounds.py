#!/usr/bin/env python3
"""GPU box: list-size-1 throughput (N = 2048, K = 1024, 2 dB) against the batch size, to show what the quantisation of the
batch into rounds of resident waves costs: one MI355X holds 256 CUs x 20 waves x 8 codewords = 40 960 codewords at a time, so
65 536 codewords are 1.6 rounds (the second one at 60 % occupancy). usage: tools/sc_rounds.py [out.json]"""
import ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import polar_amd

C.CDLL(None).srand(1)
g = polar_amd.PolarCode(11, 1024, 0.32, 0)
res = []
for B in (20480, 40960, 61440, 65536, 81920, 122880, 163840, 262144):
    llr = torch.empty((B, 2048), dtype=torch.float64, device="cuda")
    out = torch.empty((B, 1024), dtype=torch.uint8, device="cuda")
    g.synth_llr_dev(1, 0, B, g.snr_sqrt_linear(2.0), llr.data_ptr())
    for _ in range(2):
        g.decode_scl_llr_dev(llr.data_ptr(), B, 1, out.data_ptr())
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(8):
        g.decode_scl_llr_dev(llr.data_ptr(), B, 1, out.data_ptr())
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 8
    res.append({"batch": B, "rounds_of_40960": round(B / 40960, 2), "ms": round(ms, 4), "cw_per_s": round(B / ms * 1e3)})
    print(res[-1], flush=True)
    del llr, out
if len(sys.argv) > 1:
    json.dump(res, open(sys.argv[1], "w"), indent=1)
