#!/usr/bin/env python3
"""GPU box: L=1 timing for a given code. usage: sc_time_n.py n K [batch] [reps]"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, polar_amd
n, K = int(sys.argv[1]), int(sys.argv[2])
B = int(sys.argv[3]) if len(sys.argv) > 3 else 262144
R = int(sys.argv[4]) if len(sys.argv) > 4 else 10
C.CDLL(None).srand(1)
g = polar_amd.PolarCode(n, K, 0.32, 0)
d = torch.empty((B, 1 << n), dtype=torch.float64, device="cuda")
o = torch.empty((B, K), dtype=torch.uint8, device="cuda")
g.synth_llr_dev(4242, 0, B, g.snr_sqrt_linear(2.0), d.data_ptr())
g.decode_scl_llr_dev(d.data_ptr(), B, 1, o.data_ptr()); torch.cuda.synchronize()
t0 = time.time()
for _ in range(R):
    g.decode_scl_llr_dev(d.data_ptr(), B, 1, o.data_ptr())
torch.cuda.synchronize()
dt = (time.time() - t0) / R
print(f"L=1 N={1<<n} K={K}: {dt*1e3:.3f} ms per {B} codewords = {B/dt/1e6:.2f} M cw/s")
