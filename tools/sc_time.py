#!/usr/bin/env python3
"""GPU box: timing of the list-size-1 path alone (for rocprofv3 --kernel-trace --stats). usage: sc_time.py [batch] [reps]"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, polar_amd
B = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
R = int(sys.argv[2]) if len(sys.argv) > 2 else 5
C.CDLL(None).srand(1)
g = polar_amd.PolarCode(11, 1024, 0.32, 0)
d = torch.empty((B, 2048), dtype=torch.float64, device="cuda")
o = torch.empty((B, 1024), dtype=torch.uint8, device="cuda")
g.synth_llr_dev(4242, 0, B, g.snr_sqrt_linear(2.0), d.data_ptr())
g.decode_scl_llr_dev(d.data_ptr(), B, 1, o.data_ptr()); torch.cuda.synchronize()
t0 = time.time()
for _ in range(R):
    g.decode_scl_llr_dev(d.data_ptr(), B, 1, o.data_ptr())
torch.cuda.synchronize()
dt = (time.time() - t0) / R
print(f"L=1 N=2048 K=1024: {dt*1e3:.2f} ms per {B} codewords = {B/dt/1e6:.2f} M cw/s")
