#!/usr/bin/env python3
"""Per-phase cycle breakdown of scl_decode_llr_kernel (instrumented build, -DPOLAR_PROFILE).
Run on the GPU box: python tools/phase_profile.py [L] [batch] [wpc] [lds_log]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from polar_amd import build
lib = build.build(profile=True)
import polar_amd
polar_amd.LIB_PATH = lib
L = int(sys.argv[1]) if len(sys.argv) > 1 else 32
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
wpc = int(sys.argv[3]) if len(sys.argv) > 3 else 0
ll = int(sys.argv[4]) if len(sys.argv) > 4 else 0
C.CDLL(None).srand(1)
g = polar_amd.PolarCode(11, 1024, 0.32, 16)
g.set_tuning(wpc, ll)
llr = torch.empty((B, 2048), dtype=torch.float64, device="cuda")
out = torch.empty((B, 1024), dtype=torch.uint8, device="cuda")
prof = torch.zeros(max(B, 64), dtype=torch.int64, device="cuda")
g.synth_llr_dev(1, 0, B, g.snr_sqrt_linear(float(os.environ.get("EBNO", "2.0"))), llr.data_ptr())
g.decode_scl_llr_dev(llr.data_ptr(), B, L, out.data_ptr(), prof.data_ptr())
torch.cuda.synchronize()
prof.zero_()
torch.cuda.synchronize()
import time
t = time.perf_counter()
g.decode_scl_llr_dev(llr.data_ptr(), B, L, out.data_ptr(), prof.data_ptr())
torch.cuda.synchronize()
dt = time.perf_counter() - t
a = prof[:24].cpu().numpy().astype(float)
names = ["loop ovh", "layer S>SL g (HBM)", "layer S>SL f (HBM)", "layer 4<=S<=SL (LDS)", "layer S<4", "leaf frozen / rate-0 block", "leaf unfrozen (rest: flush etc.)", "partial sums", "unf: DPP reductions+decision", "unf: competitive-bad loop", "unf: stack/srcof", "unf: clone shuffles+update", "-", "unf: setup+goods rank loop", "unf: wait for leaf (ballot)", "unf: softplus+bounds"]
print(f"L={L} B={B} time {dt*1e3:.2f} ms -> {B/dt:.0f} cw/s")
cyc = a[:8].sum() + a[16:24].sum()
nwd = B / (64 // max(1, 1 << (L - 1).bit_length()))      # wave-decodes
if os.environ.get("LATPROF"):                             # the one-codeword-per-wave kernels: a wave-decode is a codeword
    nwd = B
print(f"  cycles per wave-decode: {cyc/nwd:.0f}")
for nm, v in zip(names[:8], a[:8]):
    print(f"  {nm:34s} {100*v/cyc:5.1f}%  {v/nwd:10.0f} cycles/wave-decode")
sub = ["unf: leaf wait + softplus + DPP bounds", "unf: fast path commit", "unf: goods rank loop", "unf: competitive-bad loop",
       "unf: kill/clone LIFO (LDS)", "unf: clone shuffles + update", "-", "-"]
for nm, v in zip(sub, a[16:24]):
    if v:
        print(f"    {nm:40s} {100*v/cyc:5.1f}%  {v/nwd:10.0f}")
if a[8] > 0:
    print(f"  unfrozen steps (per wave) {a[8]:.0f}: fast path {100*a[9]/a[8]:.1f}%, full-list ranking {100*a[10]/a[8]:.1f}%")
    if a[10] > 0:
        print(f"  per ranking step: competitive bad forks (union over groups) {a[11]/a[10]:.2f}, contested good forks {a[12]/a[10]:.2f}")
        print(f"  per ranking step: surviving bad forks (max over groups) {a[13]/a[10]:.2f}; steps with <= 4: {100*a[14]/a[10]:.1f}%, with none: {100*a[15]/a[10]:.1f}%")
