#!/usr/bin/env python3
"""Cycle breakdown of sc_lat_kernel by op type (a -DSCLAT_PROF build; the counters land in the handle's alpha scratch).
usage: tools/sclat_prof.py"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("POLAR_BUILD_TAG", "sclatprof")
os.environ.setdefault("POLAR_DEFS", "SCLAT_PROF POLAR_DEV_GS32")
import numpy as np, torch
from polar_amd import build
lib = build.build()
import polar_amd
polar_amd.LIB_PATH = lib
C.CDLL(None).srand(1)
g = polar_amd.PolarCode(11, 1024, 0.32, 16)
B = 1
llr = torch.empty((B, 2048), dtype=torch.float64, device="cuda")
out = torch.empty((B, 1024), dtype=torch.uint8, device="cuda")
g.synth_llr_dev(5, 0, B, g.snr_sqrt_linear(2.0), llr.data_ptr())
g.decode_scl_llr_dev(llr.data_ptr(), B, 1, out.data_ptr())
torch.cuda.synchronize()
# the scratch pointer: find it through a second handle-free route — the library zeroes nothing there, so clear it via a big memset
L = polar_amd.lib()
L.polar_debug_scratch_ptr.restype = C.c_void_p
ptr = L.polar_debug_scratch_ptr(g._h)
buf = (C.c_uint64 * 32)()
import ctypes
hip = C.CDLL("libamdhip64.so")
hip.hipMemset(C.c_void_p(ptr), 0, 32 * 8)
n = 20
for _ in range(n):
    g.decode_scl_llr_dev(llr.data_ptr(), B, 1, out.data_ptr())
torch.cuda.synchronize()
hip.hipMemcpy(buf, C.c_void_p(ptr), 32 * 8, 2)
a = np.array(buf[:], dtype=np.float64) / n
names = {0: "f, S < 64", 1: "g, S < 64", 8: "f / g, S >= 64", 3: "all-unfrozen node", 4: "combine", 6: "all-frozen bound", 7: "mixed node of 8 in registers"}
tot = a[21]
print(f"cycles per codeword (counter units): total {tot:.0f}, front pass {a[20]:.0f} ({100 * a[20] / tot:.1f} %)")
for k, nm in names.items():
    if a[10 + k]:
        print(f"  {nm:32s}: {a[10 + k]:6.0f} ops, {a[k]:9.0f} cycles ({100 * a[k] / tot:5.1f} %), {a[k] / a[10 + k]:7.0f} per op")
