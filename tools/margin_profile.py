#!/usr/bin/env python3
"""Decision margins of the list decoder (a -DPOLAR_MARGIN build, run on the GPU box): for every codeword the smallest
gap between the worst surviving and the best discarded fork metric over all pruning steps, and the gap between the
two best final candidates. A state kept in lower precision (e.g. fp32 HBM layers) is exact only for codewords whose
gaps exceed its metric error bound; the rest would have to be flagged and decoded again in fp64.
usage: tools/margin_profile.py [batch]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["POLAR_BUILD_TAG"] = "margin"
os.environ["POLAR_DEFS"] = "POLAR_DEV_GS32 POLAR_MARGIN"
import numpy as np, torch
from polar_amd import build
lib = build.build()
import polar_amd
polar_amd.LIB_PATH = lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
C.CDLL(None).srand(1)
g = polar_amd.PolarCode(11, 1024, 0.32, 16)
llr = torch.empty((B, 2048), dtype=torch.float64, device="cuda")
out = torch.empty((B, 1024), dtype=torch.uint8, device="cuda")
print(f"N=2048 K=1024 crc16 L=32, {B} codewords per point; fraction of codewords whose smallest gap is below eps")
print("Eb/N0   eps:      1e-2     1e-3     1e-4     1e-5     1e-6     1e-7     1e-8 |  final-selection gap < 1e-3   1e-5")
for ebno in (1.0, 2.0, 3.0):
    g.synth_llr_dev(31337, 0, B, g.snr_sqrt_linear(ebno), llr.data_ptr())
    g.decode_scl_llr_dev(llr.data_ptr(), B, 32, out.data_ptr())
    torch.cuda.synchronize()
    v = out.cpu().numpy()[:, :16].copy().view(np.float64)
    mg, fg = v[:, 0], v[:, 1]
    row = " ".join(f"{(mg < e).mean():8.4f}" for e in (1e-2, 1e-3, 1e-4, 1e-5, 1e-6, 1e-7, 1e-8))
    print(f"{ebno:4.1f} dB        {row} | {(fg < 1e-3).mean():8.4f} {(fg < 1e-5).mean():8.4f}   (median prune gap {np.median(mg):.3g})")
