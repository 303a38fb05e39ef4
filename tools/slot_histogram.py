#!/usr/bin/env python3
"""Distinct source slots per layer visit of the list-of-32 kernel (a -DPOLAR_SLOTHIST build, run on the GPU box) — the
measurement the round-3 verdict asked for before any "f-visit de-duplication": for every visit whose source layer lives in
the HBM scratch, how many DISTINCT slots of that layer the active paths of one codeword read (their slot pointers
`pL.get(sh + 1)`), per visit kind and layer size, at 1 / 1.5 / 2 / 2.5 dB. If the f-visits of paths with a common ancestor
read the same slot they would compute and write identical rows.
usage: tools/slot_histogram.py [batch]   ->  table on stdout (commit it under profiles/)"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["POLAR_BUILD_TAG"] = "slothist"
os.environ["POLAR_DEFS"] = "POLAR_DEV_GS32 POLAR_SLOTHIST"
import numpy as np, torch
from polar_amd import build
lib = build.build()
import polar_amd
polar_amd.LIB_PATH = lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
C.CDLL(None).srand(1)
g = polar_amd.PolarCode(11, 1024, 0.32, 16)
llr = torch.empty((B, 2048), dtype=torch.float64, device="cuda")
out = torch.empty((B, 1024), dtype=torch.uint8, device="cuda")
buf = torch.zeros(64 + 25 * 40 + B, dtype=torch.int64, device="cuda")
print(f"N=2048 K=1024 crc16 L=32, {B} codewords per point. Per visit of a layer of size S whose SOURCE layer (size 2S) is HBM-resident:")
print("distinct = number of different source slots among the active paths of the codeword; own = paths whose source slot is their own lane's")
for ebno in (1.0, 1.5, 2.0, 2.5):
    g.synth_llr_dev(31337, 0, B, g.snr_sqrt_linear(ebno), llr.data_ptr())
    buf.zero_()
    g.decode_scl_llr_dev(llr.data_ptr(), B, 32, out.data_ptr(), buf.data_ptr())
    torch.cuda.synchronize()
    h = buf[64:64 + 25 * 40].cpu().numpy().reshape(25, 40)
    print(f"\nEb/N0 = {ebno} dB")
    for kind, name in ((0, "f"), (1, "g")):
        for sh in range(3, 10):
            row = h[kind * 12 + sh]
            n = int(row[:33].sum())
            if n == 0:
                continue
            act, own = int(row[34]), int(row[35])
            mean = float((row[:33] * np.arange(33)).sum()) / n
            cdf = np.cumsum(row[:33]) / n
            pct = lambda q: int(np.searchsorted(cdf, q))
            print(f"  {name}-visit of S = {1 << sh:4d} (source 2S = {2 << sh:4d}): {n:9d} codeword-visits, active paths/visit {act / n:5.2f}, "
                  f"distinct source slots mean {mean:5.2f} (p10 {pct(0.10)}, p50 {pct(0.5)}, p90 {pct(0.9)}), paths reading their own slot {100.0 * own / max(act, 1):6.2f} %")
    reg = h[24]
    tot = int(reg[:12].sum())
    print("  f-visits evaluated from REGISTERS inside a fused pass (source = the path's own just-computed values, never a shared slot): "
          + ", ".join(f"S={1 << sh}: {int(reg[sh])}" for sh in range(3, 10) if reg[sh]) + f" (wave-visits, total {tot})")
