import ctypes as C, sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import polar_amd
from oracle_lib import Oracle
n, K, crc, L, eps, ebno = 10, 332, 11, 8, 0.32, 2.53
o = Oracle(n, K, eps, crc, srand=1)
C.CDLL(None).srand(C.c_uint(1))
g = polar_amd.PolarCode(n, K, eps, crc)
B = 384
llr, _ = o.synth_llr(777, 0, B, o.snr_sqrt_linear(ebno))
tiny = np.arange(B) % 2 == 0
llr[tiny] *= 1e-3
want = o.decode_scl_llr(llr, L)
def run(mode, nopre):
    g.set_mode(mode); g.debug_set("no_prefix", nopre)
    return g.decode_scl_llr(llr, L)
res = {(m, p): run(m, p) for m in (0, 1, 2) for p in (0, 1)}
for k, v in res.items():
    print("mode", k[0], "no_prefix", k[1], "tiny rows differing from the reference:", int((want[tiny] != v[tiny]).any(axis=1).sum()),
          "ordinary:", int((want[~tiny] != v[~tiny]).any(axis=1).sum()))
for a in res:
    for b in res:
        if a < b:
            print(a, b, "tiny rows differing:", int((res[a][tiny] != res[b][tiny]).any(axis=1).sum()))
# the tiny rows alone in a batch
t_only = llr[tiny]
g.set_mode(1); g.debug_set("no_prefix", 0)
x = g.decode_scl_llr(t_only, L)
print("mode 1, tiny rows alone vs in the mixed batch:", int((x != res[(1, 0)][tiny]).any(axis=1).sum()))
