import sys, ctypes as C
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np
import golden_util as G, polar_amd
from oracle_lib import Oracle
for name in G.code_names():
    c = G.load()[1]["codes"][name]
    if not c["specials"]: continue
    C.CDLL(None).srand(1)
    g = polar_amd.PolarCode(c["n"], c["K"], c["eps"], c["crc"])
    C.CDLL(None).srand(1)
    o = Oracle(c["n"], c["K"], c["eps"], c["crc"])
    for sname, llr, exp in G.specials(name):
        for L, want in exp.items():
            got = g.decode_scl_llr(llr, L)
            if not (got == want).all():
                _, pm = o.decode_scl_llr_pm(llr, L)
                print("MISMATCH", name, sname, L, "nbits", int((got!=want).sum()), "oracle pm", pm)
