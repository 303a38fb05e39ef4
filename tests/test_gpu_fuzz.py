"""GPU: a seeded, bounded slice of the parity fuzzers in the driver-visible suite (round-2 verdict: the fuzzers found
three real defects and lived in tools/ only). Through the C-ABI, against the oracle (PolarCode.cpp:130-190)."""
import json
import os
import time

import numpy as np
import pytest

import fuzz_util

pytestmark = pytest.mark.gpu
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")


def test_fuzz_slice_codes_a_construction_produces(built_lib, oracle_built):
    """120 random configurations (block lengths 8 .. 4096, rates <= 0.6, design parameters 0.32 .. 0.5, list sizes 1 .. 64
    incl. non-powers of two, -1 .. 4.5 dB, 15 % of them with three degenerate rows): ZERO mismatching codewords on the
    ordinary rows. The degenerate rows (all-zero, +-1000 alternating, LLRs x 1e-3: the reference decides on its own
    rounding noise there) are counted and recorded, not asserted."""
    rng = np.random.default_rng(20260928)
    t0 = time.time()
    total = bad = bad_deg = n_deg = 0
    fails = []
    for it in range(120):
        cfg = fuzz_util.draw(rng, sane=True)
        B, b, bd, deg, rows = fuzz_util.run_one(cfg, it, rng, 0.25)
        total += B; bad += b; bad_deg += bd; n_deg += 3 if deg else 0
        if b:
            fails.append((it, cfg, rows[:8].tolist()))
    rec = {"configurations": 120, "codewords": total, "mismatching_ordinary_rows": bad, "degenerate_rows": n_deg,
           "mismatching_degenerate_rows": bad_deg, "seconds": round(time.time() - t0, 1)}
    print("fuzz slice:", json.dumps(rec))
    try:
        os.makedirs(OUT, exist_ok=True)
        json.dump(rec, open(os.path.join(OUT, "fuzz_slice.json"), "w"))
    except OSError:
        pass
    assert bad == 0, fails


@pytest.mark.parametrize("n,K,eps,L,ebno", [(9, 505, 0.7, 5, 2.0), (10, 1006, 0.32, 8, 3.0), (10, 1022, 0.32, 32, 2.0)])
def test_weak_unfrozen_leaves_take_the_llr_domain_kernel(built_lib, oracle_built, n, K, eps, L, ebno):
    """Codes with unfrozen leaves in the worst channels (rate near 1 / a design parameter that does not describe the
    channel): the reference's decisions there are the rounding noise of its own arithmetic, which the LLR-domain kernel
    follows much further down than the exp-domain one (DESIGN.md "Where bit-exactness ends": 664 vs 60 differing codewords
    of 4 096 at K = 505 of 512, L = 5). The handle classifies such leaves at creation and the exp-domain kernel hands every
    codeword in which one of them comes out below 1e-8 to the LLR-domain kernel: automatic mode must not be worse than
    the LLR-domain kernel alone."""
    import ctypes as C
    import polar_amd
    from oracle_lib import Oracle
    o = Oracle(n, K, eps, 0, srand=1)
    C.CDLL(None).srand(C.c_uint(1))
    g = polar_amd.PolarCode(n, K, eps, 0)
    llr, _ = o.synth_llr(4242, 0, 2048, o.snr_sqrt_linear(ebno))
    want = o.decode_scl_llr(llr, L)
    g.set_mode(1); llr_dom = int((want != g.decode_scl_llr(llr, L)).any(axis=1).sum())
    g.set_mode(0); auto = int((want != g.decode_scl_llr(llr, L)).any(axis=1).sum())
    g.set_mode(2); forced = int((want != g.decode_scl_llr(llr, L)).any(axis=1).sum())
    print(f"n={n} K={K} eps={eps} L={L}: differing codewords of 2048 — LLR-domain {llr_dom}, automatic {auto}, exp-domain forced {forced}")
    assert auto <= llr_dom and forced <= llr_dom
