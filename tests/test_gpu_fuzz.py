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
    follows much further down than the exp-domain one (HISTORY.md "Where bit-exactness ends": 664 vs 60 differing codewords
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
    # (and a loose absolute bound for the LLR-domain kernel itself against the reference — measured 6 / 0 / 0 of 2048 —, so
    # that its drift on such codes is caught too)
    assert llr_dom <= 16


@pytest.mark.parametrize("n,K,crc,L,eps,ebno", [(10, 332, 11, 8, 0.32, 2.53), (10, 560, 24, 33, 0.5, 2.45), (10, 361, 0, 8, 0.5, 3.94),
                                                 (12, 1650, 0, 32, 0.32, 3.52)])
def test_rows_with_a_tiny_channel_take_the_llr_domain_kernel(built_lib, oracle_built, n, K, crc, L, eps, ebno):
    """The degenerate "LLRs x 1e-3" row of the fuzzers on the four configurations where the round-3 fuzz runs found it
    differing from the reference (4 of 3 600 such rows): the whole channel is tiny but every value is above the per-value
    input guard (1e-9), so the row used to reach the exp-domain kernel; the reference decides it at the rounding noise of its
    own arithmetic. The conversion pass now flags a codeword with no |llr| >= 0.1 and the LLR-domain kernel decodes it:
    automatic mode returns exactly what mode 1 returns on such rows, so it is never worse against the reference — and
    ordinary rows in the same batch are untouched."""
    import ctypes as C
    import polar_amd
    from oracle_lib import Oracle
    o = Oracle(n, K, eps, crc, srand=1)
    C.CDLL(None).srand(C.c_uint(1))
    g = polar_amd.PolarCode(n, K, eps, crc)
    B = 96 if n >= 12 else 384
    llr, _ = o.synth_llr(777, 0, B, o.snr_sqrt_linear(ebno))
    tiny = np.arange(B) % 2 == 0
    llr[tiny] *= 1e-3
    want = o.decode_scl_llr(llr, L)
    g.set_mode(1); a = g.decode_scl_llr(llr, L)
    g.set_mode(0); b = g.decode_scl_llr(llr, L)
    g.set_mode(2); c = g.decode_scl_llr(llr, L)
    assert (a[tiny] == b[tiny]).all() and (a[tiny] == c[tiny]).all()              # the tiny rows: the LLR-domain kernel's bits in every mode
    d_llr, d_auto = int((want != a).any(axis=1).sum()), int((want != b).any(axis=1).sum())
    print(f"n={n} K={K} L={L}: rows differing from the reference — LLR-domain {d_llr}, automatic {d_auto} (of {B}, half of them x 1e-3)")
    assert d_auto <= d_llr
    assert (want[~tiny] == b[~tiny]).all() and (want[~tiny] == a[~tiny]).all()   # ordinary rows: the reference's bits


def test_fuzz_slice_latency_kernels(built_lib, oracle_built):
    """The one-codeword-per-wave kernels (round 4: list size 1 with the state in LDS; list sizes 2 ... 8 with the elements of
    every layer spread over the lanes of a path) on 80 random configurations a construction produces — block lengths 8 ... 4096,
    list sizes 1 ... 8 incl. 3, 5, 6, 7, CRCs, -1 ... 4.5 dB, degenerate rows mixed in — forced on whatever the batch size
    ("lat_max_b"), against the oracle AND against the batch kernels: zero mismatching ordinary rows."""
    import ctypes as C
    import polar_amd
    from oracle_lib import Oracle
    rng = np.random.default_rng(4242)
    total = bad = used_lat = 0
    fails = []
    for it in range(80):
        cfg = fuzz_util.draw(rng, sane=True)
        cfg["L"] = int(rng.choice([1, 2, 3, 4, 4, 5, 6, 7, 8, 8]))
        n, N, K, crc, L = cfg["n"], cfg["N"], cfg["K"], cfg["crc"], cfg["L"]
        B = int(rng.choice([1, 3, 17, 40]))
        o = Oracle(n, K, cfg["eps"], crc, srand=it + 1)
        C.CDLL(None).srand(C.c_uint(it + 1))
        g = polar_amd.PolarCode(n, K, cfg["eps"], crc)
        llr, _ = o.synth_llr(3000 + it, 0, B, o.snr_sqrt_linear(cfg["ebno"]))
        deg = B >= 17 and rng.random() < 0.3
        if deg:                                                    # the three degenerate rows of tools/fuzz_parity.py
            llr[0] = 0.0
            llr[1] = np.where(np.arange(N) % 2 == 0, 1e3, -1e3)
            llr[2] *= 1e-3
        want = o.decode_scl_llr(llr, L)
        g.debug_set("lat_max_b", 1 << 40)
        got = g.decode_scl_llr(llr, L)
        g.debug_set("lat_max_b", -1)
        ref = g.decode_scl_llr(llr, L)
        rows = np.nonzero((want != got).any(axis=1))[0]
        ordinary = [int(r) for r in rows if not (deg and r < 3)]
        total += B; bad += len(ordinary)
        if ordinary or (got != ref).any():
            fails.append((it, cfg, B, ordinary[:6], int((got != ref).any(axis=1).sum())))
        g.close()
    print(f"latency-kernel fuzz slice: 80 configurations, {total} codewords, {bad} mismatching ordinary rows")
    assert not fails, fails


@pytest.mark.timeout(300)
@pytest.mark.parametrize("L", [1, 2, 4, 8, 32])
def test_nan_and_conflicting_infinities_terminate_and_stay_in_their_row(built_lib, oracle_built, L):
    """Rows holding NaN, or infinities of conflicting sign (inf - inf = NaN in the reference's g-node, PolarCode.cpp:449-450), are
    outside any meaningful domain: the reference's own std::sort over NaN metrics violates its ordering contract (the C
    restatement does not even return on such a row), so there is nothing to compare with. What the kernels owe the caller is to
    TERMINATE and to leave every other row of the batch exactly as it is without the bad rows — batch kernels and the
    one-codeword-per-wave kernels alike."""
    import ctypes as C
    import polar_amd
    from oracle_lib import Oracle
    o = Oracle(9, 181, 0.32, 16, srand=1)
    C.CDLL(None).srand(C.c_uint(1))
    g = polar_amd.PolarCode(9, 181, 0.32, 16)
    llr, _ = o.synth_llr(3052, 0, 40, o.snr_sqrt_linear(2.0))
    want = o.decode_scl_llr(llr, L)
    bad = llr.copy()
    bad[2, ::5] = np.inf * np.sign(bad[2, ::5] + 1e-300)          # mostly the right signs, some wrong: NaN in g
    bad[7, 3] = np.nan
    bad[11, 0], bad[11, 1] = np.inf, -np.inf
    bad[20] = np.nan
    ok = np.ones(40, bool); ok[[2, 7, 11, 20]] = False
    for lat in (-1, 1 << 40):
        g.debug_set("lat_max_b", lat)
        got = g.decode_scl_llr(bad, L)
        assert (got[ok] == want[ok]).all(), lat
    g.debug_set("lat_max_b", 0)
