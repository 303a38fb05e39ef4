"""Random code / list-size / SNR configurations for the parity fuzzers (tools/fuzz_parity.py and the seeded slice in
tests/test_gpu_fuzz.py): HIP decode through the C-ABI vs the oracle (C restatement of PolarCode.cpp:130-190)."""
import ctypes as C

import numpy as np

LIST_SIZES = [1, 1, 2, 3, 4, 5, 8, 12, 16, 17, 24, 31, 32, 32, 33, 64]


def draw(rng, sane):
    """One configuration. sane: rates <= 0.6 and design parameters 0.32 .. 0.5 — codes a construction produces for its
    channel; otherwise any K and design parameters 0.1 .. 0.7 (unfrozen leaves in the worst channels included)."""
    n = int(rng.integers(3, 13)); N = 1 << n
    crc = int(rng.choice([0, 0, 4, 8, 11, 16, 24]))
    if crc >= N - 1:
        crc = 0
    K = int(rng.integers(1, N - crc + 1))
    if sane:
        K = int(rng.integers(1, max(2, int(0.6 * N) - crc)))
    L = int(rng.choice(LIST_SIZES))
    eps = float(rng.choice([0.32, 0.32, 0.4, 0.5] if sane else [0.1, 0.32, 0.32, 0.5, 0.7]))
    ebno = float(rng.uniform(-1.0, 4.5))
    return dict(n=n, N=N, K=K, crc=crc, L=L, eps=eps, ebno=ebno)


def run_one(cfg, it, rng, oracle_seconds, degenerate_prob=0.15):
    """Decode one random batch on both sides. Returns (codewords, mismatching ordinary rows, mismatching degenerate rows,
    degenerate rows present)."""
    import polar_amd
    from oracle_lib import Oracle
    n, N, K, crc, L = cfg["n"], cfg["N"], cfg["K"], cfg["crc"], cfg["L"]
    rate = 131.0 * (2048 * 11 * 32) / (N * n * L)                 # oracle codewords/s, single thread (rough)
    B = int(min(4096, max(16, oracle_seconds * rate)))
    o = Oracle(n, K, cfg["eps"], crc, srand=it + 1)
    C.CDLL(None).srand(C.c_uint(it + 1))
    g = polar_amd.PolarCode(n, K, cfg["eps"], crc)
    llr, _ = o.synth_llr(1000 + it, 0, B, o.snr_sqrt_linear(cfg["ebno"]))
    deg = rng.random() < degenerate_prob
    if deg:                                                        # rows 0..2: all-zero, +-1000 alternating, everything x 1e-3
        llr[0] = 0.0
        llr[1] = np.where(np.arange(N) % 2 == 0, 1e3, -1e3)
        llr[2] *= 1e-3
    want = o.decode_scl_llr(llr, L)
    got = g.decode_scl_llr(llr, L)
    rows = np.nonzero((want != got).any(axis=1))[0]
    bad_deg = int((rows < 3).sum()) if deg else 0
    g.close()
    return B, int(len(rows)) - bad_deg, bad_deg, bool(deg), rows
