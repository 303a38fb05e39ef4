#!/usr/bin/env python3
"""Generate tests/golden/polarm_fixtures.npz — expected outputs of the PolarM-only rows (decode_sc_p1, the ASK/BICM
front end, the Monte-Carlo construction's genie decoder, the BPSK workload) from tests/polarm_numpy.py, a pure
numpy/Python evaluation of the MATLAB formulas and of Philox4x32-10. No reference file, no product header and no
oracle library is read: these rows stay "unpinned by the reference" (MATLAB cannot run here), but the device and the
C restatement are compared against THIS independent evaluation instead of against each other.

    python tests/golden/make_polarm_fixtures.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import golden_util as G          # noqa: E402  (code tables of the committed golden set: data only)
import polarm_numpy as M         # noqa: E402

out = {}
rng = np.random.default_rng(20260928)

# ---- decode_sc_p1 (PolarCode.m:290-295, 870-895): random p1, near-0/1, and the exact-0.5 quirk (:873)
for name in ("small_n5_k16_crc4", "cfg1_n9_k256"):
    c, frozen, order, _ = G.tables(name)
    N = 1 << c["n"]
    B = 24
    p1 = rng.uniform(0.02, 0.98, (B, N))
    p1[1] = 0.5                                          # every leaf exactly 0.5 -> 0.5 on unfrozen leaves
    p1[2] = rng.choice([0.5, 0.25, 0.75], N)
    p1[3] = np.clip(rng.normal(0.2, 0.3, N), 1e-6, 1 - 1e-6)
    u = np.stack([M.polar_decode(p1[b], frozen.astype(float))[0] for b in range(B)])
    out[f"scp1/{name}/p1"] = p1
    out[f"scp1/{name}/u"] = u                            # all N leaf decisions; decode_sc_p1 returns u[info_bits(1:K)]

# ---- BPSK workload: seeds -> LLRs (PolarCode.cpp:715,744-752 on Philox/Box-Muller inputs), libm evaluation
c, frozen, order, crcm = G.tables("small_n5_k16_crc4")
n, K, crc = c["n"], c["K"], c["crc"]
N = 1 << n
seed, trial0, B, ebno = 4711, 250, 6, 1.5
s = 10.0 ** (ebno / 20) * np.sqrt(K / N)               # PolarCode.cpp:744-745 (pow(10.0f, x) promotes to double)
llr = np.zeros((B, N))
infos = np.zeros((B, K), np.uint8)
for b in range(B):
    t = trial0 + b
    info = M.bits_from_words(seed, t // 100, 1, K)
    u = np.zeros(N, np.uint8)
    u[order[:K]] = info
    for i in range(crc):
        u[order[K + i]] = (crcm[i] & info).sum() % 2
    x = u.copy()                                         # PolarCode.cpp:76-87: butterfly + bit-reversed readout
    inc = 1
    while inc < N:
        for i in range(0, N, 2 * inc):
            x[i:i + inc] ^= x[i + inc:i + 2 * inc]
        inc *= 2
    rev = [int(format(i, f"0{n}b")[::-1], 2) for i in range(N)]
    coded = x[rev]
    for pr in range(N // 2):
        z = M.normal_pair(seed, t, pr, 0)
        for k in range(2):
            y = s * (2.0 * coded[2 * pr + k] - 1.0) + np.sqrt(0.5) * z[k]
            llr[b, 2 * pr + k] = -4 * y * s
    infos[b] = info
out["bpsk/params"] = np.array([seed, trial0, B], np.int64)
out["bpsk/s"] = np.array([s])
out["bpsk/llr"] = llr
out["bpsk/info"] = infos

# ---- ASK Gray BICM front end (Constellation.m:84-93, 123-144): the config-5 code, three constellations
c5, frozen5, order5, _ = G.tables("cfg5_n10_k512_ask16")
N5, K5 = 1024, 512
for cid in (1, 2, 3):
    seed, trial, snr = 600 + cid, 12, 9.0 + cid
    info = M.bits_from_words(seed, trial, 1, K5)          # fresh info every run: block = trial (info_block_div = 1)
    u = np.zeros(N5, np.uint8)
    u[order5[:K5]] = info
    x = u.copy()
    inc = 1
    while inc < N5:
        for i in range(0, N5, 2 * inc):
            x[i:i + inc] ^= x[i + inc:i + 2 * inc]
        inc *= 2
    rev = [int(format(i, "010b")[::-1], 2) for i in range(N5)]
    coded = x[rev]
    xs, sym = M.modulate(coded, cid)
    sigma = np.sqrt(0.5) * 10 ** (-snr / 20)
    y = xs + sigma * M.symbol_noise(seed, trial, len(xs))
    p1, l = M.compute_llr_bicm(y, sigma ** 2, cid)
    full = np.zeros(N5)
    full[: len(l)] = l
    out[f"bicm/{cid}/params"] = np.array([seed, trial], np.int64)
    out[f"bicm/{cid}/snr"] = np.array([snr])
    out[f"bicm/{cid}/sym"] = sym.astype(np.int32)
    out[f"bicm/{cid}/llr"] = full
    out[f"bicm/{cid}/info"] = info

# ---- Monte-Carlo construction counts (PolarCode.m:143-196 + :897-914): small cases, every constellation
for (n_, cid, snr, seed, trial0, runs) in ((6, 4, 1.0, 5, 0, 60), (7, 3, 11.0, 6, 100, 40), (6, 1, 4.0, 7, 3, 50), (8, 2, 8.0, 8, 0, 24)):
    out[f"mc/{n_}_{cid}/params"] = np.array([n_, cid, seed, trial0, runs], np.int64)
    out[f"mc/{n_}_{cid}/snr"] = np.array([snr])
    out[f"mc/{n_}_{cid}/counts"] = M.monte_carlo_counts(n_, cid, snr, seed, trial0, runs)

np.savez_compressed(os.path.join(HERE, "polarm_fixtures.npz"), **out)
print("wrote", len(out), "arrays")
