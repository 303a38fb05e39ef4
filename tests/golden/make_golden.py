#!/usr/bin/env python3
"""Generate tests/golden/polar_golden.npz from the UNMODIFIED reference (oracle/_ref, i.e.
/root/reference/PolarC compiled where it lies — `make -C oracle ref`).  Runs only in the build
container; the fixtures it writes are DATA (inputs are re-derived from seeds through the synthetic
workload definition include/polar_synth.h, pinned by a sha256 of the LLR bytes; expected outputs
are the reference's decoded info bits, bit-packed) and travel to the GPU box with the repo.

    python tests/golden/make_golden.py
"""
import ctypes as C
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle_lib import Oracle, Reference, build_oracle  # noqa: E402

libc = C.CDLL(None)

# name -> (n, K, crc, [(L, ebno_dB, B), ...])
CODES = {
    "small_n5_k16_crc4": (5, 16, 4, [(L, e, 64) for L in (1, 2, 3, 5, 8, 32) for e in (0.0, 3.0)]),
    "cfg1_n9_k256": (9, 256, 0, [(1, 1.0, 256), (1, 2.0, 256), (1, 3.0, 256), (4, 1.5, 128), (32, 1.5, 64)]),
    "n10_k512_crc8": (10, 512, 8, [(2, 1.5, 128), (8, 1.5, 128), (16, 2.0, 64)]),
    "cfg2_n11_k1024": (11, 1024, 0, [(1, 1.0, 256), (1, 2.0, 512), (2, 1.5, 128), (8, 1.5, 64)]),
    "cfg3_n11_k1024_crc16": (11, 1024, 16, [(4, 1.0, 128), (4, 1.5, 128), (4, 2.0, 128), (4, 2.5, 128),
                                            (32, 1.0, 64), (32, 1.5, 64), (32, 2.0, 64)]),
}
SEED = 20240927


def special_inputs(N, rng):
    """Edge cases of the domain: zeros, huge magnitudes (fp64 exp overflow > 709.78), exact-tie
    constructions (+-c), the |llr| = 40 branch boundary of the f-node, single erasure-like zeros."""
    sp = {}
    sp["all_zero"] = np.zeros(N)
    sgn = rng.choice([-1.0, 1.0], N)
    sp["huge_pm1000"] = 1000.0 * sgn
    sp["huge_mixed"] = sgn * rng.choice([0.5, 39.999, 40.0, 40.001, 88.8, 709.0, 710.0, 5000.0], N)
    sp["const_plus"] = np.full(N, 2.0)
    sp["const_minus"] = np.full(N, -2.0)
    sp["ties_pm1"] = rng.choice([-1.0, 1.0], N)
    x = rng.normal(3.0, 2.5, N)
    x[rng.integers(0, N, N // 8)] = 0.0
    sp["gauss_with_zeros"] = x
    sp["tiny"] = rng.normal(0, 1e-12, N)
    return sp


def main():
    build_oracle()
    out = {}
    meta = {"seed": SEED, "codes": {}, "generator": "tests/golden/make_golden.py",
            "reference": "tavildar/Polar PolarC/PolarCode.cpp compiled unmodified (oracle/Makefile target ref)"}
    for name, (n, K, crc, cases) in CODES.items():
        libc.srand(1)
        ref = Reference(n, K, 0.32, crc)
        libc.srand(1)
        orc = Oracle(n, K, 0.32, crc)        # used ONLY for the synthetic input generator below
        N = 1 << n
        out[f"{name}/frozen"] = np.packbits(ref.frozen())
        out[f"{name}/order"] = ref.order()
        out[f"{name}/crcm"] = np.packbits(ref.crc_matrix().reshape(-1)) if crc else np.zeros(0, np.uint8)
        rng = np.random.default_rng(1000 + n)
        info = rng.integers(0, 2, (16, K)).astype(np.uint8)
        out[f"{name}/enc_info"] = np.packbits(info.reshape(-1))
        out[f"{name}/enc_coded"] = np.packbits(np.stack([ref.encode(i) for i in info]).reshape(-1))
        cm = []
        for ci, (L, ebno, B) in enumerate(cases):
            s = orc.snr_sqrt_linear(ebno)
            trial0 = 1000 * ci
            llr, sent = orc.synth_llr(SEED, trial0, B, s)
            dec = ref.decode_scl_llr(llr, L)
            out[f"{name}/case{ci}/decoded"] = np.packbits(dec.reshape(-1))
            cm.append({"L": L, "ebno": ebno, "B": B, "trial0": trial0, "s_hex": float(s).hex(),
                       "llr_sha256": hashlib.sha256(llr.tobytes()).hexdigest(),
                       "block_errors": int((dec != sent).any(axis=1).sum())})
            print(name, cm[-1])
        # special raw inputs (small codes only keep the file small)
        spm = []
        if N <= 512:
            sp = special_inputs(N, np.random.default_rng(77 + n))
            for sname, x in sp.items():
                out[f"{name}/special/{sname}/llr"] = x
                for L in (1, 4, 32):
                    d = ref.decode_scl_llr(x, L)
                    out[f"{name}/special/{sname}/L{L}"] = np.packbits(d)
                spm.append(sname)
        meta["codes"][name] = {"n": n, "K": K, "crc": crc, "eps": 0.32, "cases": cm, "specials": spm}

    # ---- config 5: N=1024 K=512, Monte-Carlo constructed code for 16-ASK Gray BICM -------------
    # frozen set / info order from the reference's own data file (PolarCode.m:111-135: load the
    # error counts, stable ascending sort, first K positions are the info bits), decoded by the
    # reference's C decoder with those tables; inputs from the ASK/BICM front end of polar_synth.h
    txt = ("/root/reference/PolarM/CodeConstructionData/MC_block_length_1024_512_cc_method_monte-carlo_"
           "cc_param_13_ask16-gray_bicm_250000.txt")
    counts = np.loadtxt(txt)
    assert counts.shape == (1024,)
    order5 = np.argsort(counts, kind="stable").astype(np.uint16)        # MATLAB sort is stable
    assert counts[order5[511]] < counts[order5[512]]                    # no tie at the K boundary (702 vs 728)
    frozen5 = np.ones(1024, np.uint8)
    frozen5[order5[:512]] = 0
    name = "cfg5_n10_k512_ask16"
    ref = Reference(10, 512, 0.5, 0)
    ref.set_tables(frozen5, order5)
    orc = Oracle(10, 512, 0.5, 0)
    orc.set_tables(frozen5, order5)
    out[f"{name}/counts"] = counts.astype(np.int32)          # the data file itself (1024 integers), as a fixture
    out[f"{name}/frozen"] = np.packbits(frozen5)
    out[f"{name}/order"] = order5
    out[f"{name}/crcm"] = np.zeros(0, np.uint8)
    rng = np.random.default_rng(5)
    info = rng.integers(0, 2, (16, 512)).astype(np.uint8)
    out[f"{name}/enc_info"] = np.packbits(info.reshape(-1))
    out[f"{name}/enc_coded"] = np.packbits(np.stack([ref.encode(i) for i in info]).reshape(-1))
    cm = []
    for ci, (L, snr, B) in enumerate([(8, 12.0, 128), (8, 13.0, 128), (8, 14.0, 128), (1, 13.5, 128), (32, 12.5, 32)]):
        trial0 = 1000 * ci
        llr, sent = orc.synth_bicm_llr(3, SEED, trial0, B, snr)
        dec = ref.decode_scl_llr(llr, L)
        out[f"{name}/case{ci}/decoded"] = np.packbits(dec.reshape(-1))
        cm.append({"L": L, "snr_db": snr, "constellation": 3, "B": B, "trial0": trial0,
                   "llr_sha256": hashlib.sha256(llr.tobytes()).hexdigest(),
                   "block_errors": int((dec != sent).any(axis=1).sum())})
        print(name, cm[-1])
    meta["codes"][name] = {"n": 10, "K": 512, "crc": 0, "eps": None, "explicit_tables": True, "cases": cm, "specials": [],
                           "source": "PolarM/CodeConstructionData/..._13_ask16-gray_bicm_250000.txt"}

    # the reference's own deterministic driver output (main.cpp: n=11, K=1024, crc=0, eps=0.32,
    # Eb/N0 1:0.25:2, L = 1,2,4,8,32; 1000 runs, max_err 100) — SURVEY §6 / BASELINE.md §2
    libc.srand(1)
    ref = Reference(11, 1024, 0.32, 0)
    ebno = [1.0, 1.25, 1.5, 1.75, 2.0]
    Ls = [1, 2, 4, 8, 32]
    table = ref.get_bler_quick(ebno, Ls)
    out["main_cpp/bler"] = table
    meta["main_cpp"] = {"ebno": ebno, "L": Ls, "n": 11, "K": 1024, "crc": 0, "eps": 0.32}
    print(table.T)
    # a shorter deterministic get_bler_quick run for the CPU suite (n=9)
    libc.srand(1)
    ref = Reference(9, 256, 0.32, 8)
    t2 = ref.get_bler_quick([1.0, 1.5, 2.0, 2.5], [1, 2, 8])
    out["bler_n9/bler"] = t2
    meta["bler_n9"] = {"ebno": [1.0, 1.5, 2.0, 2.5], "L": [1, 2, 8], "n": 9, "K": 256, "crc": 8, "eps": 0.32}

    np.savez_compressed(os.path.join(HERE, "polar_golden.npz"), **out)
    json.dump(meta, open(os.path.join(HERE, "polar_golden.json"), "w"), indent=1)
    print("wrote", os.path.getsize(os.path.join(HERE, "polar_golden.npz")), "bytes")


if __name__ == "__main__":
    main()
