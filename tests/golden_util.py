"""Loader for tests/golden/polar_golden.{npz,json} (written by tests/golden/make_golden.py from the
unmodified reference)."""
import hashlib
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_npz = None
_meta = None


def load():
    global _npz, _meta
    if _npz is None:
        _npz = np.load(os.path.join(HERE, "golden", "polar_golden.npz"))
        _meta = json.load(open(os.path.join(HERE, "golden", "polar_golden.json")))
    return _npz, _meta


def code_names():
    return list(load()[1]["codes"].keys())


def tables(name):
    z, m = load()
    c = m["codes"][name]
    N, K, crc = 1 << c["n"], c["K"], c["crc"]
    frozen = np.unpackbits(z[f"{name}/frozen"])[:N]
    order = z[f"{name}/order"]
    crcm = np.unpackbits(z[f"{name}/crcm"])[: crc * K].reshape(crc, K) if crc else np.zeros((0, K), np.uint8)
    return c, frozen, order, crcm


def enc_vectors(name):
    z, m = load()
    c = m["codes"][name]
    N, K = 1 << c["n"], c["K"]
    info = np.unpackbits(z[f"{name}/enc_info"])[: 16 * K].reshape(16, K)
    coded = np.unpackbits(z[f"{name}/enc_coded"])[: 16 * N].reshape(16, N)
    return info, coded


def cases(name):
    z, m = load()
    c = m["codes"][name]
    for ci, cs in enumerate(c["cases"]):
        dec = np.unpackbits(z[f"{name}/case{ci}/decoded"])[: cs["B"] * c["K"]].reshape(cs["B"], c["K"])
        yield cs, dec


def specials(name):
    z, m = load()
    c = m["codes"][name]
    for s in c["specials"]:
        llr = z[f"{name}/special/{s}/llr"]
        exp = {L: np.unpackbits(z[f"{name}/special/{s}/L{L}"])[: c["K"]] for L in (1, 4, 32)}
        yield s, llr, exp


def all_case_ids():
    _, m = load()
    ids = []
    for name, c in m["codes"].items():
        for ci in range(len(c["cases"])):
            ids.append((name, ci))
    return ids


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def seed():
    return load()[1]["seed"]


def both_kernels(g, decode):
    """The list sizes 1 ... 8 have two kernels behind one entry point: the batch kernels (many codewords per wave, the big layers
    in an HBM scratch: throughput) and the latency kernels (ONE codeword per wave, its elements spread over the lanes, the state in
    LDS: list size 1 up to 2048 codewords and N <= 4096, list sizes 2 ... 8 up to one codeword per CU while the state fits the LDS).
    `decode()` is run with each of them forced ("lat_max_b" hook of polar_debug_set: -1 = never, a huge value = whenever the
    shape allows) and must return the same bits; returns them."""
    g.debug_set("lat_max_b", -1)
    a = decode()
    g.debug_set("lat_max_b", 1 << 40)
    b = decode()
    g.debug_set("lat_max_b", 0)
    assert (a == b).all(), "the one-codeword-per-wave kernel and the batch kernel disagree"
    return a


both_l1_kernels = both_kernels
