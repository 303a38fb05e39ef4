"""CPU suite: pins the oracle (oracle/polar_oracle.c) against the golden vectors the UNMODIFIED
reference produced (tests/golden/, see make_golden.py) — tables, encoder, decode_scl_llr on seeded
and on edge-case inputs, and the reference's own deterministic get_bler_quick/main.cpp output."""
import ctypes as C

import numpy as np
import pytest

import golden_util as G
from oracle_lib import Oracle

libc = C.CDLL(None)


def _oracle(name):
    c, frozen, order, crcm = G.tables(name)
    libc.srand(1)
    if c.get("explicit_tables"):       # config 5: tables from the reference's MC-construction data file
        o = Oracle(c["n"], c["K"], 0.5, c["crc"])
        o.set_tables(frozen, order)
    else:
        o = Oracle(c["n"], c["K"], c["eps"], c["crc"])
    return c, o, frozen, order, crcm


def _inputs(o, cs):
    if "constellation" in cs:
        return o.synth_bicm_llr(cs["constellation"], G.seed(), cs["trial0"], cs["B"], cs["snr_db"])
    s = float.fromhex(cs["s_hex"])
    assert o.snr_sqrt_linear(cs["ebno"]) == s
    return o.synth_llr(G.seed(), cs["trial0"], cs["B"], s)


@pytest.mark.parametrize("name", [n for n in G.code_names() if not G.load()[1]["codes"][n].get("explicit_tables")])
def test_construction_matches_reference(oracle_built, name):
    c, o, frozen, order, crcm = _oracle(name)
    assert (o.frozen() == frozen).all()
    assert (o.order() == order).all(), "info order (libstdc++ introsort tie order) differs"
    assert (o.crc_matrix() == crcm).all(), "rand() CRC matrix differs"


@pytest.mark.parametrize("name", G.code_names())
def test_encode_matches_reference(oracle_built, name):
    c, o, *_ = _oracle(name)
    info, coded = G.enc_vectors(name)
    for i in range(info.shape[0]):
        assert (o.encode(info[i]) == coded[i]).all()


@pytest.mark.parametrize("name,ci", G.all_case_ids())
def test_decode_scl_llr_matches_reference(oracle_built, name, ci):
    c, o, *_ = _oracle(name)
    cs, want = list(G.cases(name))[ci]
    llr, sent = _inputs(o, cs)
    assert G.sha(llr) == cs["llr_sha256"], "synthetic workload generator drifted"
    got = o.decode_scl_llr(llr, cs["L"])
    assert (got == want).all()
    assert int((got != sent).any(axis=1).sum()) == cs["block_errors"]


@pytest.mark.parametrize("name", [n for n in G.code_names() if G.load()[1]["codes"][n]["specials"]])
def test_decode_edge_inputs_match_reference(oracle_built, name):
    c, o, *_ = _oracle(name)
    for sname, llr, exp in G.specials(name):
        for L, want in exp.items():
            got = o.decode_scl_llr(llr, L)
            assert (got == want).all(), f"{name}/{sname} L={L}"


def test_get_bler_quick_n9_matches_reference(oracle_built):
    z, m = G.load()
    p = m["bler_n9"]
    libc.srand(1)
    o = Oracle(p["n"], p["K"], p["eps"], p["crc"])
    got = o.get_bler_quick_ref(p["ebno"], p["L"])
    assert (got == z["bler_n9/bler"]).all()


def test_main_cpp_table_matches_reference(oracle_built):
    """The reference's own driver (PolarC/main.cpp) is deterministic; this pins construction,
    encoder, channel arithmetic, RNG restatement, decoder and the early-stop/skip logic jointly.
    BASELINE.md §2 quotes the same table (md5 0aed3bad...)."""
    z, m = G.load()
    p = m["main_cpp"]
    libc.srand(1)
    o = Oracle(p["n"], p["K"], p["eps"], p["crc"])
    got = o.get_bler_quick_ref(p["ebno"], p["L"])
    want = z["main_cpp/bler"]
    assert (got == want).all()
    # spot values printed in BASELINE.md (6 decimals)
    assert f"{want[0][0]:.6f}" == "0.711268" and f"{want[4][0]:.6f}" == "0.120669" and f"{want[0][4]:.6f}" == "0.052000"
