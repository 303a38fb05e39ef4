"""The MEX boundary, EXECUTED: polar_amd/matlab/polar_mex.cpp compiled against the working mx / mex runtime of
tests/mex_runtime/ (column-major mxArrays of MATLAB's classes, mexErrMsgIdAndTxt as an exception), linked against
libpolar_amd.so and driven command by command the way polar_amd/matlab/PolarCode.m drives it (tests/fake_matlab.py holds
that class re-expressed call for call). Reference surface: PolarM/PolarCode.m:59-93 (constructor), :266 encode, :290
decode_sc_p1, :299 decode_scl_p1, :312 decode_scl_llr, :781 get_bler_quick; caller PolarM/main.m:4-12.

CPU part: the gateway builds, links, creates / describes / destroys handles, validates arguments and refuses to compute
without a GPU. GPU part: every command against the golden vectors of the unmodified reference (tests/golden/), the oracle and
the numpy PolarM fixtures."""
import ctypes as C
import os

import numpy as np
import pytest

import golden_util as G

libc = C.CDLL(None)
FX = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "polarm_fixtures.npz"))


def _have_gpu():
    import torch
    return torch.cuda.is_available()


@pytest.fixture(scope="module")
def mex(built_lib):
    import fake_matlab
    return fake_matlab.polar_mex()


def _explicit(mex, name):
    c, frozen, order, crcm = G.tables(name)
    args = [float(c["n"]), float(c["K"]), float(c["crc"]), frozen.astype(np.uint8), order.astype(np.uint16)]
    if c["crc"]:
        args.append(crcm.astype(np.uint8))            # crc x K, column-major on the way in
    return c, frozen, order, crcm, mex('create_explicit', *args)


# ------------------------------------------------------------------------------------------------------------ CPU
def test_gateway_builds_links_and_manages_handles(mex):
    """'create' = the reference constructor (same tables as the golden ones under the same rand() stream), 'tables' returns
    them in MATLAB's layout (crc x K column-major), 'destroy' releases; mexLock depth follows the live handles."""
    import fake_matlab
    base_lock, base_live = mex.lock_depth(), mex.live_arrays()
    for name in ("small_n5_k16_crc4", "cfg3_n11_k1024_crc16"):
        c, frozen, order, crcm = G.tables(name)
        libc.srand(1)
        h = mex('create', float(c["n"]), float(c["K"]), c["eps"], float(c["crc"]))
        assert h.dtype == np.uint64 and h.shape == (1, 1) and mex.lock_depth() == base_lock + 1
        fz, od, cm = mex('tables', h, nlhs=3)
        assert fz.shape == (1, 1 << c["n"]) and fz.dtype == np.uint8 and (fz[0] == frozen).all()
        assert od.dtype == np.uint16 and (od[0] == order).all()
        assert cm.shape == (c["crc"], c["K"]) and (cm == crcm).all()
        mex('destroy', h, nlhs=0)
        assert mex.lock_depth() == base_lock
        with pytest.raises(fake_matlab.MexError) as e:          # a stale handle is refused, not dereferenced
            mex('tables', h, nlhs=3)
        assert e.value.identifier == "polar_amd:handle"
    assert mex.live_arrays() == base_live                        # (the driver freed every array it made or was handed)


def test_gateway_explicit_tables_round_trip(mex):
    for name in G.code_names():
        c, frozen, order, crcm, h = _explicit(mex, name)
        fz, od, cm = mex('tables', h, nlhs=3)
        assert (fz[0] == frozen).all() and (od[0] == order).all() and (cm == crcm).all()
        mex('destroy', h, nlhs=0)


def test_gateway_argument_errors(mex):
    import fake_matlab
    E = fake_matlab.MexError
    with pytest.raises(E) as e:
        mex('no_such_command', np.uint64(0))
    assert e.value.identifier in ("polar_amd:handle", "polar_amd:cmd")
    with pytest.raises(E) as e:
        mex('create', 5.0, 16.0)                                  # too few arguments
    assert e.value.identifier == "polar_amd:usage"
    with pytest.raises(E) as e:
        mex('create', 0.0, 1.0, 0.32, 0.0)                        # the library's own validation, through polar_last_error
    assert e.value.identifier == "polar_amd:error" and "out of range" in e.value.message
    with pytest.raises(E) as e:
        mex('tables', 12345.0, nlhs=3)                            # a double is not a handle
    assert e.value.identifier == "polar_amd:handle"
    h = mex('create', 5.0, 16.0, 0.32, 4.0)
    try:
        with pytest.raises(E) as e:
            mex('unknown', h)
        assert e.value.identifier == "polar_amd:cmd"
        with pytest.raises(E) as e:
            mex('decode_scl_llr', h, np.zeros((5, 6)), 4.0)
        assert e.value.identifier == "polar_amd:size"
        with pytest.raises(E) as e:
            mex('decode_scl_llr', h, np.zeros((32, 32)), 4.0)     # square: rows or columns? the gateway does not guess
        assert e.value.identifier == "polar_amd:layout"
        with pytest.raises(E) as e:
            mex('decode_scl_llr', h, np.zeros((3, 32)), 4.0, 'cols')
        assert e.value.identifier == "polar_amd:size"
        with pytest.raises(E) as e:
            mex('decode_scl_llr', h, np.zeros((3, 32)), 4.0, 'diagonal')
        assert e.value.identifier == "polar_amd:layout"
        with pytest.raises(E) as e:
            mex('decode_scl_llr', h, np.zeros((1, 32), np.int32), 4.0)
        assert e.value.identifier == "polar_amd:type"
        with pytest.raises(E) as e:
            mex('encode', h, np.zeros(16))                        # PolarCode.m passes uint8(info_bits)
        assert e.value.identifier == "polar_amd:type"
        with pytest.raises(E) as e:
            mex('decode_scl_llr', h, np.zeros((1, 32)), 65.0)     # list size: the library's range check
        assert e.value.identifier == "polar_amd:error"
    finally:
        mex('destroy', h, nlhs=0)


@pytest.mark.skipif(_have_gpu(), reason="only meaningful on a box without a GPU")
def test_gateway_has_no_cpu_fallback(mex):
    import fake_matlab
    h = mex('create', 5.0, 16.0, 0.32, 4.0)
    try:
        for call in (lambda: mex('decode_scl_llr', h, np.zeros((1, 32)), 4.0),
                     lambda: mex('encode', h, np.zeros(16, np.uint8)),
                     lambda: mex('decode_sc_p1', h, np.full(32, 0.25)),
                     lambda: mex('get_bler_quick', h, np.array([1.0]), np.array([1], np.uint8), 10.0, 5.0, 1.0, nlhs=2)):
            with pytest.raises(fake_matlab.MexError) as e:
                call()
            assert e.value.identifier == "polar_amd:error" and ("no HIP device" in e.value.message or "no CPU" in e.value.message)
    finally:
        mex('destroy', h, nlhs=0)


# ------------------------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
@pytest.mark.parametrize("name", G.code_names())
def test_gateway_encode_and_decode_golden(mex, name):
    """'encode' and 'decode_scl_llr' (1 x N, B x N, N x B, double and single) against the reference's own outputs."""
    import polar_amd
    from oracle_lib import Oracle
    c, frozen, order, crcm, h = _explicit(mex, name)
    N, K = 1 << c["n"], c["K"]
    try:
        info, coded = G.enc_vectors(name)
        for i in (0, 7, 15):
            got = mex('encode', h, info[i].astype(np.uint8))
            assert got.shape == (1, N) and (got[0] == coded[i]).all()
        g = polar_amd.PolarCode.from_tables(c["n"], c["K"], c["crc"], frozen, order, crcm if c["crc"] else None)
        o = Oracle(c["n"], c["K"], 0.32, c["crc"])
        o.set_tables(frozen, order)
        if c["crc"]:
            o.set_crc_matrix(crcm)
        import torch
        for cs, want in list(G.cases(name))[:3]:
            B = min(cs["B"], 96)
            d_llr = torch.empty((cs["B"], N), dtype=torch.float64, device="cuda")
            if "constellation" in cs:
                g.synth_bicm_llr_dev(cs["constellation"], G.seed(), cs["trial0"], cs["B"], cs["snr_db"], d_llr.data_ptr())
            else:
                g.synth_llr_dev(G.seed(), cs["trial0"], cs["B"], float.fromhex(cs["s_hex"]), d_llr.data_ptr())
            torch.cuda.synchronize()
            llr = d_llr.cpu().numpy()
            assert G.sha(llr) == cs["llr_sha256"]
            llr, want = llr[:B], want[:B]
            L = float(cs["L"])
            u = mex('decode_scl_llr', h, llr[3], L)                             # 1 x N -> 1 x K (the reference's call)
            assert u.shape == (1, K) and u.dtype == np.uint8 and (u[0] == want[3]).all()
            u = mex('decode_scl_llr', h, llr[5].reshape(N, 1), L)               # a column vector is one codeword too
            assert u.shape == (1, K) and (u[0] == want[5]).all()
            u = mex('decode_scl_llr', h, llr, L)                                # B x N -> B x K
            assert u.shape == (B, K) and (u == want).all()
            u = mex('decode_scl_llr', h, np.ascontiguousarray(llr.T), L)        # N x B -> K x B, nothing transposed on the way
            assert u.shape == (K, B) and (u.T == want).all()
            assert (mex('decode_scl_llr', h, llr, L, 'rows') == want).all()
            assert (mex('decode_scl_llr', h, np.ascontiguousarray(llr.T), L, 'cols').T == want).all()
            # single precision at the boundary: the float is widened exactly on the device, so the answer is the reference's
            # on the widened values
            f = llr.astype(np.float32)
            want32 = o.decode_scl_llr(f.astype(np.float64), int(L))
            assert (mex('decode_scl_llr', h, f, L) == want32).all()
            assert (mex('decode_scl_llr', h, np.ascontiguousarray(f.T), L).T == want32).all()
        if N <= 1024:                                                           # the square case with explicit layouts
            llr = np.asarray(o.synth_llr(11, 0, N, o.snr_sqrt_linear(2.0))[0])
            want = o.decode_scl_llr(llr, 2)
            assert (mex('decode_scl_llr', h, llr, 2.0, 'rows') == want).all()
            assert (mex('decode_scl_llr', h, np.ascontiguousarray(llr.T), 2.0, 'cols').T == want).all()
    finally:
        mex('destroy', h, nlhs=0)


@pytest.mark.gpu
def test_gateway_probability_domain_decoders(mex, oracle_built):
    """'decode_scl_p1' (PolarCode.cpp:110-128) against the oracle, 'decode_sc_p1' (PolarCode.m:290-295) against the numpy fixture."""
    from oracle_lib import Oracle
    for name in ("small_n5_k16_crc4", "cfg1_n9_k256"):
        c, frozen, order, crcm, h = _explicit(mex, name)
        try:
            o = Oracle(c["n"], c["K"], 0.32, c["crc"])
            o.set_tables(frozen, order)
            if c["crc"]:
                o.set_crc_matrix(crcm)
            llr = o.synth_llr(5, 0, 6, o.snr_sqrt_linear(1.5))[0]
            p1 = 1.0 / (1.0 + np.exp(llr))
            for b in range(6):
                for L in (1, 4, 32):
                    got = mex('decode_scl_p1', h, p1[b], 1.0 - p1[b], float(L))
                    assert got.shape == (1, c["K"]) and (got[0] == o.decode_scl_p1(p1[b], 1.0 - p1[b], L)).all(), (name, b, L)
            q, u = FX[f"scp1/{name}/p1"], FX[f"scp1/{name}/u"]
            for b in (0, 1, 7, 23):
                got = mex('decode_sc_p1', h, q[b])
                assert got.dtype == np.float64 and got.shape == (1, c["K"]) and (got[0] == u[b][order[: c["K"]]]).all()
        finally:
            mex('destroy', h, nlhs=0)


@pytest.mark.gpu
def test_gateway_get_bler_quick_and_polarm_main(mex, oracle_built):
    """PolarM/main.m:4-12 re-expressed: PolarCode(N, K, epsilon, crc) then [bler, ber] = get_bler_quick(ebno_vec, list_size_vec),
    through the class of polar_amd/matlab/PolarCode.m (fake_matlab.PolarCodeM). The counters are the library's (== the oracle's
    Monte-Carlo engine, tests/test_gpu_montecarlo.py); here: the gateway passes every argument and transposes both outputs
    into PolarM's (ebno, list) indexing; with and without a device list, BPSK and the 16-ASK BICM axis."""
    import fake_matlab
    import polar_amd
    libc.srand(1)
    pc = fake_matlab.PolarCodeM(512, 256, 0.32, 8)                 # main.m:4-9 (its own N = 2048 takes longer than a test should)
    assert pc.frozen_bits.shape == (1, 512) and pc.info_bits.min() >= 1 and len(pc.info_bits) == 264
    libc.srand(1)
    g = polar_amd.PolarCode(9, 256, 0.32, 8)
    assert (pc.frozen_bits[0] == g.frozen_bits).all() and (pc.info_bits - 1 == g.channel_order_descending[:264]).all()
    assert (pc.crc_matrix == g.crc_matrix).all()
    ebno, Ls = np.array([1.0, 1.5, 2.0, 2.5]), np.array([1, 4, 32])
    bler, ber = pc.get_bler_quick(ebno, Ls, 4000, 50, 7)           # main.m:11-12
    want, wber = g.get_bler_quick(ebno, Ls, max_runs=4000, max_err=50, seed=7, return_ber=True)
    assert bler.shape == (4, 3) and (bler == want.T).all() and (ber == wber.T).all()
    assert (np.diff(bler, axis=0) <= 0.02).all() and bler[0, 0] > bler[0, 2]      # a BLER surface, not zeros
    b2, e2 = pc.get_bler_quick(ebno, Ls, 4000, 50, 7, devices=[0])               # sharded form, one device
    assert (b2 == bler).all() and (e2 == ber).all()
    # the info bits decoded through the class are the oracle's
    from oracle_lib import Oracle
    o = Oracle(9, 256, 0.32, 8)
    o.set_tables(g.frozen_bits, g.channel_order_descending)
    o.set_crc_matrix(g.crc_matrix)
    llr = o.synth_llr(3, 0, 4, o.snr_sqrt_linear(1.0))[0]
    for b in range(4):
        u = pc.decode_scl_llr(llr[b], 8)
        assert u.dtype == np.float64 and (u[0] == o.decode_scl_llr(llr[b], 8)).all()
    x = pc.encode(np.arange(256) % 2)
    assert (x[0] == o.encode((np.arange(256) % 2).astype(np.uint8))).all()
    pc.delete()
    # the BICM sweep of main_MC_CC_Comparison.m:44-119 through the same command (constellation id 3 = 16-ASK Gray)
    c, frozen, order, crcm, h = _explicit(mex, "cfg5_n10_k512_ask16")
    try:
        g5 = polar_amd.PolarCode.from_tables(c["n"], c["K"], 0, frozen, order)
        snr = np.array([11.0, 12.0, 13.0])
        b, e = mex('get_bler_quick', h, snr, np.array([1, 8], np.uint8), 2000.0, 50.0, 5.0, np.zeros((1, 0), np.int32), 3.0, nlhs=2)
        wb, we = g5.get_bler_quick(snr, [1, 8], max_runs=2000, max_err=50, seed=5, return_ber=True, constellation="ask16-gray")
        assert b.shape == (2, 3) and (b == wb).all() and (e == we).all() and b[0, 0] > b[0, 2]
    finally:
        mex('destroy', h, nlhs=0)


@pytest.mark.gpu
def test_gateway_monte_carlo_design(mex, tmp_path):
    """'mc_construction' and 'monte_carlo_design' (PolarCode.m:95-141: cache file in the reference's format, stable sort,
    explicit-table handle): counts == the numpy fixture / the library's own entry point; a second call reads the file."""
    import fake_matlab
    import polar_amd
    for key in ("6_4", "6_1", "7_3"):                # (BPSK from trial 0 with the default; 4-ASK and 16-ASK from a later first trial)
        n, cid, seed, trial0, runs = (int(x) for x in FX[f"mc/{key}/params"])
        snr = float(FX[f"mc/{key}/snr"][0])
        args = [float(n), float(cid), snr, float(seed), float(runs)] + ([float(trial0)] if trial0 else [])
        cnt = mex('mc_construction', *args)
        assert cnt.shape == (1 << n, 1) and cnt.dtype == np.float64 and (cnt[:, 0].astype(np.int64) == FX[f"mc/{key}/counts"]).all(), key
    libc.srand(1)
    pc = fake_matlab.PolarCodeM(1024, 512, 0.32, 0)
    est, path = pc.monte_carlo_code_construction(13, 2000, 'ask16-gray', 'bicm', 9, data_dir=str(tmp_path))
    assert os.path.basename(path) == "MC_block_length_1024_512_cc_method_monte-carlo_cc_param_13_ask16-gray_bicm_2000.txt"   # PolarCode.m:111-113
    counts = np.loadtxt(path).astype(np.uint64)
    want = polar_amd.mc_construction(10, 13.0, 2000, "ask16-gray", seed=9)
    assert (counts == want).all()
    order = np.argsort(counts, kind="stable")
    assert (pc.info_bits - 1 == order[:512]).all()
    fz = np.ones(1024)
    fz[order[:512]] = 0
    assert (pc.frozen_bits[0] == fz).all() and abs(est - counts[order[:512]].sum() / 2000.0) < 1e-12
    u = pc.decode_scl_llr(np.full(1024, 3.0), 8)                 # the new handle decodes (all-zero codeword)
    assert (u == 0).all()
    # second design call: the table comes from the file (make it recognisable)
    marked = counts.copy()
    marked[::2] += 100000
    open(path, "w").write("".join(f"{int(v)} \n" for v in marked))
    pc.monte_carlo_code_construction(13, 2000, 'ask16-gray', 'bicm', 9, data_dir=str(tmp_path))
    assert (pc.info_bits - 1 == np.argsort(marked, kind="stable")[:512]).all()
    pc.delete()
    # the reference's shipped 16-ASK table through the same path: the fixture's frozen set and order
    z, _ = G.load()
    c, frozen, order, crcm = G.tables("cfg5_n10_k512_ask16")
    pc2 = fake_matlab.PolarCodeM(1024, 512, 0.32, 0)
    pc2.cc_method, pc2.cc_parameter, pc2.cc_misc = 'monte-carlo', 13, 'ask16-gray_bicm_250000'
    f = tmp_path / ('MC_block_length_' + pc2.get_unique_string() + '.txt')
    f.write_text("".join(f"{int(v)} \n" for v in z["cfg5_n10_k512_ask16/counts"]))
    pc2.monte_carlo_code_construction(13, 250000, 'ask16-gray', 'bicm', 1, data_dir=str(tmp_path))
    assert (pc2.frozen_bits[0] == frozen).all() and (pc2.info_bits - 1 == order[:512]).all()
    pc2.delete()
