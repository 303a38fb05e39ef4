"""GPU parity: the HIP path (through the C-ABI) vs the CPU oracle on the same seeded inputs.
Bit-exact bar: decoded info bits, encoder output and synthetic LLRs must be identical."""
import numpy as np
import pytest

from golden_util import both_kernels, both_l1_kernels

pytestmark = pytest.mark.gpu


def _pair(n, K, crc, srand=1):
    import ctypes as C
    import polar_amd
    from oracle_lib import Oracle
    o = Oracle(n, K, 0.32, crc, srand=srand)
    C.CDLL(None).srand(C.c_uint(srand))
    g = polar_amd.PolarCode(n, K, 0.32, crc)
    return o, g


CODES = [(5, 16, 4), (8, 128, 0), (9, 256, 0), (10, 512, 8), (11, 1024, 0), (11, 1024, 16)]


@pytest.mark.parametrize("n,K,crc", CODES)
def test_tables_match_oracle(built_lib, oracle_built, n, K, crc):
    o, g = _pair(n, K, crc)
    assert (g.frozen_bits == o.frozen()).all()
    assert (g.channel_order_descending == o.order()).all()
    assert (g.bit_rev_order == o.bitrev()).all()
    assert (g.crc_matrix == o.crc_matrix()).all()


@pytest.mark.parametrize("n,K,crc", CODES)
def test_encode_matches_oracle(built_lib, oracle_built, n, K, crc):
    o, g = _pair(n, K, crc)
    rng = np.random.default_rng(n * 131 + crc)
    info = rng.integers(0, 2, (37, K)).astype(np.uint8)
    got = g.encode(info)
    for i in range(info.shape[0]):
        assert (got[i] == o.encode(info[i])).all()


@pytest.mark.parametrize("n,K,crc", CODES)
@pytest.mark.parametrize("L", [1, 2, 4, 8, 32])
def test_decode_scl_llr_matches_oracle(built_lib, oracle_built, n, K, crc, L):
    o, g = _pair(n, K, crc)
    B = 48 if n >= 11 else 96
    for ebno in (1.0, 2.0):
        llr, info = o.synth_llr(1234 + L, 0, B, o.snr_sqrt_linear(ebno))
        want = o.decode_scl_llr(llr, L)
        got = both_kernels(g, lambda: g.decode_scl_llr(llr, L)) if L <= 8 else g.decode_scl_llr(llr, L)
        bad = np.nonzero((want != got).any(axis=1))[0]
        assert bad.size == 0, f"{bad.size}/{B} codewords differ (first {bad[:5]}) n={n} K={K} crc={crc} L={L} ebno={ebno}"


@pytest.mark.parametrize("L", [3, 5, 6, 12, 20, 33, 64])
def test_odd_list_sizes_match_oracle(built_lib, oracle_built, L):
    """The reference accepts any list size (not only powers of two)."""
    o, g = _pair(9, 256, 8)
    llr, _ = o.synth_llr(4321, 50, 40, o.snr_sqrt_linear(1.0))
    got = both_kernels(g, lambda: g.decode_scl_llr(llr, L)) if L <= 8 else g.decode_scl_llr(llr, L)
    assert (o.decode_scl_llr(llr, L) == got).all()


@pytest.mark.parametrize("L", [17, 24, 31])
def test_list_sizes_17_to_31_on_the_headline_code(built_lib, oracle_built, L):
    """Lists of 17..31 paths run in groups of 32 lanes with idle lanes: table mode (N >= 1024) and the scalar-unit fork
    ranking see a list that is "full" below 32. Noisy rows, plus all-zero and +-1 rows (every metric ties)."""
    o, g = _pair(11, 1024, 16)
    llr, _ = o.synth_llr(2718, 0, 160, o.snr_sqrt_linear(1.5))
    llr[7] = 0.0
    llr[8] = np.where(np.arange(2048) % 3 == 0, 1.0, -1.0)
    want = o.decode_scl_llr(llr, L)
    got = g.decode_scl_llr(llr, L)
    bad = np.nonzero((want != got).any(axis=1))[0]
    assert bad.size == 0, f"{bad.size}/160 codewords differ (first {bad[:5]}) L={L}"


def test_ragged_batches_and_empty(built_lib, oracle_built):
    o, g = _pair(8, 128, 4)
    llr, _ = o.synth_llr(9, 0, 131, o.snr_sqrt_linear(1.5))
    want = o.decode_scl_llr(llr, 8)
    for B in (1, 2, 3, 7, 9, 63, 65, 131):       # not multiples of the codewords-per-wave
        assert (both_kernels(g, lambda: g.decode_scl_llr(llr[:B], 8)) == want[:B]).all()
    assert g.decode_scl_llr(llr[:0], 8).shape == (0, 128)
    # single-vector form
    assert (g.decode_scl_llr(llr[5], 8) == want[5]).all()


@pytest.mark.parametrize("L", [1, 2, 4, 32])
def test_pipelined_host_batches_at_chunk_boundaries(built_lib, oracle_built, L):
    """Large host-pointer batches are cut into chunks (pinned staging, copy stream, several decode lanes: polar_hostpipe.cpp
    host_decode_pipelined). Forced here on small batches with tiny chunks: a batch that is one codeword short of / exactly /
    one codeword over a whole number of chunks, fewer chunks than ring slots and many more, one / two / five lanes, doubles and
    floats — every row must be what the unpipelined path returns (== the oracle), in its place."""
    o, g = _pair(8, 128, 4)
    llr, _ = o.synth_llr(11, 0, 1000, o.snr_sqrt_linear(1.5))
    want = o.decode_scl_llr(llr, L)
    row = 256 * 8
    g.debug_set("host_pipe_min_bytes", 1)
    g.debug_set("host_ramp", -1)                 # equal chunks: their number is then ceil(B / chunk)
    for lanes in (1, 2, 5):
        g.debug_set("host_lanes", lanes)
        for chunk_cw, Bs in ((8, (9, 15, 16, 17, 31, 33, 100)), (64, (65, 127, 128, 129, 448, 1000)), (200, (201, 799, 1000))):
            g.debug_set("host_chunk_bytes", chunk_cw * row)
            for B in Bs:
                got = g.decode_scl_llr(llr[:B], L)
                assert g.debug_get("host_chunks") == -(-B // chunk_cw), (B, chunk_cw, g.debug_get("host_chunks"))
                assert g.debug_get("host_lanes") == min(lanes, -(-B // chunk_cw))
                assert (got == want[:B]).all(), (L, lanes, chunk_cw, B)
    # the default schedule: small first chunks (an eighth of the full size, doubling), then equal full-size ones
    g.debug_set("host_ramp", 0)
    g.debug_set("host_lanes", 3)
    for chunk_cw, Bs in ((64, (65, 100, 129, 500, 1000)), (512, (513, 1000))):
        g.debug_set("host_chunk_bytes", chunk_cw * row)
        for B in Bs:
            got = g.decode_scl_llr(llr[:B], L)
            assert g.debug_get("host_chunks") >= 2 and g.debug_get("host_chunk_cw") <= chunk_cw
            assert (got == want[:B]).all(), (L, "ramp", chunk_cw, B)
    g.debug_set("host_ramp", -1)
    g.debug_set("host_lanes", 2)
    # floats: half the bytes per row, the chunk is then twice the codewords
    g.debug_set("host_chunk_bytes", 64 * row)
    l32 = llr.astype(np.float32)
    want32 = o.decode_scl_llr(l32.astype(np.float64), L)
    got = g.decode_scl_llr(l32[:777], L)
    assert g.debug_get("host_chunks") == -(-777 // 128) and (got == want32[:777]).all()
    # a batch that fits one chunk is not pipelined; and the knob -1 switches the pipeline off whatever the size
    got = g.decode_scl_llr(llr[:64], L)
    assert g.debug_get("host_chunks") == 0 and (got == want[:64]).all()
    g.debug_set("host_pipe_min_bytes", -1)
    got = g.decode_scl_llr(llr, L)
    assert g.debug_get("host_chunks") == 0 and (got == want).all()
    # a setter between two pipelined calls drops the second lane's copy of the tables: the next call rebuilds it
    g.debug_set("host_pipe_min_bytes", 1)
    g.debug_set("host_lanes", 2)
    g.set_mode(1)
    assert (g.decode_scl_llr(llr, L) == want).all() and g.debug_get("host_chunks") == -(-1000 // 64)
    g.set_mode(0)
    assert (g.decode_scl_llr(llr, L) == want).all()


def test_pipelined_host_batch_at_the_default_settings(built_lib):
    """The defaults (no knob): 16 384 codewords of the headline code at list size 1 (256 MiB of doubles) and 32 768 at list
    size 4 (512 MiB) through the host-pointer entry point == the device-resident decode of the same rows; 16 384 codewords at
    list size 4 are BELOW the pipeline's threshold for the list kernels (round 6: one copy in, one launch, one copy out is
    faster there); a CRC matrix set in between reaches every decode lane."""
    import ctypes as C
    import polar_amd
    import torch
    C.CDLL(None).srand(C.c_uint(1))
    g = polar_amd.PolarCode(11, 1024, 0.32, 16)
    B = 32768
    d_llr = torch.empty((B, 2048), dtype=torch.float64, device="cuda")
    d_out = torch.empty((B, 1024), dtype=torch.uint8, device="cuda")
    g.synth_llr_dev(5, 0, B, g.snr_sqrt_linear(2.0), d_llr.data_ptr())
    llr = d_llr.cpu().numpy()
    for L, Bl in ((1, 16384), (4, 32768)):
        g.decode_scl_llr_dev(d_llr.data_ptr(), Bl, L, d_out.data_ptr())
        torch.cuda.synchronize()
        got = g.decode_scl_llr(llr[:Bl], L)
        # (L = 1, 256 MiB: full-size chunks of 64 MiB, i.e. 4096 codewords; 512 + 1024 + 2048 first, then 4 x 3200.
        #  L = 4, 512 MiB: chunks of 128 MiB, i.e. 8192 codewords; 1024 + 2048 + 4096 first, then 4 x 6400)
        assert g.debug_get("host_chunks") == 7 and g.debug_get("host_lanes") == (2 if L == 1 else 3)
        assert (got == d_out[:Bl].cpu().numpy()).all()
    B = 16384
    g.decode_scl_llr_dev(d_llr.data_ptr(), B, 4, d_out.data_ptr())
    torch.cuda.synchronize()
    got = g.decode_scl_llr(llr[:B], 4)
    assert g.debug_get("host_chunks") == 0 and (got == d_out[:B].cpu().numpy()).all()
    m = g.crc_matrix
    g.crc_matrix = m[::-1].copy()
    B = 32768                                                              # (pipelined again: the lanes' private tables)
    g.decode_scl_llr_dev(d_llr.data_ptr(), B, 4, d_out.data_ptr())
    torch.cuda.synchronize()
    assert (g.decode_scl_llr(llr, 4) == d_out.cpu().numpy()).all() and g.debug_get("host_chunks") == 7


@pytest.mark.parametrize("lds_log", [3, 4, 5])
def test_tuning_knobs_do_not_change_results(built_lib, oracle_built, lds_log):
    o, g = _pair(11, 1024, 16)
    llr, _ = o.synth_llr(77, 0, 16, o.snr_sqrt_linear(1.5))
    g.set_tuning(waves_per_cu=4, lds_log=lds_log)
    assert (g.decode_scl_llr(llr, 32) == o.decode_scl_llr(llr, 32)).all()


def test_device_synth_is_bit_identical_to_oracle(built_lib, oracle_built):
    import torch
    o, g = _pair(10, 512, 8)
    for ebno, t0 in ((0.0, 0), (2.5, 12345678901)):
        s = o.snr_sqrt_linear(ebno)
        llr, info = o.synth_llr(31337, t0, 300, s)
        d_llr = torch.empty((300, 1024), dtype=torch.float64, device="cuda")
        d_info = torch.empty((300, 512), dtype=torch.uint8, device="cuda")
        g.synth_llr_dev(31337, t0, 300, s, d_llr.data_ptr(), d_info.data_ptr())
        torch.cuda.synchronize()
        assert (d_llr.cpu().numpy() == llr).all()
        assert (d_info.cpu().numpy() == info).all()


def test_monte_carlo_counters_match_oracle(built_lib, oracle_built):
    """polar_mc_batch (GPU) vs the oracle's orc_mc_batch: identical error/run counters, including
    the 'decoded at a lower Eb/N0 => counted, not simulated' rule and disabled points."""
    o, g = _pair(8, 128, 4)
    ebno, Ls = [0.5, 1.5, 2.5, 3.5], [1, 4, 16]
    en = np.ones((3, 4), np.uint8)
    en[1, 2] = 0
    e1, r1 = np.zeros((3, 4), np.uint64), np.zeros((3, 4), np.uint64)
    e2, r2 = np.zeros((3, 4), np.uint64), np.zeros((3, 4), np.uint64)
    o.mc_batch(5, 3, 150, 2, ebno, Ls, en, e1, r1)
    g.mc_batch(5, 3, 150, 2, ebno, Ls, en, e2, r2)
    assert (e1 == e2).all() and (r1 == r2).all()
    assert e1.sum() > 0
    # get_bler_quick = batches + early stop; batch-size independent when nothing stops early
    b1 = g.get_bler_quick(ebno, Ls, max_runs=120, max_err=10**9, seed=5, batch=40)
    b2 = g.get_bler_quick(ebno, Ls, max_runs=120, max_err=10**9, seed=5, batch=120)
    assert (b1 == b2).all()


def test_full_size_properties_config2(built_lib):
    """BASELINE config 2 at full size (N=2048, K=1024, L=1, batch 65536): size-independent
    properties — noiseless round trip encode -> BPSK LLR -> decode == info, and decode of the
    device-generated 2 dB batch agrees with the sent bits except for a plausible BLER."""
    import torch
    import polar_amd
    g = polar_amd.PolarCode(11, 1024, 0.32, 0)
    B, N, K = 65536, 2048, 1024
    gen = torch.Generator(device="cuda").manual_seed(1)
    info = torch.randint(0, 2, (B, K), dtype=torch.uint8, device="cuda", generator=gen)
    coded = torch.empty((B, N), dtype=torch.uint8, device="cuda")
    g.encode_dev(info.data_ptr(), B, coded.data_ptr())
    llr = (1.0 - 2.0 * coded.to(torch.float64)) * 6.0        # bit 0 -> +6, bit 1 -> -6
    out = torch.empty((B, K), dtype=torch.uint8, device="cuda")
    g.decode_scl_llr_dev(llr.data_ptr(), B, 1, out.data_ptr())
    torch.cuda.synchronize()
    assert torch.equal(out, info)
    # linearity of the encoder: enc(a ^ b) = enc(a) ^ enc(b)
    c2 = torch.empty_like(coded)
    g.encode_dev((info ^ info.roll(1, 0)).contiguous().data_ptr(), B, c2.data_ptr())
    torch.cuda.synchronize()
    assert torch.equal(c2, coded ^ coded.roll(1, 0))
    # noisy batch: BLER at 2 dB for SC is ~0.05 (results/polar_performance.jpeg; golden table 0.052)
    sent = torch.empty((B, K), dtype=torch.uint8, device="cuda")
    g.synth_llr_dev(99, 0, B, g.snr_sqrt_linear(2.0), llr.data_ptr(), sent.data_ptr())
    g.decode_scl_llr_dev(llr.data_ptr(), B, 1, out.data_ptr())
    torch.cuda.synchronize()
    bler = (out != sent).any(dim=1).double().mean().item()
    assert 0.035 < bler < 0.07, bler


def test_full_size_roundtrip_config4(built_lib):
    """N=2048 K=1024 crc16 L=32: noiseless round trip and erasure-like robustness on 4096 codewords."""
    import ctypes as C
    import torch
    import polar_amd
    C.CDLL(None).srand(C.c_uint(1))
    g = polar_amd.PolarCode(11, 1024, 0.32, 16)
    B, N, K = 4096, 2048, 1024
    gen = torch.Generator(device="cuda").manual_seed(2)
    info = torch.randint(0, 2, (B, K), dtype=torch.uint8, device="cuda", generator=gen)
    coded = torch.empty((B, N), dtype=torch.uint8, device="cuda")
    g.encode_dev(info.data_ptr(), B, coded.data_ptr())
    llr = (1.0 - 2.0 * coded.to(torch.float64)) * 4.0
    er = torch.rand((B, N), device="cuda", generator=gen) < 0.2      # 20% erasures (llr = 0)
    llr = torch.where(er, torch.zeros_like(llr), llr).contiguous()
    out = torch.empty((B, K), dtype=torch.uint8, device="cuda")
    g.decode_scl_llr_dev(llr.data_ptr(), B, 32, out.data_ptr())
    torch.cuda.synchronize()
    assert torch.equal(out, info)


@pytest.mark.parametrize("n,K,crc", [(5, 16, 4), (8, 128, 0), (10, 512, 8), (11, 1024, 16)])
@pytest.mark.parametrize("L", [1, 4, 8, 32])
def test_decode_scl_p1_matches_oracle(built_lib, oracle_built, n, K, crc, L):
    """Probability-domain SCL (PolarCode.cpp:110-128, 375-420) incl. the cross-path normalisation; n = 11 is the headline
    code (N = 2048, K = 1024, CRC 16) in the probability domain."""
    o, g = _pair(n, K, crc)
    llr, _ = o.synth_llr(2222, 0, 24 if n < 11 else 12, o.snr_sqrt_linear(2.0))
    p1 = 1.0 / (1.0 + np.exp(llr))
    p0 = 1.0 - p1
    got = g.decode_scl_p1(p1, p0, L)
    for i in range(llr.shape[0]):
        assert (got[i] == o.decode_scl_p1(p1[i], p0[i], L)).all(), (i, L)


def test_decode_sc_p1_matches_oracle(built_lib, oracle_built):
    """PolarM decode_sc_p1 (cnop/vnop recursion): bitwise-equal doubles, incl. the y == 0.5 -> 0.5 quirk."""
    o, g = _pair(9, 256, 0)
    llr, _ = o.synth_llr(3333, 0, 70, o.snr_sqrt_linear(2.0))
    p1 = 1.0 / (1.0 + np.exp(llr))
    p1[3, ::7] = 0.5
    p1[4, :] = 0.5
    got = both_kernels(g, lambda: g.decode_sc_p1(p1))        # one lane per codeword / one codeword per wave (round 6): the same doubles
    for i in range(p1.shape[0]):
        assert (got[i] == o.decode_sc_p1(p1[i])).all(), i
    # agrees with the LLR decoder at L = 1 on ordinary inputs (SURVEY 8c cross-check)
    assert (got[10:] == g.decode_scl_llr(llr[10:], 1)).all()


def _cpu_reference(n, K, crc, frozen=None, order=None):
    """The unmodified reference build when it travelled with the snapshot (oracle/_ref), else the pinned restatement."""
    import oracle_lib
    cls = oracle_lib.Reference if oracle_lib.have_reference() else oracle_lib.Oracle
    c = cls(n, K, 0.32, crc, srand=1)
    if frozen is not None:
        c.set_tables(frozen, order)
    return c


@pytest.mark.parametrize("ebno", [1.0, 2.5])
def test_decode_sc_p1_equals_the_reference_decode_scl_llr_at_list_size_1(built_lib, oracle_built, ebno):
    """SURVEY 8c cross-check for the PolarM-only row a18, against the REFERENCE itself (round 5 compared with this library's own
    L = 1 decoder): PolarM's decode_sc_p1 on p1 = 1 / (1 + e^llr) returns, for every row that holds no exact tie, the bits
    PolarC's decode_scl_llr(llr, 1) returns on llr — 4096 rows per SNR, decode failures included. (The probability-domain
    recursion is another arithmetic than the LLR one: agreement is on DECISIONS; rows where the two CPU sides themselves
    disagree — the restatement of decode_sc_p1 against the reference's LLR decoder — are counted and must be rare, and the
    device must side with the decode_sc_p1 restatement there.)"""
    o, g = _pair(10, 512, 0)
    ref = _cpu_reference(10, 512, 0)
    B = 4096
    llr, _ = o.synth_llr(4242, 0, B, o.snr_sqrt_linear(ebno))
    p1 = 1.0 / (1.0 + np.exp(llr))
    got = g.decode_sc_p1(p1)
    want = ref.decode_scl_llr(llr, 1)
    differ = np.nonzero((got != want).any(axis=1))[0]
    assert set(np.unique(got)) <= {0.0, 1.0}
    assert differ.size <= 2, (ebno, differ[:10])                 # (a leaf within rounding of 0.5 may go either way: none seen)
    for i in differ:
        assert (got[i] == o.decode_sc_p1(p1[i])).all(), i
    assert (got == want).all(axis=1).mean() > 0.999


def test_decode_sc_p1_on_16ask_bicm_equals_the_reference_at_list_size_1(built_lib, oracle_built):
    """The same cross-check on BASELINE configuration 5's own path (main_MC_CC_Comparison.m:90-96: the 16-ASK demapper's p1 goes
    into decode_sc_p1): device BICM LLRs and p1 = 1 / (1 + e^llr) at two SNRs, the reference's shipped construction table;
    device decode_sc_p1(p1) == unmodified PolarC decode_scl_llr(llr, 1), and device decode_scl_llr(llr, 8) == PolarC's at L = 8."""
    import torch
    import polar_amd
    import golden_util as G
    c, frozen, order, crcm = G.tables("cfg5_n10_k512_ask16")
    g = polar_amd.PolarCode.from_tables(10, 512, 0, frozen, order)
    ref = _cpu_reference(10, 512, 0, frozen, order)
    for snr, B in ((11.0, 2048), (13.0, 2048)):
        d = torch.empty((B, 1024), dtype=torch.float64, device="cuda")
        g.synth_bicm_llr_dev("ask16-gray", 99, 0, B, snr, d.data_ptr())
        torch.cuda.synchronize()
        llr = d.cpu().numpy()
        want1 = ref.decode_scl_llr(llr, 1)
        got = g.decode_sc_p1(1.0 / (1.0 + np.exp(llr)))
        assert ((got != want1).any(axis=1)).sum() <= 1, snr
        assert (g.decode_scl_llr(llr, 1) == want1).all()
        assert (g.decode_scl_llr(llr[:512], 8) == ref.decode_scl_llr(llr[:512], 8)).all()


@pytest.mark.parametrize("n,K,crc", [(9, 256, 0), (10, 512, 8), (11, 1024, 16)])
@pytest.mark.parametrize("L", [1, 4, 32])
def test_winning_path_metric_matches_oracle(built_lib, oracle_built, n, K, crc, L):
    """Floating-point side of the parity: the winning path metric (PolarCode.cpp:483,580-601 sums of
    log(1+exp(.))) agrees with the oracle's fp64/glibc value to 1e-10 relative (tolerance for libm
    vs the kernel's table-driven exp/log1p, ~1e-16 per term over <= N terms); the bits are exact."""
    import torch
    o, g = _pair(n, K, crc)
    B = 24
    llr, _ = o.synth_llr(808, 0, B, o.snr_sqrt_linear(1.5))
    d = torch.tensor(llr, device="cuda")
    out = torch.empty((B, K), dtype=torch.uint8, device="cuda")
    pm = torch.zeros(B, dtype=torch.float64, device="cuda")
    for lat in ((-1, 1 << 40) if L <= 8 else (0,)):           # the batch kernel, the one-codeword-per-wave kernel (list sizes 2 .. 8)
        g.debug_set("lat_max_b", lat)
        out.zero_(); pm.zero_()
        g.decode_scl_llr_dev(d.data_ptr(), B, L, out.data_ptr(), pm.data_ptr())
        torch.cuda.synchronize()
        got = pm.cpu().numpy()
        for i in range(B):
            bits, want = o.decode_scl_llr_pm(llr[i], L)
            assert (out[i].cpu().numpy() == bits).all()
            assert abs(got[i] - want) <= 1e-10 * max(1.0, abs(want)), (i, got[i], want)
    g.debug_set("lat_max_b", 0)


def test_no_finite_candidate_and_a_list_that_never_filled(built_lib, oracle_built):
    """Every path meets a frozen leaf with llr < -709.78 (path metric +inf) and the list has more entries than 2^K paths:
    the reference returns l_p = 0, a path that was never activated — the zeros of initializeDataStructures
    (PolarCode.cpp:195-230, 609-644). Found by tools/fuzz_parity.py."""
    o, g = _pair(4, 3, 0)
    llr = np.zeros((4, 16))
    llr[0] = np.where(np.arange(16) % 2 == 0, 1e3, -1e3)
    llr[1], llr[2], llr[3] = 0.5 * llr[0], 0.7 * llr[0], 0.8 * llr[0]
    for L in (1, 2, 4, 8, 16, 24, 32, 64):
        want = o.decode_scl_llr(llr, L)
        for mode in (0, 1):
            g.set_mode(mode)
            assert (g.decode_scl_llr(llr, L) == want).all(), (L, mode)


@pytest.mark.parametrize("n,F", [(9, 256), (10, 64), (10, 256), (11, 128), (11, 256)])
def test_list_size_one_unfrozen_leaves_in_the_worst_channels(built_lib, oracle_built, n, F):
    """Explicit tables no construction produces: the first F leaves frozen, every other leaf unfrozen. The first unfrozen
    leaves then carry LLRs far below the rounding granularity of the path metric, where the reference's decision is the
    tie-break (bit 0) of PolarCode.cpp:505-553, not the sign: the L = 1 kernel hands such codewords to the general kernel
    (found by tools/fuzz_parity.py: 43 % mismatching codewords at n = 11, F = 128 before the guard). Where the general
    kernel is exact on these inputs (it is at 8 dB; at 3 dB a stray codeword may remain — printed, not asserted: the
    reference's value at such a leaf is its own rounding noise) the L = 1 kernel must be too; and at both SNRs the L = 1
    kernel must return exactly what the LLR-domain kernel returns."""
    import polar_amd
    N = 1 << n
    frozen = np.zeros(N, np.uint8); frozen[:F] = 1
    K = N - F
    order = np.concatenate([np.nonzero(frozen == 0)[0], np.nonzero(frozen)[0]]).astype(np.uint16)
    o, _ = _pair(n, K, 0)
    o.set_tables(frozen, order)
    g = polar_amd.PolarCode.from_tables(n, K, 0, frozen, order)
    for ebno in (8.0, 3.0):
        llr, _ = o.synth_llr(77, 0, 1000, o.snr_sqrt_linear(ebno))
        want = o.decode_scl_llr(llr, 1)
        g.set_mode(0); sc_out = both_l1_kernels(g, lambda: g.decode_scl_llr(llr, 1))
        g.set_mode(1); gen_out = g.decode_scl_llr(llr, 1)
        # the L = 1 kernel against what the LLR-domain kernel produces: identical, codeword for codeword, at both SNRs
        assert (sc_out == gen_out).all(), (ebno, int((sc_out != gen_out).any(axis=1).sum()))
        gen = int((want != gen_out).any(axis=1).sum())
        print(f"n={n} F={F} Eb/N0={ebno}: LLR-domain kernel vs reference: {gen} differing codewords of 1000")
        # (a loose absolute bound against the reference, so that a drift of the LLR-domain kernel on such codes does not pass
        # unnoticed behind the kernel-vs-kernel comparison: measured 0 of 1000 at 8 dB on all five codes, 0 ... 1 at 3 dB)
        assert gen == 0 if ebno == 8.0 else gen <= 4


@pytest.mark.parametrize("n,K,crc", [(1, 1, 0), (2, 2, 0), (2, 3, 0), (3, 4, 0), (3, 5, 1), (3, 8, 0), (4, 8, 0), (4, 11, 2), (4, 1, 0)])
def test_tiny_block_lengths(built_lib, oracle_built, n, K, crc):
    """N = 2 .. 16: fewer elements than a row of the L = 1 kernel holds, all-unfrozen and single-bit codes."""
    o, g = _pair(n, K, crc)
    llr, _ = o.synth_llr(5, 0, 300, o.snr_sqrt_linear(1.0))
    for L in (1, 2, 4, 8, 32):
        got = both_kernels(g, lambda: g.decode_scl_llr(llr, L)) if L <= 8 else g.decode_scl_llr(llr, L)
        assert (o.decode_scl_llr(llr, L) == got).all(), L


@pytest.mark.parametrize("n,K,crc,B", [(13, 4096, 0, 40), (14, 8192, 24, 24), (15, 16384, 0, 11), (12, 3000, 8, 64), (6, 40, 0, 100)])
def test_list_size_one_long_codes(built_lib, oracle_built, n, K, crc, B):
    """L = 1 (pruned SC kernel) beyond the sizes whose channel row is permuted through LDS (n > 12: scattered stores of the
    conversion kernel), with the deepest fused F-chains, ragged batches (not multiples of the eight codewords per wave)."""
    o, g = _pair(n, K, crc)
    llr, _ = o.synth_llr(707, 0, B, o.snr_sqrt_linear(1.5))
    assert (o.decode_scl_llr(llr, 1) == both_l1_kernels(g, lambda: g.decode_scl_llr(llr, 1))).all()   # (n > 12: both runs take the eight-codeword kernel)


@pytest.mark.parametrize("n,K,crc,L,B", [(12, 2048, 16, 8, 24), (13, 4096, 0, 4, 16), (14, 8192, 24, 2, 8),
                                         (15, 16384, 32, 32, 4), (11, 1024, 32, 64, 12), (7, 100, 7, 16, 64)])
def test_long_codes_and_extreme_parameters(built_lib, oracle_built, n, K, crc, L, B):
    """Maximum sizes of the class surface: N up to 32768 (uint16_t block length of the reference),
    L = 64, crc = 32, K not a power of two."""
    o, g = _pair(n, K, crc)
    llr, _ = o.synth_llr(606, 0, B, o.snr_sqrt_linear(1.0))
    want = o.decode_scl_llr(llr, L)
    got = g.decode_scl_llr(llr, L)
    assert (want == got).all()


@pytest.mark.parametrize("n,K,crc", [(9, 256, 8), (10, 512, 0), (11, 1024, 16), (13, 4096, 0)])
def test_list_size_one_reads_the_callers_rows_in_place(built_lib, oracle_built, n, K, crc, monkeypatch):
    """From N = 512 on the two top-layer visits of the L = 1 kernel read the caller's rows where they lie (bit-reversed
    64-byte chunks, converted on the fly; the G visit adds in the LLR domain) instead of a permuted, converted copy made
    by a front pass. Both paths must give the reference's bits, including rows the input guard sends to the general kernel."""
    import polar_amd
    o, g = _pair(n, K, crc)
    B = 203                                      # ragged: not a multiple of the eight codewords per wave
    llr, _ = o.synth_llr(818, 0, B, o.snr_sqrt_linear(1.0))
    llr[3, 5] = 0.0
    llr[4, 100] = np.inf
    llr[5] *= 1e-12
    llr[6, : 1 << (n - 1)] = 800.0               # L-form values through the top G visit
    llr[7] = -llr[7]
    want = o.decode_scl_llr(llr, 1)
    got = both_l1_kernels(g, lambda: g.decode_scl_llr(llr, 1))
    g.debug_set("sc_no_fold", 1)
    got_front = both_l1_kernels(g, lambda: g.decode_scl_llr(llr, 1))
    g.debug_set("sc_no_fold", 0)
    assert (got == want).all() and (got_front == want).all()
    f = llr.astype(np.float32)
    want32 = o.decode_scl_llr(f.astype(np.float64), 1)
    assert (both_l1_kernels(g, lambda: g.decode_scl_llr(f, 1)) == want32).all()


@pytest.mark.parametrize("n,K,crc,L", [(9, 256, 8, 8), (11, 1024, 16, 32), (6, 20, 3, 1), (11, 1024, 16, 1)])
def test_float32_llr_boundary(built_lib, oracle_built, n, K, crc, L):
    """the boundary also takes single-precision LLRs; they are widened exactly, so the
    result is the reference's decode of (double)llr."""
    import torch
    o, g = _pair(n, K, crc)
    B = 40
    llr, _ = o.synth_llr(515, 0, B, o.snr_sqrt_linear(1.5))
    f = llr.astype(np.float32)
    f[0, :4] = [0.0, -0.0, np.float32(1e-30), np.float32(-3e38)]
    want = o.decode_scl_llr(f.astype(np.float64), L)
    assert ((both_kernels(g, lambda: g.decode_scl_llr(f, L)) if L <= 8 else g.decode_scl_llr(f, L)) == want).all()
    d = torch.tensor(f, device="cuda")
    out = torch.empty((B, K), dtype=torch.uint8, device="cuda")
    g.decode_scl_llr_dev_f32(d.data_ptr(), B, L, out.data_ptr())
    torch.cuda.synchronize()
    assert (out.cpu().numpy() == want).all()


@pytest.mark.parametrize("n,K,crc,L,B", [(6, 30, 4, 32, 3 * 8192 + 17), (5, 16, 0, 8, 2 * 32768 + 5), (7, 64, 8, 64, 2 * 4096 + 3)])
def test_more_codewords_than_resident_waves(built_lib, oracle_built, n, K, crc, L, B):
    """Batches larger than one round of the persistent grid (256 CUs x 16 waves x 64/GS codewords):
    the groups after a wave's first one are handed out by a device counter, in no fixed order — every
    codeword must still be decoded exactly once, bit-exactly, at its own output position."""
    o, g = _pair(n, K, crc)
    llr, _ = o.synth_llr(4242, 0, B, o.snr_sqrt_linear(1.0))
    got = g.decode_scl_llr(llr, L)
    want = o.decode_scl_llr(llr, L)
    bad = np.nonzero((want != got).any(axis=1))[0]
    assert bad.size == 0, (bad.size, bad[:8])
    assert (g.decode_scl_llr(llr, L) == got).all()        # and reproducibly so


@pytest.mark.parametrize("L", [1, 4, 8, 32])
def test_degenerate_rows_mixed_with_normal_ones(built_lib, oracle_built, L):
    """One batch that mixes ordinary noisy codewords with rows the fast kernels must hand to the general kernel
    (zeros, infinite LLRs of known bits, -inf runs, +-1000 saturated, exact +-1 ties, sub-1e-12 noise, the |llr| = 40 boundary): every row — flagged
    or not, whichever wave it shares with which neighbour — must equal the oracle's decode of that row."""
    o, g = _pair(9, 256, 8)
    N = 512
    rng = np.random.default_rng(99 + L)
    llr, _ = o.synth_llr(2024, 0, 384, o.snr_sqrt_linear(1.5))
    sgn = rng.choice([-1.0, 1.0], (16, N))
    # "known bits": a noisy codeword with 5 % of its positions at +-inf of the CORRECT sign (infinities of conflicting
    # sign meet as inf - inf = NaN in the reference's g-node, PolarCode.cpp:449: outside any meaningful domain)
    coded = o.encode(rng.integers(0, 2, 256).astype(np.uint8)).astype(float)
    known = (1.0 - 2.0 * coded) * np.where(rng.random(N) < 0.05, np.inf, np.abs(rng.normal(3.0, 1.5, N)) + 0.5)
    special = [np.zeros(N), 1000.0 * sgn[0], sgn[1], rng.normal(0, 1e-12, N), sgn[2] * rng.choice([0.5, 39.999, 40.0, 40.001, 710.0, 5000.0], N),
               known, np.where(rng.random(N) < 0.05, -np.inf, 2.0) * np.abs(sgn[4]),
               np.where(rng.random(N) < 0.1, 0.0, rng.normal(3, 2.5, N)), np.full(N, 2.0), np.full(N, -2.0), 1e80 * sgn[5], 800.0 * sgn[6]]
    rows = rng.choice(384, len(special), replace=False)
    for r, s in zip(rows, special):
        llr[r] = s
    got = both_kernels(g, lambda: g.decode_scl_llr(llr, L)) if L <= 8 else g.decode_scl_llr(llr, L)
    bad = [int(r) for r in range(384) if (got[r] != o.decode_scl_llr(llr[r], L)).any()]
    assert not bad, f"rows {bad} differ (special rows: {sorted(int(r) for r in rows)})"


@pytest.mark.parametrize("L", [1, 4, 32])
def test_one_codeword_at_a_time_equals_the_batch(built_lib, oracle_built, L):
    """The reference's loops call the decoder one codeword at a time (PolarCode.cpp:756, PolarM/main_MC_CC_Comparison.m:96):
    B = 1 through the host-pointer ABI returns, codeword for codeword, what one batched call returns (and the oracle's bits).
    Latency / crossover table: profiles/r03/latency_table.json, DESIGN.md §7."""
    o, g = _pair(11, 1024, 16)
    llr, _ = o.synth_llr(31, 0, 24, o.snr_sqrt_linear(1.5))
    batch = g.decode_scl_llr(llr, L)
    assert (batch == o.decode_scl_llr(llr, L)).all()
    for i in range(24):
        assert (g.decode_scl_llr(llr[i], L) == batch[i]).all(), i          # (L = 1: the one-codeword-per-wave kernel, round 4)
    if L <= 8:
        g.debug_set("lat_max_b", -1)                                       # ... and the batch kernel with B = 1
        for i in range(24):
            assert (g.decode_scl_llr(llr[i], L) == batch[i]).all(), i
        g.debug_set("lat_max_b", 0)


def test_reserve_presizes_the_scratch(built_lib, oracle_built):
    """polar_reserve(B, L): decodes of at most that size afterwards find every scratch buffer in place (same results)."""
    o, g = _pair(9, 256, 8)
    g.reserve(3000, 32)
    llr, _ = o.synth_llr(9, 0, 700, o.snr_sqrt_linear(2.0))
    for L in (1, 4, 32):
        assert (g.decode_scl_llr(llr, L) == o.decode_scl_llr(llr, L)).all(), L


def test_reserve_means_no_allocation_inside_the_asynchronous_calls(built_lib, oracle_built):
    """After polar_reserve(B, L) the device-resident decodes within (B, L) — every list size 1 .. L (the list-size-1 kernel
    with its flag words and scratch, the 2-lane groups, every lane group up to L), doubles and floats, with and without the
    path-metric output — make NO hipMalloc / hipFree (an implicit device synchronisation inside a nominally asynchronous
    call): asserted on the library's allocation counter. A larger batch then does allocate (the counter works)."""
    import torch
    o, g = _pair(9, 256, 8)
    B = 2500
    g.reserve(B, 32)
    llr = torch.empty((B, 512), dtype=torch.float64, device="cuda")
    g.synth_llr_dev(3, 0, B, g.snr_sqrt_linear(2.0), llr.data_ptr())
    f32 = llr.to(torch.float32)
    out = torch.empty((B, 256), dtype=torch.uint8, device="cuda")
    pm = torch.empty(B, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    a0 = g.debug_get("allocs")
    for L in (32, 1, 2, 3, 4, 7, 8, 16, 17, 31):
        for b in (B, B // 3, 1):
            g.decode_scl_llr_dev(llr.data_ptr(), b, L, out.data_ptr())
            g.decode_scl_llr_dev(llr.data_ptr(), b, L, out.data_ptr(), pm_ptr=pm.data_ptr())
            g.decode_scl_llr_dev_f32(f32.data_ptr(), b, L, out.data_ptr())
    torch.cuda.synchronize()
    assert g.debug_get("allocs") == a0
    big = torch.empty((2 * B, 512), dtype=torch.float64, device="cuda")
    g.synth_llr_dev(3, 0, 2 * B, g.snr_sqrt_linear(2.0), big.data_ptr())
    out2 = torch.empty((2 * B, 256), dtype=torch.uint8, device="cuda")
    g.decode_scl_llr_dev(big.data_ptr(), 2 * B, 8, out2.data_ptr())
    torch.cuda.synchronize()
    assert g.debug_get("allocs") > a0


@pytest.mark.parametrize("Lr", [1, 2])
def test_reserve_covers_the_latency_kernels_and_misaligned_rows(built_lib, oracle_built, Lr):
    """polar_reserve(B, 2) with B above the latency threshold ran only the batch kernel of the 2-lane groups; the first SMALL call
    then took the one-codeword-per-wave kernel and allocated its flag and work-list buffers inside the asynchronous call — and
    so did a list-size-1 call from rows that are not 16-byte aligned (the converted copy). Both are part of the reservation now."""
    import torch
    o, g = _pair(11, 1024, 16)
    B = 4096
    g.reserve(B, Lr)
    llr = torch.empty(B * 2048 + 1, dtype=torch.float64, device="cuda")
    g.synth_llr_dev(3, 0, B, g.snr_sqrt_linear(2.0), llr.data_ptr())
    out = torch.empty((B, 1024), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    a0 = g.debug_get("allocs")
    for L in range(1, Lr + 1):
        for b in (1, 3, 200, B):
            g.decode_scl_llr_dev(llr.data_ptr(), b, L, out.data_ptr())
    g.decode_scl_llr_dev(llr.data_ptr() + 8, B, 1, out.data_ptr())          # 8-byte aligned only
    torch.cuda.synchronize()
    assert g.debug_get("allocs") == a0


def test_list_size_one_accepts_rows_that_are_not_16_byte_aligned(built_lib, oracle_built):
    """The in-place channel reads of the list-size-1 kernel are 16-byte vector loads; a caller's pointer with only the
    natural alignment of its element type (a view into a larger buffer, offset by one double / one float) is decoded through
    the converted copy instead: same bits."""
    import torch
    o, g = _pair(11, 1024, 0)
    B = 40
    llr, _ = o.synth_llr(17, 0, B, o.snr_sqrt_linear(2.0))
    want = o.decode_scl_llr(llr, 1)
    buf = torch.zeros(B * 2048 + 3, dtype=torch.float64, device="cuda")
    out = torch.empty((B, 1024), dtype=torch.uint8, device="cuda")
    for off in (0, 1):
        v = buf[off: off + B * 2048]
        v.copy_(torch.from_numpy(llr.reshape(-1)))
        assert v.data_ptr() % 16 == 8 * off
        for lat in (-1, 0):                      # the eight-codeword kernel (in-place reads or front pass), the one-codeword kernel
            g.debug_set("lat_max_b", lat)
            out.zero_()
            g.decode_scl_llr_dev(v.data_ptr(), B, 1, out.data_ptr())
            torch.cuda.synchronize()
            assert (out.cpu().numpy() == want).all(), (off, lat)
    f = llr.astype(np.float32)
    want32 = o.decode_scl_llr(f.astype(np.float64), 1)
    buf32 = torch.zeros(B * 2048 + 5, dtype=torch.float32, device="cuda")
    for off in (0, 1, 2, 3):
        v = buf32[off: off + B * 2048]
        v.copy_(torch.from_numpy(f.reshape(-1)))
        for lat in (-1, 0):
            g.debug_set("lat_max_b", lat)
            out.zero_()
            g.decode_scl_llr_dev_f32(v.data_ptr(), B, 1, out.data_ptr())
            torch.cuda.synchronize()
            assert (out.cpu().numpy() == want32).all(), (off, lat)


@pytest.mark.parametrize("n,K,crc", [(11, 1024, 16), (9, 256, 8), (10, 512, 0), (7, 40, 4)])
@pytest.mark.parametrize("L", [4, 8, 32])
def test_conversion_fused_into_the_prefix_pass(built_lib, oracle_built, n, K, crc, L):
    """Round 4: for the exp-domain list kernels the all-frozen-prefix kernel converts the caller's raw channel rows itself (first
    pass staged through LDS, N <= 2048) instead of reading what a separate conversion pass wrote, and computes the layers below
    from the staged copy. Same bits as the round-3 sequence ("no_fuse_front" hook) and as the oracle, including the rows the input
    guard hands to the LLR-domain kernel (zeros, a sub-1e-9 value, an infinity of the right sign, a row scaled by 1e-3), for
    doubles and floats, and for a batch that is not a multiple of the eight codewords per block."""
    o, g = _pair(n, K, crc)
    N = 1 << n
    B = 203
    llr, _ = o.synth_llr(515 + L, 0, B, o.snr_sqrt_linear(1.5))
    llr[3] = 0.0
    llr[4, 5] = 1e-12
    llr[5, 7] = np.inf * np.sign(llr[5, 7] + 1e-300)
    llr[6] *= 1e-3
    llr[200, 1] = 0.0
    want = o.decode_scl_llr(llr, L)
    g.debug_set("lat_max_b", -1)
    got = g.decode_scl_llr(llr, L)
    g.debug_set("no_fuse_front", 1)
    old = g.decode_scl_llr(llr, L)
    g.debug_set("no_fuse_front", 0)
    ok = np.ones(B, bool); ok[6] = False        # (the "x 1e-3" row is decided at the reference's rounding noise: kernel vs kernel only)
    assert (got == old).all() and (got[ok] == want[ok]).all()
    f = llr.astype(np.float32)
    want32 = o.decode_scl_llr(f.astype(np.float64), L)
    got32 = g.decode_scl_llr(f, L)
    g.debug_set("no_fuse_front", 1)
    old32 = g.decode_scl_llr(f, L)
    g.debug_set("no_fuse_front", 0)
    assert (got32 == old32).all() and (got32[ok] == want32[ok]).all()
    g.debug_set("lat_max_b", 0)
