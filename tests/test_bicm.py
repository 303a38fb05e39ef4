"""ASK Gray / BICM front end (PolarM/Constellation.m:84-93, 123-144; config 5). MATLAB cannot run
in the build image, so this row is 'parity unpinned' by the reference: the restatement is checked
against an independent numpy evaluation of the same formulas (CPU), and the device kernel against
the restatement bit for bit (GPU)."""
import numpy as np
import pytest

from oracle_lib import Oracle

LEVELS = {1: ([-3, -1, 3, 1], 5.0), 2: ([-7, -5, -1, -3, 7, 5, 1, 3], 21.0),
          3: ([-15, -13, -9, -11, -1, -3, -7, -5, 15, 13, 9, 11, 1, 3, 7, 5], 85.0)}


def _numpy_bicm(coded, cid, noise, snr_db):
    lv, div = LEVELS[cid]
    pts = np.array(lv, float) / np.sqrt(div)
    pts = pts / np.sqrt(np.mean(pts ** 2))                       # Constellation.m:80
    nb = int(np.log2(len(lv)))
    bits = coded[: len(coded) // nb * nb].reshape(-1, nb)
    sym = (bits * (1 << np.arange(nb))).sum(1)                    # :86-91 (LSB first)
    sigma = np.sqrt(0.5) * 10 ** (-snr_db / 20)                   # main_MC_CC_Comparison.m:90
    y = pts[sym] + noise * sigma
    ps = np.exp(-np.abs(y[:, None] - pts[None, :]) ** 2 / 2 / sigma ** 2)      # :127
    llr = np.zeros((len(y), nb))
    for m in range(nb):
        b = (np.arange(len(lv)) >> m) & 1                         # bit_sym_map :71-78
        llr[:, m] = np.log(ps[:, b == 0].sum(1) / ps[:, b == 1].sum(1))         # :142
    return llr.reshape(-1), y


@pytest.mark.parametrize("cid", [1, 2, 3])
def test_bicm_front_end_matches_independent_numpy(oracle_built, cid):
    o = Oracle(10, 512, 0.32, 0, srand=1)
    nb = {1: 2, 2: 3, 3: 4}[cid]
    nsym = 1024 // nb
    for snr in (4.0, 9.0, 13.0):
        llr, info = o.synth_bicm_llr(cid, 77, 5, 3, snr)
        for t in range(3):
            coded = o.encode(info[t])
            assert (llr[t, nsym * nb:] == 0).all()          # unused tail positions: p1 = 0.5 <=> llr = 0
            hard = (llr[t, : nsym * nb] < 0).astype(np.uint8)
            assert (hard != coded[: nsym * nb]).mean() < 0.35
    # formula check: independent Python evaluation of the noise generator + numpy evaluation of
    # Constellation.m's modulate / compute_llr_bicm with libm exp/log (tolerance = libm vs the
    # fixed-order exp/log of polar_synth.h)
    l1, i1 = o.synth_bicm_llr(cid, 9, 0, 1, 6.0)
    coded = o.encode(i1[0])
    z = _oracle_symbol_noise(9, 0, nsym)
    want, _ = _numpy_bicm(coded, cid, z, 6.0)
    assert np.allclose(l1[0, : len(want)], want, rtol=1e-9, atol=1e-9)


def _philox(c, k):
    c = [int(x) for x in c]
    k = [int(x) for x in k]
    for _ in range(10):
        p0 = 0xD2511F53 * c[0]
        p1 = 0xCD9E8D57 * c[2]
        c = [((p1 >> 32) ^ c[1] ^ k[0]) & 0xFFFFFFFF, p1 & 0xFFFFFFFF, ((p0 >> 32) ^ c[3] ^ k[1]) & 0xFFFFFFFF, p0 & 0xFFFFFFFF]
        k = [(k[0] + 0x9E3779B9) & 0xFFFFFFFF, (k[1] + 0xBB67AE85) & 0xFFFFFFFF]
    return c


def _oracle_symbol_noise(seed, trial, nsym):
    """Independent (pure Python + numpy libm) evaluation of polar_synth_symbol_noise."""
    z = np.zeros(nsym)
    for s in range(nsym):
        r = _philox([s >> 1, trial & 0xFFFFFFFF, trial >> 32, 2], [seed & 0xFFFFFFFF, seed >> 32])
        u1 = ((((r[0] << 32) | r[1]) >> 12) + 0.5) * 2.0 ** -52
        u2 = ((((r[2] << 32) | r[3]) >> 12) + 0.5) * 2.0 ** -52
        rad = np.sqrt(-2 * np.log(u1))
        z[s] = rad * (np.sin(2 * np.pi * u2) if s & 1 else np.cos(2 * np.pi * u2))
    return z


def test_bpsk_noise_generator_matches_independent_python(oracle_built):
    """The BPSK workload too: Philox + Box-Muller + PolarCode.cpp:715,747,752 arithmetic."""
    o = Oracle(6, 32, 0.32, 0, srand=1)
    s = o.snr_sqrt_linear(2.0)
    llr, info = o.synth_llr(123, 7, 1, s)
    coded = o.encode(info[0])
    want = np.zeros(64)
    for pr in range(32):
        r = _philox([pr, 7, 0, 0], [123, 0])
        u1 = ((((r[0] << 32) | r[1]) >> 12) + 0.5) * 2.0 ** -52
        u2 = ((((r[2] << 32) | r[3]) >> 12) + 0.5) * 2.0 ** -52
        rad = np.sqrt(-2 * np.log(u1))
        zz = (rad * np.cos(2 * np.pi * u2), rad * np.sin(2 * np.pi * u2))
        for k in range(2):
            bp = 2.0 * coded[2 * pr + k] - 1.0
            y = s * bp + np.sqrt(0.5) * zz[k]
            want[2 * pr + k] = -4 * y * s
    assert np.allclose(llr[0], want, rtol=1e-12, atol=1e-12)


@pytest.mark.gpu
def test_device_bicm_synth_and_mc_match_oracle(built_lib, oracle_built):
    import torch
    import golden_util as G
    import polar_amd
    c, frozen, order, crcm = G.tables("cfg5_n10_k512_ask16")
    g = polar_amd.PolarCode.from_tables(10, 512, 0, frozen, order, None)
    o = Oracle(10, 512, 0.5, 0)
    o.set_tables(frozen, order)
    for cid in (1, 2, 3):
        llr, info = o.synth_bicm_llr(cid, 31, 1000, 70, 11.5)
        d_llr = torch.empty((70, 1024), dtype=torch.float64, device="cuda")
        d_info = torch.empty((70, 512), dtype=torch.uint8, device="cuda")
        g.synth_bicm_llr_dev(cid, 31, 1000, 70, 11.5, d_llr.data_ptr(), d_info.data_ptr())
        torch.cuda.synchronize()
        assert (d_llr.cpu().numpy() == llr).all()
        assert (d_info.cpu().numpy() == info).all()
    snr, Ls = [11.0, 12.0, 13.0, 14.0], [1, 8]
    en = np.ones((2, 4), np.uint8)
    e1, r1 = np.zeros((2, 4), np.uint64), np.zeros((2, 4), np.uint64)
    e2, r2 = np.zeros((2, 4), np.uint64), np.zeros((2, 4), np.uint64)
    o.mc_batch_bicm(3, 8, 0, 120, 1, snr, Ls, en, e1, r1)
    g.mc_batch_bicm(3, 8, 0, 120, 1, snr, Ls, en, e2, r2)
    assert (e1 == e2).all() and (r1 == r2).all() and e1.sum() > 0


@pytest.mark.gpu
def test_config5_curve_shape(built_lib):
    """16-ASK Gray BICM, (1024,512) MC-constructed code, SC (L=1): results/monte_carlo.png reads
    ~0.2 @ Eb/N0 9.25 dB, ~1e-2 @ 10.5 dB; Eb/N0 = SNR - 10log10(4) + 10log10(2)."""
    import golden_util as G
    import polar_amd
    c, frozen, order, crcm = G.tables("cfg5_n10_k512_ask16")
    g = polar_amd.PolarCode.from_tables(10, 512, 0, frozen, order, None)
    snr = [9.25 + 3.0103, 10.5 + 3.0103]
    en = np.ones((2, 2), np.uint8)
    e, r = np.zeros((2, 2), np.uint64), np.zeros((2, 2), np.uint64)
    g.mc_batch_bicm(3, 2, 0, 4000, 1, snr, [1, 8], en, e, r)
    bler = e / r
    assert 0.08 < bler[0, 0] < 0.4, bler
    assert 0.002 < bler[0, 1] < 0.04, bler
    assert (bler[1] <= bler[0] + 1e-9).all()          # L=8 no worse than SC
