"""Pure numpy / Python evaluation of the PolarM-only pieces of the hot path, written from the MATLAB text and from
the published Philox4x32-10 definition — no product header, no oracle library involved:
  * PolarCode.m:870-895  polar_decode / cnop / vnop (recursive probability-domain SC, incl. the y = 0.5 -> 0.5 leaf)
  * PolarCode.m:897-914  polar_decode_monte (genie-aided SC of the Monte-Carlo construction)
  * PolarCode.m:855-868  polar_encode (natural recursion)
  * Constellation.m:19-32, 80, 84-93, 123-144  ASK Gray tables, modulate, compute_llr_bicm
  * the counter-based inputs (Philox4x32-10 + Box-Muller) with numpy's libm log/sin/cos/exp
Used by tests/golden/make_polarm_fixtures.py (fixture generator) and by the CPU tests."""
import numpy as np

LEVELS = {1: ([-3, -1, 3, 1], 5.0), 2: ([-7, -5, -1, -3, 7, 5, 1, 3], 21.0),
          3: ([-15, -13, -9, -11, -1, -3, -7, -5, 15, 13, 9, 11, 1, 3, 7, 5], 85.0),
          4: ([1, -1], 1.0)}                          # Constellation.m:19-30 (id 4 = bpsk)


def philox(c, k):
    c = [int(x) for x in c]
    k = [int(x) for x in k]
    for _ in range(10):
        p0 = 0xD2511F53 * c[0]
        p1 = 0xCD9E8D57 * c[2]
        c = [((p1 >> 32) ^ c[1] ^ k[0]) & 0xFFFFFFFF, p1 & 0xFFFFFFFF, ((p0 >> 32) ^ c[3] ^ k[1]) & 0xFFFFFFFF, p0 & 0xFFFFFFFF]
        k = [(k[0] + 0x9E3779B9) & 0xFFFFFFFF, (k[1] + 0xBB67AE85) & 0xFFFFFFFF]
    return c


def _u01(hi, lo):
    return ((((hi << 32) | lo) >> 12) + 0.5) * 2.0 ** -52


def normal_pair(seed, trial, idx, stream):
    """Box-Muller pair number `idx` of `trial` on `stream` (0: BPSK element pairs, 2: symbol pairs)."""
    r = philox([idx, trial & 0xFFFFFFFF, trial >> 32, stream], [seed & 0xFFFFFFFF, seed >> 32])
    u1, u2 = _u01(r[0], r[1]), _u01(r[2], r[3])
    rad = np.sqrt(-2 * np.log(u1))
    return rad * np.cos(2 * np.pi * u2), rad * np.sin(2 * np.pi * u2)


def symbol_noise(seed, trial, nsym):
    z = np.zeros(nsym)
    for s in range(nsym):
        z[s] = normal_pair(seed, trial, s >> 1, 2)[s & 1]
    return z


def bits_from_words(seed, key, stream, nbits):
    """bit i = (word[(i >> 5) & 3] >> (i & 31)) & 1 of Philox block i // 128 (stream 1: info, key = trial // 100;
    stream 3: Monte-Carlo construction message, key = run)."""
    bits = np.zeros(nbits, np.uint8)
    for w in range((nbits + 127) // 128):
        r = philox([w, key & 0xFFFFFFFF, key >> 32, stream], [seed & 0xFFFFFFFF, seed >> 32])
        for i in range(min(128, nbits - 128 * w)):
            bits[128 * w + i] = (r[(i >> 5) & 3] >> (i & 31)) & 1
    return bits


def cnop(a, b):                                       # PolarCode.m:889-891
    return a * (1 - b) + b * (1 - a)


def vnop(a, b):                                       # PolarCode.m:893-895
    return a * b / (a * b + (1 - a) * (1 - b))


def polar_encode(u):                                  # PolarCode.m:857-868
    u = np.asarray(u)
    if len(u) == 1:
        return u.copy()
    return np.concatenate([polar_encode((u[0::2] + u[1::2]) % 2), polar_encode(u[1::2])])


def polar_decode(y, f):                               # PolarCode.m:870-887; returns (u, x)
    N = len(y)
    if N == 1:
        x = np.array([(1 - np.sign(1 - 2 * y[0])) / 2 if f[0] == 0 else 0.0])
        return x.copy(), x
    u1est = cnop(y[0::2], y[1::2])
    uhat1, x1 = polar_decode(u1est, f[: N // 2])
    u2est = vnop(cnop(x1, y[0::2]), y[1::2])
    uhat2, x2 = polar_decode(u2est, f[N // 2:])
    x = np.empty(N)
    x[0::2] = cnop(x1, x2)
    x[1::2] = x2
    return np.concatenate([uhat1, uhat2]), x


def decode_monte(y, info):                            # PolarCode.m:897-914; returns (x, ber)
    N = len(y)
    if N == 1:
        ok = (y[0] > 0.5 and info[0] == 1) or (y[0] <= 0.5 and info[0] == 0)
        return np.array([float(info[0])]), np.array([0 if ok else 1], np.uint8)
    u1est = cnop(y[0::2], y[1::2])
    x1, b1 = decode_monte(u1est, info[: N // 2])
    u2est = vnop(cnop(x1, y[0::2]), y[1::2])
    x2, b2 = decode_monte(u2est, info[N // 2:])
    x = np.empty(N)
    x[0::2] = cnop(x1, x2)
    x[1::2] = x2
    return x, np.concatenate([b1, b2])


def constellation(cid):                               # Constellation.m:19-32, 80
    lv, div = LEVELS[cid]
    pts = np.array(lv, float) / np.sqrt(div)
    return pts / np.sqrt(np.mean(pts ** 2)), int(np.log2(len(lv)))


def modulate(coded, cid):                             # Constellation.m:84-93 (LSB first)
    pts, nb = constellation(cid)
    bits = np.asarray(coded)[: len(coded) // nb * nb].reshape(-1, nb)
    sym = (bits * (1 << np.arange(nb))).sum(1)
    return pts[sym], sym


def compute_llr_bicm(y, n0, cid):                     # Constellation.m:123-144 -> (p1, llr), interleaved (sym-1)*n_bits + j
    pts, nb = constellation(cid)
    ps = np.exp(-np.abs(y[:, None] - pts[None, :]) ** 2 / 2 / n0)
    llr = np.zeros((len(y), nb))
    p1 = np.zeros((len(y), nb))
    for m in range(nb):
        b = (np.arange(len(pts)) >> m) & 1                # bit_sym_map :71-78
        s0, s1 = ps[:, b == 0].sum(1), ps[:, b == 1].sum(1)
        llr[:, m] = np.log(s0 / s1)
        p1[:, m] = s1 / (s0 + s1)
    return p1.reshape(-1), llr.reshape(-1)


def monte_carlo_counts(n, cid, design_snr_db, seed, trial0, runs):   # PolarCode.m:143-196, receiver 'bicm'
    N = 1 << n
    cnt = np.zeros(N, np.int64)
    _, nb = constellation(cid)
    sigma = np.sqrt(0.5) * 10 ** (-design_snr_db / 20)
    for t in range(trial0, trial0 + runs):
        info = bits_from_words(seed, t, 3, N)
        coded = polar_encode(info)
        x, _ = modulate(coded, cid)
        y = x + sigma * symbol_noise(seed, t, len(x))
        p1 = np.full(N, 0.5)
        p1[: len(x) * nb], _ = compute_llr_bicm(y, sigma ** 2, cid)
        cnt += decode_monte(p1, info)[1]
    return cnt
