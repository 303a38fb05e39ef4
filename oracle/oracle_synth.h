/* oracle/oracle_synth.h — TEST INFRASTRUCTURE: the oracle's OWN statement of the synthetic workload.
 *
 * The workload (counter-based Philox4x32-10 inputs, Box-Muller normals, the channel arithmetic of
 * PolarCode.cpp:715,744-752, the ASK Gray / BICM front end of PolarM/Constellation.m:19-32,80,84-93,123-144) is
 * DEFINED by the product header include/polar_synth.h; bit-identity between host and device requires one fixed
 * order of IEEE basic operations, so the definition pins: the generator and its counter/key layout, "52 bits ->
 * (k + 1/2) 2^-52", the atanh-series logarithm, the Taylor sin/cos of the first octant, the 13-term exponential.
 * This file restates that definition independently (own structure: coefficient tables and loops instead of
 * unrolled Horner chains), so that "device == oracle" compares two separately written sources; both are also
 * checked against a third, numpy, evaluation (tests/polarm_numpy.py, tests/test_polarm_fixtures.py).
 * Build with -ffp-contract=off. */
#ifndef ORACLE_SYNTH_H
#define ORACLE_SYNTH_H

#include <stdint.h>
#include <string.h>

enum { OSY_NOISE = 0, OSY_INFO = 1, OSY_SYMNOISE = 2, OSY_MCINFO = 3 };          /* 4th counter word */
enum { OSY_ASK4 = 1, OSY_ASK8 = 2, OSY_ASK16 = 3, OSY_BPSK = 4 };

/* Philox4x32-10, Salmon et al. SC'11: multipliers 0xD2511F53 / 0xCD9E8D57, Weyl increments 0x9E3779B9 / 0xBB67AE85 */
static void osy_philox(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c[4], k[2];
    memcpy(c, ctr, sizeof c);
    memcpy(k, key, sizeof k);
    for (int round = 0; round < 10; ++round) {
        const uint64_t lo_prod = 0xD2511F53ull * c[0], hi_prod = 0xCD9E8D57ull * c[2];
        const uint32_t nc[4] = {(uint32_t)(hi_prod >> 32) ^ c[1] ^ k[0], (uint32_t)hi_prod,
                                (uint32_t)(lo_prod >> 32) ^ c[3] ^ k[1], (uint32_t)lo_prod};
        memcpy(c, nc, sizeof c);
        k[0] += 0x9E3779B9u;
        k[1] += 0xBB67AE85u;
    }
    memcpy(out, c, sizeof c);
}
static void osy_block(uint64_t seed, uint64_t key64, uint32_t idx, uint32_t stream, uint32_t out[4]) {
    const uint32_t ctr[4] = {idx, (uint32_t)key64, (uint32_t)(key64 >> 32), stream};
    const uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    osy_philox(ctr, key, out);
}
static double osy_unit(uint32_t hi, uint32_t lo) {            /* top 52 of the 64 bits, centred: (k + 1/2) 2^-52 */
    const uint64_t k = ((uint64_t)hi << 32 | lo) >> 12;
    return ((double)k + 0.5) * 0x1p-52;
}
/* ln x on (0,1]: x = m 2^e with m in (sqrt(1/2), sqrt 2], ln m = 2 s (1 + z/3 + ... + z^12/25), s = (m-1)/(m+1), z = s^2 */
static double osy_ln(double x) {
    uint64_t bits;
    memcpy(&bits, &x, 8);
    int e = (int)(bits >> 52 & 0x7FF) - 1023;
    bits = (bits & 0xFFFFFFFFFFFFFull) | 0x3FF0000000000000ull;
    double m;
    memcpy(&m, &bits, 8);
    if (m > 1.4142135623730951) { m = m * 0.5; e = e + 1; }
    const double s = (m - 1.0) / (m + 1.0), z = s * s;
    double p = 1.0 / 25.0;
    for (int d = 23; d >= 3; d -= 2) p = p * z + 1.0 / (double)d;
    p = p * z + 1.0;
    return (double)e * 0.6931471805599453 + 2.0 * s * p;
}
/* sin, cos of 2 pi u: quadrant q = floor(4u), remainder folded to [0, 1/2] of a quadrant, Taylor to x^17 / x^16 */
static void osy_sincos_turn(double u, double *sn, double *cs) {
    const double t = u * 4.0;
    const int q = (int)t;
    double f = t - (double)q;
    const int fold = f > 0.5;
    if (fold) f = 1.0 - f;
    const double x = f * 1.5707963267948966, z = x * x;
    static const double inv_fact_odd[9] = {355687428096000.0, 1307674368000.0, 6227020800.0, 39916800.0, 362880.0, 5040.0, 120.0, 6.0, 1.0};
    static const double inv_fact_even[9] = {20922789888000.0, 87178291200.0, 479001600.0, 3628800.0, 40320.0, 720.0, 24.0, 2.0, 1.0};
    double ps = 1.0 / inv_fact_odd[0], pc = 1.0 / inv_fact_even[0];
    for (int i = 1; i < 9; ++i) {
        ps = 1.0 / inv_fact_odd[i] - ps * z;
        pc = 1.0 / inv_fact_even[i] - pc * z;
    }
    double s0 = x * ps, c0 = pc;
    if (fold) { const double tmp = s0; s0 = c0; c0 = tmp; }
    switch (q) {                                               /* rotation by q quarter turns */
        case 0: *sn = s0; *cs = c0; break;
        case 1: *sn = c0; *cs = -s0; break;
        case 2: *sn = -s0; *cs = -c0; break;
        default: *sn = -c0; *cs = s0; break;
    }
}
static void osy_normal_pair(const uint32_t r[4], double *z_cos, double *z_sin) {   /* Box-Muller */
    const double rad = __builtin_sqrt(-2.0 * osy_ln(osy_unit(r[0], r[1])));
    double sn, cs;
    osy_sincos_turn(osy_unit(r[2], r[3]), &sn, &cs);
    *z_cos = rad * cs;
    *z_sin = rad * sn;
}
static void osy_noise_pair(uint64_t seed, uint64_t trial, uint32_t pair, double *z0, double *z1) {
    uint32_t r[4];
    osy_block(seed, trial, pair, OSY_NOISE, r);
    osy_normal_pair(r, z0, z1);
}
static double osy_symbol_noise(uint64_t seed, uint64_t trial, uint32_t sym) {
    uint32_t r[4];
    double zc, zs;
    osy_block(seed, trial, sym >> 1, OSY_SYMNOISE, r);
    osy_normal_pair(r, &zc, &zs);
    return (sym & 1) ? zs : zc;
}
/* bit i of a 128-bit block = (word[(i >> 5) & 3] >> (i & 31)) & 1 */
static int osy_bit(uint64_t seed, uint64_t key64, uint32_t stream, int i) {
    uint32_t r[4];
    osy_block(seed, key64, (uint32_t)(i >> 7), stream, r);
    const int k = i & 127;
    return (int)(r[(k >> 5) & 3] >> (k & 31) & 1u);
}
/* PolarCode.cpp:715,747,752 with N_0 = 1 */
static double osy_bpsk_llr(double s, int coded_bit, double z) {
    const double y = s * (coded_bit ? 1.0 : -1.0) + 0.7071067811865476 * z;
    return (-4.0 * y) * s;
}

/* ---- ASK Gray constellations and the BICM demapper ---- */
static int osy_nbits(int id) { return id == OSY_BPSK ? 1 : id + 1; }
static double osy_point(int id, int sym) {                    /* Constellation.m:19-30 levels / sqrt(mean square of the grid) */
    static const int ask4[4] = {-3, -1, 3, 1}, ask8[8] = {-7, -5, -1, -3, 7, 5, 1, 3};
    static const int ask16[16] = {-15, -13, -9, -11, -1, -3, -7, -5, 15, 13, 9, 11, 1, 3, 7, 5};
    switch (id) {
        case OSY_BPSK: return ((sym & 1) ? -1.0 : 1.0) / __builtin_sqrt(1.0);
        case OSY_ASK4: return (double)ask4[sym & 3] / __builtin_sqrt(5.0);
        case OSY_ASK8: return (double)ask8[sym & 7] / __builtin_sqrt(21.0);
        default: return (double)ask16[sym & 15] / __builtin_sqrt(85.0);
    }
}
static double osy_norm(int id) {                              /* Constellation.m:80 */
    const int ns = 1 << osy_nbits(id);
    double acc = 0.0;
    for (int s = 0; s < ns; ++s) { const double x = osy_point(id, s); acc = acc + x * x; }
    return __builtin_sqrt(acc / (double)ns);
}
/* e^x, x <= 0: x = k ln2 + r (k = trunc(x log2 e - 1/2)), e^r by its Taylor polynomial of degree 13, times 2^k */
static double osy_exp_neg(double x) {
    if (x < -708.0) return 0.0;
    const int k = (int)(x * 1.4426950408889634 - 0.5);
    const double kd = (double)k;
    const double r = (x - kd * 0.693147180369123816490) - kd * 1.90821492927058770002e-10;
    static const double fact[12] = {6227020800.0, 479001600.0, 39916800.0, 3628800.0, 362880.0, 40320.0, 5040.0, 720.0, 120.0, 24.0, 6.0, 2.0};
    double p = 1.0 / fact[0];
    for (int i = 1; i < 11; ++i) p = p * r + 1.0 / fact[i];
    p = p * r + 0.5;
    p = p * r + 1.0;
    p = p * r + 1.0;
    uint64_t bits = (uint64_t)(1023 + k) << 52;
    double scale;
    memcpy(&scale, &bits, 8);
    return p * scale;
}
/* Constellation.m:123-144 for one received symbol: llr_j = log(p0_j / p1_j), p1_j / (p0_j + p1_j); outputs may be NULL */
static void osy_demap(int id, double norm, double y, double n0, double *llr_out, double *p1_out) {
    const int nb = osy_nbits(id), ns = 1 << nb;
    double sum[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
    for (int s = 0; s < ns; ++s) {
        const double d = y - osy_point(id, s) / norm, ad = d < 0 ? -d : d;
        const double ps = osy_exp_neg(-(ad * ad) / 2 / n0);
        for (int m = 0; m < nb; ++m) sum[(s >> m) & 1][m] = sum[(s >> m) & 1][m] + ps;
    }
    for (int m = 0; m < nb; ++m) {
        if (llr_out) llr_out[m] = osy_ln(sum[0][m] / sum[1][m]);
        if (p1_out) p1_out[m] = sum[1][m] / (sum[0][m] + sum[1][m]);
    }
}

#endif /* ORACLE_SYNTH_H */
