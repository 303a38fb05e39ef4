/* polar_synth.h — definition of the synthetic BPSK/AWGN workload.
 *
 * The reference draws its Monte-Carlo inputs from order-dependent serial
 * generators (glibc rand(), std::default_random_engine + std::normal_distribution,
 * PolarCode.cpp:688-710) that cannot be reproduced by a parallel device.  This
 * header DEFINES the replacement: a counter-based generator (Philox4x32-10) keyed
 * by (seed, trial index), so that any trial can be produced independently by any
 * lane of any GPU — or by the CPU oracle — with bit-identical results, and the
 * channel arithmetic of PolarCode.cpp:715,744-752 applied on top.
 *
 * Bit-identity between gcc (host) and hipcc (gfx950 device) is by construction:
 * only IEEE-754 correctly-rounded basic operations (+ - * / sqrt) and integer
 * bit manipulation are used, in one fixed order; no libm/ocml transcendental is
 * called.  BOTH compilers must build this with -ffp-contract=off.
 *
 * It is a workload definition shared by the product (device generator used by the
 * Monte-Carlo engine and bench.py) and by the test oracle; it contains no decoder
 * logic.
 */
#ifndef POLAR_SYNTH_H
#define POLAR_SYNTH_H

#include <stdint.h>

#if defined(__HIPCC__)
#define POLAR_SYNTH_FN __host__ __device__ static inline
#else
#define POLAR_SYNTH_FN static inline
#endif

/* stream ids (4th counter word) */
#define POLAR_SYNTH_STREAM_NOISE 0u
#define POLAR_SYNTH_STREAM_INFO 1u

/* ---- Philox4x32-10 (Salmon et al., SC'11; standard constants) ---- */
POLAR_SYNTH_FN void polar_philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                     uint32_t k0, uint32_t k1, uint32_t out[4]) {
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* 52 random bits -> double in (0,1), exactly representable: (k + 0.5) * 2^-52 */
POLAR_SYNTH_FN double polar_synth_u01(uint32_t hi, uint32_t lo) {
    uint64_t k = (((uint64_t)hi << 32) | lo) >> 12;
    return ((double)k + 0.5) * 2.220446049250313e-16; /* 2^-52 */
}

/* natural log for x in (0,1], fixed operation order (atanh series), ~1e-16 relative */
POLAR_SYNTH_FN double polar_synth_log(double x) {
    union { double d; uint64_t u; } v;
    v.d = x;
    int e = (int)((v.u >> 52) & 0x7FF) - 1023;
    v.u = (v.u & 0x000FFFFFFFFFFFFFull) | 0x3FF0000000000000ull; /* m in [1,2) */
    double m = v.d;
    if (m > 1.4142135623730951) { m = m * 0.5; e = e + 1; } /* m in (sqrt(.5), sqrt(2)] */
    double s = (m - 1.0) / (m + 1.0);
    double z = s * s;
    double p = 1.0 / 25.0;
    p = p * z + 1.0 / 23.0;
    p = p * z + 1.0 / 21.0;
    p = p * z + 1.0 / 19.0;
    p = p * z + 1.0 / 17.0;
    p = p * z + 1.0 / 15.0;
    p = p * z + 1.0 / 13.0;
    p = p * z + 1.0 / 11.0;
    p = p * z + 1.0 / 9.0;
    p = p * z + 1.0 / 7.0;
    p = p * z + 1.0 / 5.0;
    p = p * z + 1.0 / 3.0;
    p = p * z + 1.0;
    return (double)e * 0.6931471805599453 + 2.0 * s * p;
}

/* sin and cos of 2*pi*u for u in (0,1), fixed operation order, ~1e-16 absolute */
POLAR_SYNTH_FN void polar_synth_sincos2pi(double u, double *sn, double *cs) {
    double t = u * 4.0;          /* exact */
    int q = (int)t;              /* quadrant 0..3 */
    double f = t - (double)q;    /* exact, in [0,1) */
    int swap = 0;
    if (f > 0.5) { f = 1.0 - f; swap = 1; } /* exact */
    double x = f * 1.5707963267948966; /* in [0, pi/4] */
    double z = x * x;
    /* sin x = x * (1 - z/6 + z^2/120 - ...) up to x^17 */
    double ps = 1.0 / 355687428096000.0;          /* 1/17! */
    ps = 1.0 / 1307674368000.0 - ps * z;          /* 1/15! */
    ps = 1.0 / 6227020800.0 - ps * z;             /* 1/13! */
    ps = 1.0 / 39916800.0 - ps * z;               /* 1/11! */
    ps = 1.0 / 362880.0 - ps * z;                 /* 1/9!  */
    ps = 1.0 / 5040.0 - ps * z;                   /* 1/7!  */
    ps = 1.0 / 120.0 - ps * z;                    /* 1/5!  */
    ps = 1.0 / 6.0 - ps * z;                      /* 1/3!  */
    ps = 1.0 - ps * z;
    double s0 = x * ps;
    /* cos x up to x^16 */
    double pc = 1.0 / 20922789888000.0;           /* 1/16! */
    pc = 1.0 / 87178291200.0 - pc * z;            /* 1/14! */
    pc = 1.0 / 479001600.0 - pc * z;              /* 1/12! */
    pc = 1.0 / 3628800.0 - pc * z;                /* 1/10! */
    pc = 1.0 / 40320.0 - pc * z;                  /* 1/8!  */
    pc = 1.0 / 720.0 - pc * z;                    /* 1/6!  */
    pc = 1.0 / 24.0 - pc * z;                     /* 1/4!  */
    pc = 1.0 / 2.0 - pc * z;                      /* 1/2!  */
    pc = 1.0 - pc * z;
    double c0 = pc;
    if (swap) { double tmp = s0; s0 = c0; c0 = tmp; } /* angle = pi/2 - x within the quadrant */
    /* rotate by q * pi/2 */
    double so, co;
    if (q == 0) { so = s0; co = c0; }
    else if (q == 1) { so = c0; co = -s0; }
    else if (q == 2) { so = -s0; co = -c0; }
    else { so = -c0; co = s0; }
    *sn = so; *cs = co;
}

/* Two independent N(0,1) variates for element pair `pair` (elements 2*pair, 2*pair+1)
 * of trial `trial` (Box-Muller on two 52-bit uniforms). */
POLAR_SYNTH_FN void polar_synth_noise_pair(uint64_t seed, uint64_t trial, uint32_t pair,
                                           double *z0, double *z1) {
    uint32_t r[4];
    polar_philox4x32(pair, (uint32_t)trial, (uint32_t)(trial >> 32), POLAR_SYNTH_STREAM_NOISE,
                     (uint32_t)seed, (uint32_t)(seed >> 32), r);
    double u1 = polar_synth_u01(r[0], r[1]);
    double u2 = polar_synth_u01(r[2], r[3]);
    double rad;
    {
        double a = -2.0 * polar_synth_log(u1);
#if defined(__HIP_DEVICE_COMPILE__)
        rad = __builtin_sqrt(a);
#else
        rad = __builtin_sqrt(a);
#endif
    }
    double sn, cs;
    polar_synth_sincos2pi(u2, &sn, &cs);
    *z0 = rad * cs;
    *z1 = rad * sn;
}

/* 128 information bits: word w (bits 128w .. 128w+127) of info block `block`.
 * The reference refreshes its info word every 100 runs (PolarCode.cpp:703-707);
 * callers pass block = trial / 100 to mirror that. Bit i of the block is
 * (out[(i>>5)&3] >> (i&31)) & 1. */
POLAR_SYNTH_FN void polar_synth_info_word(uint64_t seed, uint64_t block, uint32_t w, uint32_t out[4]) {
    polar_philox4x32(w, (uint32_t)block, (uint32_t)(block >> 32), POLAR_SYNTH_STREAM_INFO,
                     (uint32_t)seed, (uint32_t)(seed >> 32), out);
}

/* BPSK + AWGN + LLR, arithmetic exactly as PolarCode.cpp:715,747,752 with N_0 = 1:
 *   bpsk = 2c - 1;  y = s*bpsk + sqrt(N_0/2)*z;  llr = -4*y*s/N_0
 * `s` (= 10^(EbN0/20) * sqrt(K/N), PolarCode.cpp:744-745) is computed ONCE by the
 * host and passed in, so device and host never disagree on a libm pow(). */
POLAR_SYNTH_FN double polar_synth_llr(double s, int coded_bit, double z) {
    double bpsk = coded_bit ? 1.0 : -1.0;
    double y = s * bpsk + 0.7071067811865476 * z; /* sqrt(1/2) */
    return (-4.0 * y) * s;
}

/* ===================== ASK Gray / BICM front end (PolarM/Constellation.m) =====================
 * Constellation tables :19-32, unit-energy normalisation :80, symbol index = sum 2^(j-1) bit_j
 * (LSB first) :86-91, bit_sym_map(sym, j) = bit j-1 of sym :71-78, BICM demapper :123-144.
 * MATLAB cannot run in the build image, so this part is "parity unpinned" by the reference; it is
 * cross-checked against an independent numpy evaluation (tests/test_bicm.py). exp/log use the
 * fixed-order routines of this header so that host and device agree bit for bit. */
#define POLAR_SYNTH_STREAM_SYMNOISE 2u
#define POLAR_CONST_ASK4_GRAY 1
#define POLAR_CONST_ASK8_GRAY 2
#define POLAR_CONST_ASK16_GRAY 3
#define POLAR_CONST_BPSK 4          /* Constellation.m:19 bpsk = [1 -1] (the default of the MC code construction) */
#define POLAR_SYNTH_STREAM_MCINFO 3u

POLAR_SYNTH_FN int polar_const_nbits(int id) {
    return id == POLAR_CONST_BPSK ? 1 : (id == POLAR_CONST_ASK4_GRAY ? 2 : (id == POLAR_CONST_ASK8_GRAY ? 3 : 4));
}

/* un-normalised integer levels (Constellation.m:21,25,29-30) and the sqrt() divisor */
POLAR_SYNTH_FN double polar_const_point(int id, int sym) {
    double lvl, div;
    if (id == POLAR_CONST_BPSK) {
        lvl = (sym & 1) ? -1.0 : 1.0; div = 1.0;
    } else if (id == POLAR_CONST_ASK4_GRAY) {
        const int t[4] = {-3, -1, 3, 1};
        lvl = (double)t[sym & 3]; div = 5.0;
    } else if (id == POLAR_CONST_ASK8_GRAY) {
        const int t[8] = {-7, -5, -1, -3, 7, 5, 1, 3};
        lvl = (double)t[sym & 7]; div = 21.0;
    } else {
        const int t[16] = {-15, -13, -9, -11, -1, -3, -7, -5, 15, 13, 9, 11, 1, 3, 7, 5};
        lvl = (double)t[sym & 15]; div = 85.0;
    }
    return lvl / __builtin_sqrt(div);
}
/* constellation_points / sqrt(mean(constellation_points.^2)) (:80); mean = sequential sum / n_sym */
POLAR_SYNTH_FN double polar_const_norm(int id) {
    const int ns = 1 << polar_const_nbits(id);
    double acc = 0.0;
    for (int s = 0; s < ns; ++s) { double x = polar_const_point(id, s); acc = acc + x * x; }
    return __builtin_sqrt(acc / (double)ns);
}

/* e^x for x <= 0 (fixed operation order; flushes to 0 below 2^-1022) */
POLAR_SYNTH_FN double polar_synth_exp_neg(double x) {
    if (x < -708.0) return 0.0;
    double t = x * 1.4426950408889634;
    int k = (int)(t - 0.5);                         /* x <= 0: round to nearest by truncation */
    double kd = (double)k;
    double r = (x - kd * 0.693147180369123816490) - kd * 1.90821492927058770002e-10;  /* fdlibm ln2 hi/lo */
    double p = 1.0 / 6227020800.0;                  /* 1/13! */
    p = p * r + 1.0 / 479001600.0;
    p = p * r + 1.0 / 39916800.0;
    p = p * r + 1.0 / 3628800.0;
    p = p * r + 1.0 / 362880.0;
    p = p * r + 1.0 / 40320.0;
    p = p * r + 1.0 / 5040.0;
    p = p * r + 1.0 / 720.0;
    p = p * r + 1.0 / 120.0;
    p = p * r + 1.0 / 24.0;
    p = p * r + 1.0 / 6.0;
    p = p * r + 0.5;
    p = p * r + 1.0;
    p = p * r + 1.0;
    union { double d; uint64_t u; } v;
    v.u = (uint64_t)(1023 + k) << 52;               /* 2^k, k in [-1022, 0] */
    return p * v.d;
}

/* one received symbol -> n_bits LLRs and/or bit probabilities, Constellation.m:123-144:
 * p_sym = exp(-|y-x|^2/2/n0), llr = log(p0/p1) (:142), p1 = p1/(p0+p1) (:143). Either output
 * pointer may be NULL. */
POLAR_SYNTH_FN void polar_synth_bicm_demap2(int id, double norm, double y, double n0, double *llr_out, double *p1_out) {
    const int nb = polar_const_nbits(id), ns = 1 << nb;
    double p0[4] = {0, 0, 0, 0}, p1[4] = {0, 0, 0, 0};
    for (int s = 0; s < ns; ++s) {
        double d = y - polar_const_point(id, s) / norm;
        double ad = d < 0 ? -d : d;
        double ps = polar_synth_exp_neg(-(ad * ad) / 2 / n0);
        for (int m = 0; m < nb; ++m) {
            if (((s >> m) & 1) == 0) p0[m] = p0[m] + ps; else p1[m] = p1[m] + ps;
        }
    }
    for (int m = 0; m < nb; ++m) {
        if (llr_out) llr_out[m] = polar_synth_log(p0[m] / p1[m]);
        if (p1_out) p1_out[m] = p1[m] / (p0[m] + p1[m]);
    }
}
POLAR_SYNTH_FN void polar_synth_bicm_demap(int id, double norm, double y, double n0, double *llr_out) {
    polar_synth_bicm_demap2(id, norm, y, n0, llr_out, (double *)0);
}

/* 128 of the N random message bits of Monte-Carlo construction run `trial` (PolarCode.m:153:
 * dummy_info = rand(1, N) < 0.5, fresh for every run); same bit addressing as polar_synth_info_word */
POLAR_SYNTH_FN void polar_synth_mc_info_word(uint64_t seed, uint64_t trial, uint32_t w, uint32_t out[4]) {
    polar_philox4x32(w, (uint32_t)trial, (uint32_t)(trial >> 32), POLAR_SYNTH_STREAM_MCINFO,
                     (uint32_t)seed, (uint32_t)(seed >> 32), out);
}

/* N(0,1) variate for symbol `sym` of trial `trial` (one Box-Muller pair per two symbols) */
POLAR_SYNTH_FN double polar_synth_symbol_noise(uint64_t seed, uint64_t trial, uint32_t sym) {
    uint32_t r[4];
    polar_philox4x32(sym >> 1, (uint32_t)trial, (uint32_t)(trial >> 32), POLAR_SYNTH_STREAM_SYMNOISE,
                     (uint32_t)seed, (uint32_t)(seed >> 32), r);
    double u1 = polar_synth_u01(r[0], r[1]);
    double u2 = polar_synth_u01(r[2], r[3]);
    double rad = __builtin_sqrt(-2.0 * polar_synth_log(u1));
    double sn, cs;
    polar_synth_sincos2pi(u2, &sn, &cs);
    return (sym & 1) ? rad * sn : rad * cs;
}

#endif /* POLAR_SYNTH_H */
