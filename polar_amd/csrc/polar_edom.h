// polar_edom.h — device-side fp64 helpers shared by the decode kernels (gfx950): the table-driven exp / log of the
// LLR-domain kernel and the exp-domain ("E-form") node arithmetic of the fast kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "polar_device.h"

#ifndef POLAR_NO_COLD_HINTS
#define POLAR_UNLIKELY(x) __builtin_expect(!!(x), 0)
#define POLAR_LIKELY(x) __builtin_expect(!!(x), 1)
#else
#define POLAR_UNLIKELY(x) (x)
#define POLAR_LIKELY(x) (x)
#endif
// second batch of layout hints (measured: within the run-to-run noise on top of the first two; kept)
#define POLAR_UNLIKELY2(x) POLAR_UNLIKELY(x)
#define POLAR_LIKELY2(x) POLAR_LIKELY(x)

namespace {

// LDS tables of the fp64 exp / log routines: T[64] = 2^(f/64), RC[129] = 1/(1+j/128), LC[129] = log(1+j/128)
struct Tabs { const double *T, *RC, *LC; };   // LDS: T[64], RC[129], LC[129]

// NOTE: the Horner starts below are written as a separate multiply and add on purpose. A fused
// fma(x, c1, c2) has two constant operands, one of which must sit in a VGPR pair; the compiler hoists
// that pair out of every loop, runs out of registers and RELOADS it from scratch (with a full
// s_waitcnt vmcnt(0)) inside each f-node. mul-by-constant + add-constant needs no VGPR constant.
__device__ __forceinline__ double exp_neg(double x, const Tabs &tb) {   // e^-x, x >= 0
    // -x = k*ln2/64 + s with k = rint(-x*64/ln2) <= 0:  e^-x = 2^(k>>6) * T[k&63] * e^s,  T[f] = 2^(f/64)
    // (arithmetic shift / two's-complement mask of the NEGATIVE index: no negation, no second table)
    const double kd = __builtin_rint(x * -92.332482616893657);           // -64/ln2
    const int k = (int)kd;
    double s = __builtin_fma(kd, -0.010830424696223417, -x);             // ln2/64, high part (low 16 bits zero)
    s = __builtin_fma(kd, -2.5728046223276688e-14, s);                   // ln2/64, low part
    double p = s * (1.0 / 120.0) + 1.0 / 24.0;
    p = __builtin_fma(p, s, 1.0 / 6.0);
    p = __builtin_fma(p, s, 0.5);
    p = __builtin_fma(p, s, 1.0);
    p = __builtin_fma(p, s, 1.0);
    return __builtin_ldexp(tb.T[k & 63] * p, k >> 6);
}
// table slot of m in [1,2]: m rounded to 7 mantissa bits IS the expansion point c_j = 1 + j/128
// (integer work on the high word instead of subtract / scale / rint / convert / fma)
__device__ __forceinline__ int log_slot(double m, double &c) {
    const int ch = (__double2hiint(m) + 0x1000) & (int)0xFFFFE000;
    c = __hiloint2double(ch, 0);
    return (ch - 0x3FF00000) >> 13;                                       // j in [0, 128]
}
__device__ __forceinline__ double log_1p2(double m, const Tabs &tb) {    // log(m), m in [1,2]
    double c;
    const int j = log_slot(m, c);
    const double q = (m - c) * tb.RC[j];
    double p = q * (-1.0 / 6.0) + 0.2;
    p = __builtin_fma(p, q, -0.25);
    p = __builtin_fma(p, q, 1.0 / 3.0);
    p = __builtin_fma(p, q, -0.5);
    p = __builtin_fma(p, q, 1.0);
    return __builtin_fma(q, p, tb.LC[j]);
}

// ================= exp-domain ("E-form") node arithmetic =======================================
// The LLR-domain f-node needs four transcendentals (two exp + two log1p, ~70 instructions); in the
// likelihood-ratio domain it is ONE division. A stored value v is
//     |v| <= 1 :  E-form,  |v| = e^-|x|,  sign(v) = sign(x)          (x = the reference's LLR, |x| < T_E)
//     |v| >  1 :  L-form,  v = x itself                               (|x| >= T_E = 690: e^-|x| would underflow)
// so that a relative rounding error of 1.1e-16 in |v| is an ABSOLUTE error of 1.1e-16 in x: the same level
// of accuracy as the table-driven LLR-domain f-node above (and as the reference's own 1 + e^x), for every
// magnitude below 690.
//     f exact (both |x| < 40, PolarCode.cpp:438-441):  E_y = (E_a + E_b) / (1 + E_a E_b),  sign = sa*sb
//     f min-sum (:442-446):                            the input with the smaller |x|, sign = sa*sb
//     g (:449-450), signs equal after (1-2u):          E_y = E_a E_b
//                   signs opposite:                    E_y = min(E_a,E_b) / max(E_a,E_b), sign of the larger |x|
//     g with an L-form input or an underflowing product (a few % of the wave-steps in the two lowest
//     layers at 2 dB, none above): the reference's own addition in the LLR domain, with log / exp at the
//     regime boundary only.
// The path metric stays in the LLR domain: log(1+e^-|x|) = log1p(E), |x| = -log(E) at the leaves.
// Decisions within ~1e-10 (relative) of the reference's |x| < 40 test are not taken here: the codeword is
// flagged and decoded again by the LLR-domain kernel (guard mask, see scl_decode_llr_kernel).
constexpr double ED_T = 690.0;                       // E-form iff |x| < ED_T
constexpr double ED_EMIN = 2.3e-300;                 // < e^-690 = 2.26e-300 ... products below this leave the E-form
constexpr double ED_C40_HI = 4.248354255291589e-18 * (1.0 + 1e-10);   // e^-40 (1 +- 1e-10)
constexpr double ED_C40_LO = 4.248354255291589e-18 * (1.0 - 1e-10);
#ifndef ED_NR
#define ED_NR 1     // v_rcp_f64 is good to 2^-24: one Newton step (2^-48) and the quotient correction (error squared again)
#endif
// num / den for normal operands well inside the exponent range: v_rcp_f64 seed, Newton steps on the
// reciprocal, one correction of the quotient (which squares the remaining error: <= 1 ulp)
__device__ __forceinline__ double ed_div(double num, double den) {
    double r = __builtin_amdgcn_rcp(den);
    double e = __builtin_fma(-den, r, 1.0);
    r = __builtin_fma(r, e, r);
#if ED_NR >= 2
    e = __builtin_fma(-den, r, 1.0);
    r = __builtin_fma(r, e, r);
#endif
    const double q = num * r;
    const double e2 = __builtin_fma(-den, q, num);
    return __builtin_fma(e2, r, q);
}
__device__ __forceinline__ double ed_with_sign(double r, int signword) {     // r >= 0
    int hi; asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(hi) : "v"(signword), "s"(0x80000000u), "v"(__double2hiint(r)));
    return __hiloint2double(hi, __double2loint(r));
}
// max / min of the MAGNITUDES as single instructions with |.| source modifiers. Written through the builtins the compiler puts a
// canonicalising v_max_f64 x, x in front of each operand (IEEE mode: maxnum must quiet signalling NaNs) — two more VALU
// instructions per pair, on values that are never NaN here (finite stored forms; non-finite channel values are flagged before).
__device__ __forceinline__ double ed_absmax(double a, double b) { double r; asm("v_max_f64 %0, |%1|, |%2|" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ double ed_absmin(double a, double b) { double r; asm("v_min_f64 %0, |%1|, |%2|" : "=v"(r) : "v"(a), "v"(b)); return r; }
// f-node. `guard` collects (as a wave mask) the lanes whose |x| < 40 decision is too close to call.
__device__ __forceinline__ double f_node_e(double a, double b, u64 &guard) {
    const double fa = fabs(a), fb = fabs(b);
    const double mx = ed_absmax(a, b), mn = ed_absmin(a, b);
    const double q = ed_div(fa + fb, __builtin_fma(fa, fb, 1.0));
    const u64 m_hi = __builtin_amdgcn_fcmp(mn, ED_C40_HI, 2);            // mn > e^-40 (1 + 1e-10): certainly |x| < 40
    const u64 m_lo = __builtin_amdgcn_fcmp(mn, ED_C40_LO, 2);
    const u64 m_l = __builtin_amdgcn_fcmp(mx, 1.0, 2);                   // an L-form input (rare)
    guard |= (m_hi ^ m_lo) & ~m_l;
    // both E-form: exact value, or (min-sum, :442-446) the smaller |x| = the larger E
    double r = __builtin_amdgcn_inverse_ballot_w64(m_hi) ? q : mx;
    // one L-form -> the E-form input (mn); both L-form -> the smaller |x| (mn)
    // (POLAR_UNLIKELY: the block is laid out away from the hot path. The list kernel's code is several times the 64-KiB
    // instruction cache two CUs share, and rarely-taken blocks sitting between the hot ones cost fetch misses on every
    // pass: marking this branch and the one in g_node_e measured +5 % on the headline kernel.)
    if (POLAR_UNLIKELY(m_l != 0)) r = __builtin_amdgcn_inverse_ballot_w64(m_l) ? mn : r;
    return ed_with_sign(r, __double2hiint(a) ^ __double2hiint(b));
}
// the same node with the guard kept PER LANE: `gacc` = the smallest distance seen so far between a node's smaller E and the
// threshold; the codeword is flagged at its end when any of its lanes came within 2e-10 (relative) of it. One add and one
// min instead of a second compare, two scalar mask operations and the merge into the wave mask — which the register
// allocator keeps in a VGPR pair (two more VALU instructions per node).
__device__ __forceinline__ double f_node_e_acc(double a, double b, double &gacc) {
    const double fa = fabs(a), fb = fabs(b);
    const double mx = ed_absmax(a, b), mn = ed_absmin(a, b);
    const double q = ed_div(fa + fb, __builtin_fma(fa, fb, 1.0));
    const u64 m_hi = __builtin_amdgcn_fcmp(mn, ED_C40_HI, 2);            // mn > e^-40 (1 + 1e-10): certainly |x| < 40
    const u64 m_l = __builtin_amdgcn_fcmp(mx, 1.0, 2);                   // an L-form input (rare)
#if !defined(POLAR_NO_GUARD_INT) && !(defined(POLAR_ED_TU) && POLAR_ED_TU == 2)
    // (A/B, round 4: groups of 4 lanes +0.8 %, groups of 8 +5 %; the list-of-32 translation unit, whose register allocation
    // every change of the node code shifts, measured -0.45 % with it and keeps the per-node distance below)
    // Round 4: the distance to the threshold is not tracked at every node (an fp64 add and an fp64 min) but only where it
    // can matter: every double within 2e-10 (relative) of e^-40 — the whole flagging window, and 2^-20 relative around it —
    // has the HIGH WORD 0x3C539792, so ONE 32-bit integer compare of the smaller E's high word finds the candidates and the
    // exact distance is taken in the rare block (shared with the L-form select) only.
    const u64 m_near = __builtin_amdgcn_uicmp((unsigned)__double2hiint(mn), 0x3C539792u, 32);     // ICMP_EQ
    double r = __builtin_amdgcn_inverse_ballot_w64(m_hi) ? q : mx;
    if (POLAR_UNLIKELY((m_l | m_near) != 0)) {
        r = __builtin_amdgcn_inverse_ballot_w64(m_l) ? mn : r;
        gacc = ed_absmin(gacc, mn - ED_C40_HI);
    }
#else
    gacc = ed_absmin(gacc, mn - ED_C40_HI);
    double r = __builtin_amdgcn_inverse_ballot_w64(m_hi) ? q : mx;
    if (POLAR_UNLIKELY(m_l != 0)) r = __builtin_amdgcn_inverse_ballot_w64(m_l) ? mn : r;
#endif
    return ed_with_sign(r, __double2hiint(a) ^ __double2hiint(b));
}
constexpr double ED_GACC_FLAG = 4.248354255291589e-18 * 2.0000001e-10;      // flagged: |mn - e^-40 (1 + 1e-10)| <= this
// general natural logarithm of a positive normal double (tables of log_1p2)
__device__ __forceinline__ double ed_log(double x, const Tabs &tb) {
    const int hi = __double2hiint(x);
    const double e = (double)((hi >> 20) - 1023);
    const double m = __hiloint2double((hi & 0x000FFFFF) | 0x3FF00000, __double2loint(x));
    // ln2 split: high part with 32 significant bits (e * hi is exact), low part the rest
    return __builtin_fma(e, 6.93147180369123816490e-01, __builtin_fma(e, 1.90821492927058770002e-10, log_1p2(m, tb)));
}
// |x| of a stored value (E-form: -log E; L-form: itself)
__device__ __forceinline__ double ed_abs_llr(double v, const Tabs &tb) {
    const double m = fabs(v);
    const double l = -ed_log(__builtin_fmin(__builtin_fmax(m, ED_EMIN), 1.0), tb);
    return (m > 1.0) ? m : l;
}
// canonical stored form of an LLR x
__device__ __forceinline__ double ed_from_llr(double x, const Tabs &tb) {
    const double fx = fabs(x);
    const double e = exp_neg(__builtin_fmin(fx, 700.0), tb);
    return (fx >= ED_T) ? x : ed_with_sign(e, __double2hiint(x));
}
// rare path of the g-node: needed in a few % of the wave-steps of the two lowest layers only
// (measured: as a real call — noinline — the kernel is 9 % SLOWER: the call ABI costs scratch spills in the callers)
__device__ __forceinline__ double g_node_e_rare(double a, double b, int ha, int hb, const double *tabs) {
    const Tabs tb = {tabs, tabs + 64, tabs + 64 + 129};
    // reference arithmetic in the LLR domain
    const double xa = ed_with_sign(ed_abs_llr(a, tb), ha), xb = ed_with_sign(ed_abs_llr(b, tb), hb);
    return ed_from_llr(xa + xb, tb);
}
// g-node: (1-2u) a + b; `usign` carries u in bit 31 (the other bits are ignored)
__device__ __forceinline__ double g_node_e(double a, double b, unsigned usign, const Tabs &tb) {
    const int ha = __double2hiint(a) ^ (int)usign, hb = __double2hiint(b);      // only the sign bits of ha/hb are used
    // (the sign test as a wave mask straight from the compare: through a bool the compiler materialises 0/1 in a VGPR and
    // compares it again for the ballot below — two VALU instructions per node)
    const u64 m_same = __builtin_amdgcn_sicmp((int)(ha ^ hb), 0, 39);      // ICMP_SGE: signs equal after (1 - 2u)
    const bool same = __builtin_amdgcn_inverse_ballot_w64(m_same);
    const double p = fabs(a) * fabs(b);
    const double lo = ed_absmin(a, b), hi = ed_absmax(a, b);
    const double q = ed_div(lo, hi);                      // == 1.0 exactly when |a| == |b| (b - a = 0)
    const double r = same ? p : q;
    int sg = (fabs(a) < fabs(b)) ? ha : hb;               // opposite signs: the larger |x| (smaller E) decides
    sg = same ? hb : sg;                                  // (redundant — equal signs: either input's — but without it the headline
                                                          //  kernel measured 0.5 .. 1.5 % slower: instruction layout, not arithmetic)
    double res = ed_with_sign(r, sg);
    const u64 m_rare = __builtin_amdgcn_fcmp(hi, 1.0, 2) | (m_same & __builtin_amdgcn_fcmp(p, ED_EMIN, 4));      // OGT; OLT
    if (POLAR_UNLIKELY(m_rare != 0)) {
        const double sl = g_node_e_rare(a, b, ha, hb, tb.T);
        if (__builtin_amdgcn_inverse_ballot_w64(m_rare)) res = sl;
    }
    return res;
}
// channel LLR -> stored form, with the input guard (non-finite, or so small that the reference's f/g
// results are its own rounding noise)
__device__ __forceinline__ double ed_from_channel(double x, const Tabs &tb, bool &flag) {
    const double fx = fabs(x);
    flag = !(fx < __builtin_inf()) || fx < 1e-9;
    return ed_from_llr(x, tb);
}

}  // namespace
