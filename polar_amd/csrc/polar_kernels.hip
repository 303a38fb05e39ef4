// polar_kernels.hip — gfx950 (CDNA4) kernels for the polar SC/SCL hot path.
//
// Design: ONE LANE PER LIST PATH.  A wavefront (64 lanes) decodes
// G = 64/GS codewords at once, GS = pow2ceil(L) lanes per codeword; lane `lig` of a group
// IS path index `lig` of the reference (PolarCode.cpp's `l`).  All codewords and all paths
// follow the same successive-cancellation schedule (it depends only on phi and the frozen
// mask), so the 64 lanes run the N-step recursion in lockstep with no divergence; the only
// cross-lane work is the fork/prune step (rank of 2L fork metrics, LIFO path-index stack,
// clone = register shuffle).
//
// Memory: every per-path array is laid out [layer][element][lane] so that the 64 lanes of a
// wave touch 64 consecutive doubles (512 B) — coalesced in HBM/L2, conflict-free in LDS.
// The Tal-Vardy lazy copy (getArrayPointer_*, PolarCode.cpp:305-373) becomes a per-lane,
// per-layer "slot pointer": a path WRITES its own slot (= its lane) and READS the slot its
// pointer names; cloning copies the pointers (registers) only. Because a (layer) array is
// always completely rewritten by all active paths in the same step, no copy-on-write and no
// reference counting is needed, and a permutation of slots inside a group keeps the access
// inside the same 512-byte row.
//
// Element order inside a layer is bit-reversed w.r.t. the reference (position j holds the
// reference's beta = bitrev(j)), so that a node combines elements (j, j+S) and partial sums
// are combined by word-wise XOR/concatenation instead of a bit interleave.
//
// Arithmetic is IEEE double with the reference's formulas and operation order
// (PolarCode.cpp:437-451, 483, 505-506); build with -ffp-contract=off.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "polar_kernels.h"
#include "polar_device.h"
#include "polar_edom.h"

#ifdef POLAR_PROFILE
#define PROF_DECL u64 prof_acc[24] = {0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0}; u64 prof_t = __builtin_readcyclecounter();
#define PROF(i) { u64 t_ = __builtin_readcyclecounter(); prof_acc[i] += t_ - prof_t; prof_t = t_; }
#define PROF_OUT if (lane == 0 && p.pm_out) { for (int i_ = 0; i_ < 24; ++i_) atomicAdd((u64 *)p.pm_out + i_, prof_acc[i_]); }
#define PROF_CNT(i, v) { prof_acc[i] += (u64)(v); }
#else
#define PROF_DECL
#define PROF(i)
#define PROF_CNT(i, v)
#define PROF_OUT
#endif

namespace {

// ---- fp64 transcendentals for the f-node and the path metric --------------------------------
// The reference evaluates log((e^(a+b)+1)/(e^a+e^b)) and log(1+e^x) with libm. Bit-identity with
// glibc's exp/log is not reachable on a GPU (ocml differs in the last ulp as well); what parity
// needs is that every DECISION (sign of a leaf LLR, order of path metrics) is the reference's,
// i.e. an absolute accuracy far below any decision margin. These routines keep ~1e-16 absolute
// accuracy (the rounding level of the reference's own 1+e^x) at ~1/3 of the instruction count
// of the libm-style sequence, using two small LDS tables (no division):
//   exp(-x) = T[k&63] * 2^(k>>6) * p5(s),  -x = k*ln2/64 + s  (k <= 0, T[f] = 2^(f/64))
//   log(m)  = LC[j] + log1p((m - c_j)/c_j),   c_j = 1 + j/128 = m rounded to 7 mantissa bits
// and the identity  f(a,b) = sgn(a)sgn(b)min(|a|,|b|) + h(|a+b|) - h(|a-b|),  h(x) = log1p(e^-x).
// Structural exactness is preserved: h(x) == 0 exactly for x >= 36.74 (where the reference's
// 1+e^-x rounds to 1), f(0,b) == 0 exactly, f is symmetric, log(1+e^x) -> +inf for x > 709.78.
__device__ __forceinline__ double h_fn(double x, const Tabs &tb) {       // log1p(e^-x), x >= 0
    return log_1p2(1.0 + exp_neg(x, tb), tb);
}
// h(x) - h(y): the two evaluations of an f-node written in lockstep, so that their table reads are
// issued together (2 LDS round trips per f-node instead of 4) and the two dependent fp64 chains
// overlap. Same operations and rounding as h_fn(x) - h_fn(y).
__device__ __forceinline__ double h_diff(double x, double y, const Tabs &tb) {
    const double kx = __builtin_rint(x * -92.332482616893657), ky = __builtin_rint(y * -92.332482616893657);
    const int ix = (int)kx, iy = (int)ky;
    const double tx = tb.T[ix & 63], ty = tb.T[iy & 63];
    double sx = __builtin_fma(kx, -0.010830424696223417, -x), sy = __builtin_fma(ky, -0.010830424696223417, -y);
    sx = __builtin_fma(kx, -2.5728046223276688e-14, sx); sy = __builtin_fma(ky, -2.5728046223276688e-14, sy);
    double px = sx * (1.0 / 120.0) + 1.0 / 24.0, py = sy * (1.0 / 120.0) + 1.0 / 24.0;
    px = __builtin_fma(px, sx, 1.0 / 6.0); py = __builtin_fma(py, sy, 1.0 / 6.0);
    px = __builtin_fma(px, sx, 0.5); py = __builtin_fma(py, sy, 0.5);
    px = __builtin_fma(px, sx, 1.0); py = __builtin_fma(py, sy, 1.0);
    px = __builtin_fma(px, sx, 1.0); py = __builtin_fma(py, sy, 1.0);
    const double mx = 1.0 + __builtin_ldexp(tx * px, ix >> 6), my = 1.0 + __builtin_ldexp(ty * py, iy >> 6);
    double cx, cy;
    const int nx = log_slot(mx, cx), ny = log_slot(my, cy);
    const double rcx = tb.RC[nx], rcy = tb.RC[ny], lcx = tb.LC[nx], lcy = tb.LC[ny];
    const double qx = (mx - cx) * rcx, qy = (my - cy) * rcy;
    double ux = qx * (-1.0 / 6.0) + 0.2, uy = qy * (-1.0 / 6.0) + 0.2;
    ux = __builtin_fma(ux, qx, -0.25); uy = __builtin_fma(uy, qy, -0.25);
    ux = __builtin_fma(ux, qx, 1.0 / 3.0); uy = __builtin_fma(uy, qy, 1.0 / 3.0);
    ux = __builtin_fma(ux, qx, -0.5); uy = __builtin_fma(uy, qy, -0.5);
    ux = __builtin_fma(ux, qx, 1.0); uy = __builtin_fma(uy, qy, 1.0);
    return __builtin_fma(qx, ux, lcx) - __builtin_fma(qy, uy, lcy);
}
__device__ __attribute__((noinline)) double f_literal(double a, double b) {
    return log((exp(a + b) + 1) / (exp(a) + exp(b)));
}
__device__ __attribute__((noinline)) double softplus_literal(double x) { return log(1 + exp(x)); }
// f-node (check node), exact + min-sum branches: PolarCode.cpp:437-446
__device__ __forceinline__ double f_node(double a, double b, const Tabs &tb) {
    const double fa = fabs(a), fb = fabs(b);
    const double mx = __builtin_fmax(fa, fb);      // (v_max_f64 / v_min_f64 with |.| source modifiers)
    const double mn = __builtin_fmin(fa, fb);
    // sgn(a)*sgn(b)*min(|a|,|b|) (PolarCode.cpp:443-445, sgn(0) = 0): the magnitude with the XOR of the
    // two sign bits, done on the high word instead of int->double conversions and multiplies; it is
    // also the leading term of the exact expression below
    const int sx = (__double2hiint(a) ^ __double2hiint(b)) & (int)0x80000000;
    const double ms = __hiloint2double(__double2hiint(mn) | sx, __double2loint(mn));
    if (40 > mx) {
        // |f| <= min(|a|,|b|): when that is within a few orders of the rounding noise (1e-16) the
        // reference's result IS its rounding noise (e.g. exactly 0 once e^a, e^b round to 1), so the
        // literal expression is evaluated for those (physically never occurring) elements.
        if (POLAR_UNLIKELY2(mn < 9.5367431640625e-07)) return f_literal(a, b);
        return ms + h_diff(fabs(a + b), fabs(a - b), tb);
    }
    return (mn == 0.0) ? 0.0 : ms;     // min-sum branch
}
// Two f-nodes at once: same results as f_node() twice, but ONE wave-uniform branch around the two exact
// evaluations, so that their four h() chains sit in one basic block and overlap (a per-node divergent
// branch serialises the nodes). Used where the nodes are otherwise strictly serial (rate-0 blocks); in
// the unrolled layer loops it was measured slower (register pressure: -1.4 % fused loop, -25 % LDS visits). Lanes that do not need the exact value compute it on whatever they
// hold (finite garbage at worst: the table index is masked) and discard it.
__device__ __forceinline__ void f_node2(double a0, double b0, double a1, double b1, const Tabs &tb, double &r0, double &r1) {
    const double fa0 = fabs(a0), fb0 = fabs(b0), fa1 = fabs(a1), fb1 = fabs(b1);
    const double mx0 = __builtin_fmax(fa0, fb0), mn0 = __builtin_fmin(fa0, fb0);
    const double mx1 = __builtin_fmax(fa1, fb1), mn1 = __builtin_fmin(fa1, fb1);
    const int s0 = (__double2hiint(a0) ^ __double2hiint(b0)) & (int)0x80000000;
    const int s1 = (__double2hiint(a1) ^ __double2hiint(b1)) & (int)0x80000000;
    const double m0 = __hiloint2double(__double2hiint(mn0) | s0, __double2loint(mn0));     // sgn*sgn*min (min-sum value)
    const double m1 = __hiloint2double(__double2hiint(mn1) | s1, __double2loint(mn1));
    r0 = (mn0 == 0.0) ? 0.0 : m0;
    r1 = (mn1 == 0.0) ? 0.0 : m1;
    const bool e0 = 40 > mx0, e1 = 40 > mx1;
    if (wave_any(e0 || e1)) {
        const double x0 = m0 + h_diff(fabs(a0 + b0), fabs(a0 - b0), tb);
        const double x1 = m1 + h_diff(fabs(a1 + b1), fabs(a1 - b1), tb);
        const bool t0 = e0 && mn0 < 9.5367431640625e-07, t1 = e1 && mn1 < 9.5367431640625e-07;
        if (e0) r0 = x0;
        if (e1) r1 = x1;
        if (POLAR_UNLIKELY2(wave_any(t0 || t1))) {                    // noise regime (see f_node)
            if (t0) r0 = f_literal(a0, b0);
            if (t1) r1 = f_literal(a1, b1);
        }
    }
}
// g-node: PolarCode.cpp:449-450  (1 - 2u)*a + b
__device__ __forceinline__ double g_node(double a, double b, unsigned u) {
    // (1 - 2u) is +1 or -1 and the product with it is exact: flip the sign bit of a, then add
    const double sa = __hiloint2double(__double2hiint(a) ^ (int)(u << 31), __double2loint(a));
    return sa + b;
}
// The path-metric terms log(1 + exp(-+llr)) of PolarCode.cpp:483,505-506 for a = |llr| >= 0, with ONE
// h evaluation: log(1+e^-a) = h(a) (exactly 0 for a >= 36.74, where the reference's 1+e^-a rounds to
// 1), log(1+e^a) = a + h(a) (+inf beyond the fp64 exp overflow point 709.78, as the reference).
// `skip` (wave-uniform: every lane has a >= 37) avoids the transcendental altogether.
__device__ __forceinline__ void softplus_pair(double a, bool skip, const Tabs &tb, double &sneg, double &spos) {
    double hx = 0.0;
    if (!skip) {
        if (POLAR_UNLIKELY2(a < 9.5367431640625e-07)) {            // noise regime: literal expressions (see f_node)
            sneg = softplus_literal(-a);
            spos = softplus_literal(a);
            return;
        }
        // per-lane saturation: 1 + e^-a rounds to 1 for a >= 37, and an infinite (or > 1e78) leaf LLR must not
        // reach the range reduction of exp_neg (inf * c - inf = NaN would poison the metric)
        hx = (a >= 37.0) ? 0.0 : h_fn(__builtin_fmin(a, 37.0), tb);
    }
    sneg = hx;
    spos = (a > 709.782712893384) ? __builtin_inf() : a + hx;
}


// leaf terms for the path metric. LLR-domain kernel: `leaf` is the LLR; E-domain: stored form.
//   neg  = (llr < 0);  al = |llr|;  sneg = log(1+e^-|llr|);  spos = log(1+e^|llr|)
// actw: the wave mask of `active` (kept by the caller: a ballot of a compound bool costs a round trip through a VGPR)
template <bool ED>
__device__ __forceinline__ void leaf_terms(double leaf, bool active, u64 actw, const Tabs &tb, bool &neg, double &al, double &sneg, double &spos) {
    if (!ED) {
        al = fabs(leaf);
        neg = leaf < 0;
        const bool skip = wave_all(!active || al >= 37.0);
        sneg = 0.0; spos = 0.0;
        if (active) softplus_pair(al, skip, tb, sneg, spos);
    } else {
        const double m = fabs(leaf);
        const bool isl = m > 1.0;
        neg = (__double2hiint(leaf) < 0) && m != 1.0;
        al = m;
        const u64 m_e = actw & __builtin_amdgcn_fcmp(m, 1.0, 13);           // ULE: active lanes holding an E-form value
        if (POLAR_LIKELY2(m_e != 0)) {
            const double l = -ed_log(__builtin_fmin(__builtin_fmax(m, ED_EMIN), 1.0), tb);
            if (!isl) al = l;
        }
        const double onep = 1.0 + m;                 // == 1 exactly from E <= 2^-53 on, as the reference's 1 + e^-|x|
        sneg = 0.0;
        if (POLAR_LIKELY2((m_e & __builtin_amdgcn_fcmp(onep, 1.0, 14)) != 0)) {      // UNE
            const double h = log_1p2(__builtin_fmin(onep, 2.0), tb);
            if (!isl) sneg = h;
        }
        spos = (al > 709.782712893384) ? __builtin_inf() : al + sneg;
    }
}
}  // namespace

// Channel LLRs at the boundary are doubles (the reference's type) or floats (polar_decode_scl_llr_batch_dev_f32):
// a float is widened — exactly — in the load itself, there is no staging copy. ch_row() = row `cw` of p.llr.
template <bool ED>
__device__ __forceinline__ const double *ch_row(const PolarDecodeParams &p, size_t cw, int N) {
    return p.llr_f32 ? reinterpret_cast<const double *>(reinterpret_cast<const float *>(p.llr) + cw * (size_t)N)
                     : p.llr + cw * (size_t)N;
}
// (the exp-domain kernels never see floats — ed_front_kernel has widened and converted the channel values — but folding
// that into the macro measured 1 % SLOWER on the headline kernel: register allocation of the hot loops shifts)
#define CH(row, i) (p.llr_f32 ? (double)reinterpret_cast<const float *>(row)[i] : (row)[i])

// Layer storage helpers --------------------------------------------------------------------
// LDS:    layers with S <= SL; layer of size S starts at element (S-1); element e at [e*64 + lane]
// global: layers with S >  SL; layer of size S starts at element (S-2*SL)
//
// GS  : lanes per codeword (power of two >= L)
// LDS_LOG : log2 of the largest layer size kept in LDS
#ifndef FU
#define FU 4
#endif
#ifndef POLAR_SKIP_L1
#define POLAR_SKIP_L1 1      // the layer of size 1 is never stored (+0.45 %)
#endif
#define POLAR_SADDR 1        // HBM rows of the four-layer visits addressed as SGPR base + 32-bit lane offset (+0.4 %)
#ifndef OCC
#define OCC 4
#endif
// per-phase re-derivation of the lane-dependent invariants (see lane_id_opaque in polar_device.h)
#define LANE_CTX                                   \
    const int lane = lane_id_opaque();             \
    const int lig = lane & (GS - 1);               \
    const int gbase = lane & ~(GS - 1);            \
    (void)lig; (void)gbase;
// NL: 0 = block length taken from p (any code); else log2 of the block length this instantiation is compiled for (the
// headline shapes: every layer size, row offset and loop bound is then a constant)
// LAT = 1: the LATENCY form for small batches (round 4) — ONE codeword per wave. The 64 / GS lane groups that otherwise hold
// different codewords share the ELEMENTS of one codeword: lane = e * GS + l decodes path l and owns the elements j = e (mod 64 / GS) of
// every layer. Everything a path carries (metric, slot pointers, partial sums, history) is replicated in its 64 / GS lanes, which
// execute the leaf steps — fork, prune, clone, CRC, selection: the code below, unchanged — in lockstep with identical operands; only
// the layer visits differ (lat_visit), and the whole state lives in LDS: layers (N - 1) GS doubles (element j of slot s at
// [(j GS + s)]: a wave access is 64 consecutive doubles), the converted channel, the partial-sum and history words. A lone wave of the
// batch kernel pays an HBM round trip per dependent access of its scratch layers and evaluates every element of a layer serially.
template <int GS, int LDS_LOG, int PIPE, bool ED, int NL = 0, int LAT = 0>
__global__ __launch_bounds__(PIPE ? 64 : 256, PIPE ? 2 : OCC) void scl_decode_llr_kernel(PolarDecodeParams p) {
    // ED: exp-domain node arithmetic (see f_node_e); the channel values at p.llr are then in stored form
    // (ed_front_kernel) and every codeword whose decisions are not safely reproduced is reported in p.flags
    // PIPE=1: one wave per block (8 waves/CU, register double-buffering); PIPE=0: four independent
    // waves per block sharing the transcendental tables (16 waves/CU with LDS_LOG = 3)
    constexpr int WPB = PIPE ? 1 : 4;
    constexpr int G = LAT ? 1 : 64 / GS;           // codewords per wave
    constexpr int EL = 64 / GS;                    // (LAT) lanes that share the elements of a path's layers
    // partial-sum and history words: one column per LANE ([word][64]); LAT: per PATH ([word][GS] — the 64 / GS lanes of a path hold
    // the same words and write the same values to the same place)
    constexpr int CST = LAT ? GS : 64;
#define POLAR_CL (LAT ? lig : lane)
#define POLAR_CGB (LAT ? 0 : gbase)
    constexpr int SL = 1 << LDS_LOG;
    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // wave in block (uniform: keeps every per-wave base pointer in SGPRs)
    const int wave_id = blockIdx.x * WPB + wib;             // owns one slice of the global scratch
    const int nwaves = gridDim.x * WPB;
    const int lig = lane & (GS - 1);   // path index l of the reference
    const int gbase = lane & ~(GS - 1);
    const int grp = LAT ? 0 : lane / GS;
    const int n = NL ? NL : p.n, N = NL ? (1 << NL) : p.N, K = p.K, L = p.L;
    const u64 gmask = (GS == 64) ? ~0ull : ((1ull << GS) - 1ull);

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double *tabs = reinterpret_cast<double *>(smem);                           // T[64] RC[129] LC[129] (+2 pad), per block
    constexpr size_t WAVE_LDS = (size_t)(2 * SL - 1) * 64 * 8 + 128 * 8 + 128;  // bytes per wave
    unsigned char *wbase = smem + 324 * 8 + (size_t)wib * WAVE_LDS;
    double *lds_llr = reinterpret_cast<double *>(wbase);                       // [(2*SL-1)][64] (LAT: not there — its layers are lat_a)
    double *sortbuf = lds_llr + (LAT ? 0 : (size_t)(2 * SL - 1) * 64);         // [128]
    // (plain pointers, ordered by wave_mem_fence(): a volatile-qualified pointer loses its LDS address
    // space and every access becomes a system-coherent FLAT operation that waits for all memory)
    unsigned char *stackv = reinterpret_cast<unsigned char *>(sortbuf + 128);   // [64]
    unsigned char *srcof = stackv + 64;                                         // [64]
    for (int i = threadIdx.x; i < 322; i += WPB * 64) tabs[i] = p.tabs[i];
    if (WPB > 1) __syncthreads();
    const Tabs tb = {tabs, tabs + 64, tabs + 64 + 129};
    u64 guard = 0;                                  // (ED) wave mask of lanes with an undecidable |x| < 40 test
    double gacc = __builtin_inf();                  // (ED) per lane: smallest distance of a node's smaller E to the |x| < 40 threshold
    auto FN = [&](double a, double b) -> double {
        if constexpr (ED) return f_node_e_acc(a, b, gacc); else return f_node(a, b, tb);
    };
    // g-node of element with partial-sum bit `bi` of the word `cw_` (u = (cw_ >> bi) & 1)
    auto GN = [&](double a, double b, uint32_t cw_, int bi) -> double {
        if constexpr (ED) return g_node_e(a, b, cw_ << (31 - bi), tb); else return g_node(a, b, (cw_ >> bi) & 1u);
    };
    // (fallback pass: codewords come from p.cw_list; Monte-Carlo: only the first *p.n_dev rows are alive)
    const long Bv = p.cw_count ? (long)*p.cw_count : (p.n_dev ? ((long)*p.n_dev < p.B ? (long)*p.n_dev : p.B) : p.B);

    // per-wave global scratch (LAT: the same arrays in LDS, behind the wave's other LDS data)
    const size_t big_elems = (N > 2 * SL) ? (size_t)(N - 2 * SL) : 0;
    double *g_llr = LAT ? nullptr : p.llr_scr + (size_t)wave_id * big_elems * 64;
    const int cwords = (N >= 128) ? (N / 32 - 2) : 0;                          // words of big C layers (S >= 64)
    double *lat_ch = reinterpret_cast<double *>(stackv + 128);                  // (LAT) [N] channel, stored form, kernel element order
    double *lat_a = lat_ch + N;                                                // (LAT) layers: size S at (S - 1) GS, element j of slot s at (j GS + s)
    uint32_t *lat_w = reinterpret_cast<uint32_t *>(lat_a + (size_t)N * GS);
    uint32_t *g_cl = LAT ? lat_w : p.c_scr + (size_t)wave_id * 2 * (size_t)cwords * 64;
    uint32_t *g_cr = g_cl + (size_t)cwords * CST;
    uint32_t *g_hist = LAT ? g_cr + (size_t)cwords * CST : p.hist_scr + (size_t)wave_id * 3 * (size_t)p.W * 64;    // decision words [W][64]
    uint32_t *g_horg = g_hist + (size_t)p.W * CST;                             // link to the previous word's slot
    uint32_t *g_tb = g_horg + (size_t)p.W * CST;                               // winner's words, per-lane copy
    // ---- table mode (list size 17..32, N >= 1024, exp-domain): layers 1 and 2 are never stored per path.
    // Every path's layer-1 value x1[e] = g(ch, ch', u1[e]) is one of TWO numbers, its layer-2 value one of 2 / 4 / 8
    // (phi = N/4: g of the shared first-half layer 1 with u2[j]; phi = N/2: f(x1[j], x1[j+N/4]) -> u1[j], u1[j+N/4];
    // phi = 3N/4: g(...) -> additionally u2[j]) — the bits being the path's partial sums. The 32 lanes of a codeword
    // build those values ONCE per codeword (T2[j][variant], 64 B per element: one cache line serves all paths) and each
    // path keeps 3 bits per element (V words, reached through a slot pointer like every per-path array). The visits of
    // layer 3 gather their inputs from the table. Two of the seven HBM-resident layers disappear.
    const int S1 = N / 2, S2 = N / 4;
    const bool tbl = ED && !PIPE && GS == 32 && N >= 1024 && p.tab_scr != nullptr && p.prefix_q > 0;
    double *tab_w = tbl ? p.tab_scr + (size_t)wave_id * G * (size_t)(3 * N) : nullptr;       // per codeword: X[N/2][2], T2[N/4][8]
    uint32_t *g_v = tbl ? p.var_scr + (size_t)wave_id * (size_t)(S2 / 8) * 64 : nullptr;     // V words [N/32][64]

    // Work distribution: a wave's first group of codewords is its own index, every further one comes
    // from a device counter. Waves do not take equally long (per-wave time spreads by ~ +-15 %), and with
    // a static stride the launch ends when the unluckiest wave has finished ALL its groups.
    for (long g0 = (long)wave_id * G; g0 < Bv;) {
        const long cwi = g0 + grp;                 // position in the work list
        const bool valid = (cwi < Bv);
        const long cw = (p.cw_list && valid) ? (long)p.cw_list[cwi] : cwi;   // codeword (row of llr / out)
        guard = 0;
        gacc = __builtin_inf();
        auto cw_of_lane = [&](int ln) -> size_t {
            const long i = g0 + (LAT ? 0 : ln / GS);
            return p.cw_list ? (size_t)p.cw_list[i < Bv ? i : Bv - 1] : (size_t)i;     // (lanes past the end of the work list: any valid row)
        };
        auto FN2 = [&](double a0_, double b0_, double a1_, double b1_, double &r0_, double &r1_) {
            if constexpr (ED) { r0_ = f_node_e_acc(a0_, b0_, gacc); r1_ = f_node_e_acc(a1_, b1_, gacc); }
            else f_node2(a0_, b0_, a1_, b1_, tb, r0_, r1_);
        };

        // initializeDataStructures + assignInitialPath (PolarCode.cpp:195-272): the inactive
        // stack holds 0..L-1, the first pop (initial path) is L-1.
        bool active = valid && (lig == L - 1);
        u64 actw = __ballot(active);               // wave mask of `active`, refreshed where it changes (initial path, kill / clone)
        double pm = 0.0;
        int sp = L - 1;                            // group-uniform stack pointer
        if (lig < L - 1) stackv[gbase + lig] = (unsigned char)lig;
        P16 pL = {0, 0};                           // LLR slot pointer per layer (index sh = n - lam)
        P16 pC = {0, 0};                           // column-0 C slot pointer for big layers (index sh)
        u64 clsmall = 0;                           // column-0 partial sums of layers with S <= 32: bits [S, 2S)
        uint32_t hword = 0;                        // decisions of the current 32 unfrozen steps
        int origin = lig;                          // slot that holds this path's flushed history
        unsigned t = 0;                            // unfrozen steps so far (wave-uniform)
        if constexpr (LAT) {
            // channel row -> stored form (or the plain LLRs for the LLR-domain arithmetic), kernel element order: element e is channel
            // position bitrev_n(e), so that the pair (2b, 2b + 1) the top layer combines sits at (j, j + N/2); input guard of ed_front_kernel
            bool any = false, sized = false;
            const size_t row = (size_t)(valid ? cw : 0) * (size_t)N;
            for (int i0 = 0; i0 < N; i0 += 64 * 8) {
                double x[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int i = i0 + 64 * k + lane;
                    x[k] = (i < N) ? (p.llr_f32 ? (double)reinterpret_cast<const float *>(p.llr)[row + i] : p.llr[row + i]) : 1.0;
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int i = i0 + 64 * k + lane;
                    double v = x[k];
                    if constexpr (ED) { bool f; v = ed_from_channel(x[k], tb, f); any |= f; sized |= fabs(x[k]) >= 0.1; }
                    if (i < N) lat_ch[__brev((unsigned)i) >> (32 - n)] = v;
                }
            }
            if constexpr (ED) { if (wave_any(any) || !wave_any(sized)) guard = ~0ull; }
        }
        wave_mem_fence();

        // ================= all-frozen prefix (computed by prefix_kernel) =================
        // Until the first unfrozen position only ONE path exists per codeword and every decision is the
        // frozen 0: prefix_kernel has already produced, per codeword, the f-chain of the layers above
        // the prefix block (node 0, sizes N/2..Q, contiguous in p.pre) and the path metric of the first
        // Pe leaves. The walk resumes at phi = Pe; a layer of size 2S >= Q is read from that buffer
        // (same addresses for every path of the codeword: broadcast) until its first rewrite at phi = 2S.
        int phi_start = 0, forced_top = 0;
        const double *pre_cw = nullptr;
        if (p.prefix_q > 0) {
            const int Q = p.prefix_q, Pe = p.prefix_len;
            pre_cw = p.pre + (size_t)(valid ? cw : 0) * (size_t)(N - Q + 1);
            if (active) {
                pm = pre_cw[0];
                // zero partial sums of every completed (all-frozen) left subtree
                for (int S = 64; S <= Q && S <= N / 2; S <<= 1) {
                    uint32_t *cz = g_cl + (size_t)(S / 32 - 2) * 64 + lane;
                    for (int w = 0; w < S / 32; ++w) cz[(size_t)w * 64] = 0u;
                    pC.set(__builtin_ctz((unsigned)S), lig);
                }
            }
            wave_mem_fence();
            phi_start = Pe;
            if (Pe < Q) forced_top = n - __builtin_ctz((unsigned)Q) + 1;     // recompute from the block downwards
        }

        PROF_DECL
#ifdef POLAR_MARGIN
        // development aid (tools/margin_profile.py): the smallest gap, over all pruning fork steps of this codeword, between
        // the worst surviving and the best discarded fork metric — what a lower-precision state would have to resolve
        double mingap = __builtin_inf();
#endif
        // per-leaf control word (frozen flag, rate-0 block size): a SCALAR load through the constant
        // address space, issued one leaf ahead — as a plain global load it is a vector memory round
        // trip on the critical path of every leaf
        typedef const uint32_t __attribute__((address_space(4))) *kconst_u32;
        const kconst_u32 ctlp = (kconst_u32)(uintptr_t)p.ctl;
        uint32_t ctl_next = ctlp[phi_start];
        for (int phi = phi_start; phi < N; ++phi) {
            PROF(0)
            const uint32_t ctl = ctl_next;
            // recursivelyUpdateC (PolarCode.cpp:457-473) from the layer of size S upwards: X = column 1 of
            // that layer (the S bits just completed by a RIGHT child with node index ph)
            auto update_c = [&](int S, uint32_t X, int ph) {
                LANE_CTX
                for (;;) {
                    if (4 * S > N) break;                   // C_0 is never read (PolarCode.cpp writes it, nobody uses it)
                    const int psi = ph >> 1;
                    const bool to_right = (psi & 1);        // result becomes column 1 of C_{lam-1}
                    const int sh = __builtin_ctz((unsigned)S);
                    if (S <= 16) {
                        uint32_t cl = (uint32_t)(clsmall >> S) & ((1u << S) - 1u);
                        uint32_t nw = (cl ^ X) | (X << S);   // 2S bits
                        if (!to_right) {
                            const int S2 = 2 * S;
                            const u64 m = ((S2 == 32) ? 0xFFFFFFFFull : ((1ull << S2) - 1ull)) << S2;
                            if (active) clsmall = (clsmall & ~m) | ((u64)nw << S2);
                        }
                        X = nw;
                    } else if (S == 32) {
                        uint32_t cl = (uint32_t)(clsmall >> 32);
                        uint32_t *dst = (to_right ? g_cr : g_cl) + (size_t)0 * CST + POLAR_CL;   // layer size 64 -> word offset 0
                        if (active) { dst[0] = cl ^ X; dst[CST] = X; }
                        if (!to_right && active) pC.set(sh + 1, lig);
                    } else {
                        const int nwd = S / 32;
                        const uint32_t *cl = g_cl + (size_t)(nwd - 2) * CST + POLAR_CGB + pC.get(sh);
                        const uint32_t *cr = g_cr + (size_t)(nwd - 2) * CST + POLAR_CL;
                        uint32_t *dst = (to_right ? g_cr : g_cl) + (size_t)(2 * nwd - 2) * CST + POLAR_CL;
                        if (active) {
                            // all loads of a chunk first, then the stores: one memory round trip per chunk instead of
                            // one per word (a load behind a store waits for the store's acknowledgement as well)
                            auto chunk = [&](auto CH_) {
                                constexpr int CH = decltype(CH_)::value;
                                for (int w = 0; w < nwd; w += CH) {
                                    uint32_t r[CH], l[CH];
#pragma unroll
                                    for (int i = 0; i < CH; ++i) { r[i] = cr[(size_t)(w + i) * CST]; l[i] = cl[(size_t)(w + i) * CST]; }
#pragma unroll
                                    for (int i = 0; i < CH; ++i) { dst[(size_t)(w + i) * CST] = l[i] ^ r[i]; dst[(size_t)(w + i + nwd) * CST] = r[i]; }
                                }
                            };
                            if (nwd >= 8) chunk(std::integral_constant<int, 8>{});
                            else if (nwd == 4) chunk(std::integral_constant<int, 4>{});
                            else chunk(std::integral_constant<int, 2>{});
                            if (!to_right) pC.set(sh + 1, lig);
                        }
                    }
                    wave_mem_fence();
                    if (!to_right) break;
                    S *= 2;
                    ph = psi;
                }
            };
            // all-frozen aligned block of 2^zb leaves starting here (host schedule), 0 = ordinary leaf
            const int zb = (int)(ctl >> 1) & 0x7F;
            {
                const int nphi = phi + (1 << zb);
                ctl_next = ctlp[nphi < N ? nphi : 0];
            }
            const int lam_stop = n - zb;
            // ---------------- recursivelyCalcLLR(n, phi): PolarCode.cpp:422-455 ----------------
            const int lam_top = (phi == phi_start && forced_top) ? forced_top : (phi ? (n - __builtin_ctz((unsigned)phi)) : 1);
            double leaf = 0.0;
            for (int lam = lam_top; lam <= lam_stop; ++lam) {
                if constexpr (LAT) {
                    // ---- one layer, its elements spread over the 64 / GS lanes of each path (see the template's comment)
                    LANE_CTX
                    const int sh_ = n - lam, S_ = 1 << sh_, e_ = lane / GS;
                    const bool odd_ = (phi >> sh_) & 1;
                    const int pin_ = (lam > 1) ? pL.get(sh_ + 1) : 0;
                    const double *srcp = (lam > 1) ? lat_a + (size_t)(2 * S_ - 1) * GS + pin_ : lat_ch;
                    const int sstr = (lam > 1) ? GS : 1;
                    double *dstp = lat_a + (size_t)(S_ - 1) * GS + lig;
                    const uint32_t *cwp_ = (odd_ && S_ > 32) ? g_cl + (size_t)(S_ / 32 - 2) * CST + pC.get(sh_) : nullptr;
                    auto one = [&](int j, double a_, double b_) -> double {
                        if (!odd_) return FN(a_, b_);
                        if (S_ <= 32) return GN(a_, b_, (uint32_t)(clsmall >> S_), j);
                        return GN(a_, b_, cwp_[(size_t)(j >> 5) * CST], j & 31);
                    };
                    if (S_ >= 4 * EL) {
                        for (int j0 = e_; j0 < S_; j0 += 4 * EL) {
                            double a_[4], b_[4], r_[4];
#pragma unroll
                            for (int k = 0; k < 4; ++k) { a_[k] = srcp[(size_t)(j0 + k * EL) * sstr]; b_[k] = srcp[(size_t)(j0 + k * EL + S_) * sstr]; }
#pragma unroll
                            for (int k = 0; k < 4; ++k) r_[k] = one(j0 + k * EL, a_[k], b_[k]);
                            if (active) {
#pragma unroll
                                for (int k = 0; k < 4; ++k) dstp[(size_t)(j0 + k * EL) * GS] = r_[k];
                            }
                        }
                    } else {
                        for (int j = e_; j < (S_ > EL ? S_ : EL); j += EL) {
                            const bool in_ = j < S_;
                            const double a_ = srcp[(size_t)(in_ ? j : 0) * sstr], b_ = srcp[(size_t)((in_ ? j : 0) + S_) * sstr];
                            const double r_ = one(in_ ? j : 0, a_, b_);
                            if (active && in_) dstp[(size_t)j * GS] = r_;
                        }
                    }
                    if (active) pL.set(sh_, lig);
                    wave_mem_fence();
                    if (S_ == 1) leaf = lat_a[lig];            // (every lane of the path: the element-0 lane wrote it)
                    PROF(S_ >= EL ? (odd_ ? 1 : 2) : (S_ >= 4 ? 3 : 4))
                    continue;
                }
                if (POLAR_UNLIKELY2(tbl && lam <= 2 && phi >= S2)) {
                    // phi = N/4, N/2, 3N/4: the visits of layers 1 and 2 are replaced by the table build
                    const int kind = phi / S2;                      // 1: h (g of the shared layer 1), 2: f, 3: g
                    LANE_CTX
                    double *Xc = tab_w + (size_t)(lane / GS) * (size_t)(3 * N), *T2c = Xc + N;
                    if (valid) {                                    // all lanes of the codeword, whatever their path's state
                        if (kind == 1) {
                            const double *x1f = p.pre + cw_of_lane(lane) * (size_t)(N - p.prefix_q + 1) + 1;     // layer 1, first half (prefix kernel)
                            for (int j = lig; j < S2; j += GS) {
                                const double a = x1f[j], b = x1f[j + S2];
                                T2c[8 * j + 0] = g_node_e(a, b, 0u, tb);
                                T2c[8 * j + 1] = g_node_e(a, b, 0x80000000u, tb);
                            }
                        } else {
                            if (kind == 2) {
                                const double *chr = ch_row<ED>(p, cw_of_lane(lane), N);
                                for (int e = lig; e < S1; e += GS) {
                                    const unsigned i0 = __brev((unsigned)e) >> (32 - n);
                                    const double a = CH(chr, i0), b = CH(chr, i0 + 1);
                                    Xc[2 * e + 0] = g_node_e(a, b, 0u, tb);
                                    Xc[2 * e + 1] = g_node_e(a, b, 0x80000000u, tb);
                                }
                                wave_mem_fence();
                            }
                            for (int j = lig; j < S2; j += GS) {
                                const double a0 = Xc[2 * j], a1 = Xc[2 * j + 1], b0 = Xc[2 * (j + S2)], b1 = Xc[2 * (j + S2) + 1];
                                if (kind == 2) {
                                    T2c[8 * j + 0] = f_node_e(a0, b0, guard); T2c[8 * j + 1] = f_node_e(a1, b0, guard);
                                    T2c[8 * j + 2] = f_node_e(a0, b1, guard); T2c[8 * j + 3] = f_node_e(a1, b1, guard);
                                } else {
#pragma unroll
                                    for (int v = 0; v < 8; ++v)
                                        T2c[8 * j + v] = g_node_e((v & 1) ? a1 : a0, (v & 2) ? b1 : b0, (unsigned)(v >> 2) << 31, tb);
                                }
                            }
                        }
                    }
                    // the path's variant nibbles V[e] = u1[e] | u1[e + N/4] << 1 | u2[e] << 2 (kind 1: u2[e] only), 8 per word
                    // (packing them per consumer pass instead — two words per pass — costs more in the build than the
                    // visit saves: -5 %)
                    if (active) {
                        const uint32_t *c1p = g_cl + (size_t)(S1 / 32 - 2) * 64 + gbase + pC.get(n - 1);
                        const uint32_t *c2p = g_cl + (size_t)(S2 / 32 - 2) * 64 + gbase + pC.get(n - 2);
                        auto spread = [](uint32_t x) {               // bit i of the low byte -> bit 4i
                            uint32_t t = (x | (x << 12)) & 0x000F000Fu;
                            t = (t | (t << 6)) & 0x03030303u;
                            return (t | (t << 3)) & 0x11111111u;
                        };
                        for (int w32 = 0; w32 < S2 / 32; ++w32) {
                            const uint32_t ua = (kind >= 2) ? c1p[(size_t)w32 * 64] : 0u, ub = (kind >= 2) ? c1p[(size_t)(w32 + S2 / 32) * 64] : 0u;
                            const uint32_t uc = (kind != 2) ? c2p[(size_t)w32 * 64] : 0u;
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                uint32_t vw;
                                if (kind == 1) vw = spread((uc >> (8 * q)) & 0xFFu);
                                else vw = spread((ua >> (8 * q)) & 0xFFu) | (spread((ub >> (8 * q)) & 0xFFu) << 1) | (spread((uc >> (8 * q)) & 0xFFu) << 2);
                                g_v[(size_t)(4 * w32 + q) * 64 + lane] = vw;
                            }
                        }
                        pC.set(15, lig);                             // slot of this path's V words
                    }
                    wave_mem_fence();
                    lam = 2;
                    continue;                                        // next: layer 3 with the table as its source
                }
                const int sh = n - lam;
                const int S = 1 << sh;
                const bool odd = (phi >> sh) & 1;
                // ---- fused visit of two consecutive layers (lam: size S, op f/g; lam+1: size S/2, always
                // f right after): the values of layer lam are written (the later g-visit of lam+1 needs
                // them) but NOT re-read from HBM for the f-visit of lam+1. Only when the source of lam
                // is HBM-resident (channel LLRs or a scratch layer).
#ifdef POLAR_SLOTHIST
                // Measurement build (tools/slot_histogram.py; round-3 verdict item 2): for every visit whose SOURCE layer is
                // HBM-resident, how many DISTINCT source slots the active paths of a codeword read (pL.get(sh + 1)) — per
                // visit kind (f / g) and layer size. An f-visit whose source slots coincide would compute and store identical
                // rows. hist[kind][sh][distinct] += 1, plus sum of active paths and of paths reading their OWN slot.
                if (GS == 32 && lam > 1 && 2 * S > SL && p.pm_out) {
                    LANE_CTX
                    const bool in_pre_ = p.prefix_q > 0 && 2 * S >= p.prefix_q && phi < 2 * S;
                    const bool tab_ = tbl && lam == 3 && phi >= S2;
                    if (!in_pre_ && !tab_) {
                        const int pin_ = pL.get(sh + 1);
                        int d0 = 0, d1 = 0;
                        for (int s_ = 0; s_ < 32; ++s_) {
                            const u64 m_ = __ballot(active && pin_ == s_);
                            d0 += (m_ & 0xFFFFFFFFull) != 0; d1 += (m_ >> 32) != 0;
                        }
                        const u64 own_ = __ballot(active && pin_ == lig);
                        u64 *hb = reinterpret_cast<u64 *>(p.pm_out) + 64 + (size_t)((odd ? 1 : 0) * 12 + sh) * 40;
                        if (lane == 0) {
                            if (actw & 0xFFFFFFFFull) { atomicAdd(hb + d0, 1ull); atomicAdd(hb + 34, (u64)__popcll(actw & 0xFFFFFFFFull)); atomicAdd(hb + 35, (u64)__popcll(own_ & 0xFFFFFFFFull)); }
                            if (actw >> 32) { atomicAdd(hb + d1, 1ull); atomicAdd(hb + 34, (u64)__popcll(actw >> 32)); atomicAdd(hb + 35, (u64)__popcll(own_ >> 32)); }
                        }
                    }
                }
#endif
                if (!PIPE && S >= 8 && 2 * S > SL && lam + 1 <= lam_stop && ((phi >> (sh - 1)) & 1) == 0) {   // (lam+1 is an f-visit)
                    const int H = S / 2;
#ifdef POLAR_SADDR
                    // (not from the prefix buffer, whose rows are per codeword: once per codeword, left to the two-layer body)
                    const bool deep = (lam + 3 <= lam_stop) && (phi & (S - 1)) == 0 && !(lam > 1 && p.prefix_q > 0 && 2 * S >= p.prefix_q && phi < 2 * S);
#else
                    const bool deep = (lam + 3 <= lam_stop) && (phi & (S - 1)) == 0;        // four layers at once (else two)
#endif
                    if (active) {
                        LANE_CTX
                        const bool in_is_ch = (lam == 1);
                        const bool in_pre = !in_is_ch && p.prefix_q > 0 && 2 * S >= p.prefix_q && phi < 2 * S;
                        // (active lanes are valid ones: codeword g0 + lane / GS)
                        const double *in0 = in_is_ch ? ch_row<ED>(p, cw_of_lane(lane), N) : nullptr;
                        const double *pre_cw = in_pre ? p.pre + cw_of_lane(lane) * (size_t)(N - p.prefix_q + 1) : nullptr;
                        const int pin = (in_is_ch || in_pre) ? 0 : pL.get(sh + 1);
                        const size_t istr = in_pre ? 1 : 64;      // prefix layers are contiguous per codeword
                        const double *gin = in_is_ch ? nullptr : (in_pre ? pre_cw + 1 + (size_t)(N - 4 * S)
                                                                         : g_llr + (size_t)(2 * S - 2 * SL) * 64 + gbase + pin);
                        uint32_t cb0 = 0, cb1 = 0;          // partial-sum bits for elements j.. and j+H..
                        const uint32_t *cwp = nullptr;
                        if (odd) {
                            if (S <= 32) { cb0 = (uint32_t)(clsmall >> S); cb1 = cb0 >> H; }
                            else cwp = g_cl + (size_t)(S / 32 - 2) * 64 + gbase + pC.get(sh);
                        }
                        // the body is instantiated once per address-space combination of its outputs: a pointer
                        // that may be LDS or global degrades every access to a FLAT instruction, which waits on
                        // vmcnt AND lgkmcnt (i.e. for every outstanding HBM store) before a dependent use
                        auto fused_body = [&](const double *inp, double *out0, double *out1) {
                            for (int j = 0; j < H; j += FU) {
                                double a0[FU], b0[FU], a1[FU], b1[FU];
                                if (in_is_ch) {
#pragma unroll
                                    for (int k = 0; k < FU; ++k) {
                                        unsigned i0 = __brev((unsigned)(j + k)) >> (32 - n);
                                        unsigned i1 = __brev((unsigned)(j + k + H)) >> (32 - n);
                                        a0[k] = CH(in0, i0); b0[k] = CH(in0, i0 + 1);
                                        a1[k] = CH(in0, i1); b1[k] = CH(in0, i1 + 1);
                                    }
                                } else {
#pragma unroll
                                    for (int k = 0; k < FU; ++k) {
                                        a0[k] = inp[(size_t)(j + k) * istr];
                                        b0[k] = inp[(size_t)(j + k + S) * istr];
                                        a1[k] = inp[(size_t)(j + k + H) * istr];
                                        b1[k] = inp[(size_t)(j + k + H + S) * istr];
                                    }
                                }
                                if (odd && S > 32 && (j & 31) == 0) {
                                    cb0 = cwp[(size_t)(j >> 5) * 64];
                                    cb1 = (H >= 32) ? cwp[(size_t)((j + H) >> 5) * 64] : (cb0 >> H);
                                }
                                // element by element: (x0, x1) of layer lam -> stored -> y of layer lam+1 -> stored
#pragma unroll
                                for (int k = 0; k < FU; ++k) {
                                    double x0, x1;
                                    if (odd) {
                                        const int bi = (S > 32) ? ((j + k) & 31) : (j + k);
                                        x0 = GN(a0[k], b0[k], cb0, bi);
                                        x1 = GN(a1[k], b1[k], cb1, bi);
                                    } else {
                                        x0 = FN(a0[k], b0[k]);
                                        x1 = FN(a1[k], b1[k]);
                                    }
                                    out0[(size_t)(j + k) * 64] = x0;
                                    out0[(size_t)(j + k + H) * 64] = x1;
                                    out1[(size_t)(j + k) * 64] = FN(x0, x1);
                                }
                            }
                        };
                        // ---- four layers in one pass (lam .. lam+3, sizes S, S/2, S/4, S/8): element j of the lowest
                        // one is a 3-stage f-tree over the eight elements j + m*S/8 of layer lam, so none of the three
                        // intermediate layers is re-read from HBM by the f-visit below it (they are still written:
                        // the later g-visits need them). 16 loads in flight per pass, as the two-layer body.
                        // (measured and dropped: not storing layer 1's second half — the largest array — and re-deriving it
                        // from the channel values at its only later reader, the g-visit of layer 2 at phi = 3N/4: -9 % HBM
                        // bytes, but -1.7 % throughput; the re-derivation pass itself is slower than the traffic it saves)
                        const bool tsrc = tbl && lam == 3 && phi >= S2;       // inputs come from the layer-2 table
                        // GMM: bit k set = o_k is HBM-resident and passed as the UNIFORM base of its rows (the lane offset is added as a
                        // 32-bit register offset: global_load/store with an SGPR base, no 64-bit VALU add per access)
                        auto fused4 = [&](const double *inp, double *o0, double *o1, double *o2, double *o3, auto NTT, auto TSS, auto GMM) {
                            constexpr bool TS = decltype(TSS)::value;
                            constexpr int GM = decltype(GMM)::value;
                            const uint32_t lb = (uint32_t)lane * 8u;
                            const uint32_t lin = (uint32_t)(gbase + pin) * 8u;
                            const char *inu = reinterpret_cast<const char *>(g_llr + (size_t)(2 * S - 2 * SL) * 64);      // source rows (uniform)
                            auto st = [&](double *o, auto gbit, auto ntbit, size_t row, double val) {
                                constexpr bool G_ = decltype(gbit)::value, N_ = decltype(ntbit)::value;
                                double *q_ = G_ ? reinterpret_cast<double *>(reinterpret_cast<char *>(o) + row * 512 + lb) : o + row * 64;
                                if (N_) __builtin_nontemporal_store(val, q_); else *q_ = val;
                            };
                            // streaming (non-temporal) accesses for the layers of size >= 64 (bit 0: input, 1: o0, 2: o1): they
                            // are written once and read once or twice much later; the layers of size 16 and 32 stay cacheable
                            // (measured +1.7 %)
                            constexpr int NT = decltype(NTT)::value;
                            const int E = S >> 3;
                            uint32_t cw8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                            if (odd && S <= 32) cw8[0] = (uint32_t)(clsmall >> S);
                            double a[8], b[8], v[8];
                            const double *T2c = TS ? tab_w + (size_t)(lane / GS) * (size_t)(3 * N) + N : nullptr;
                            const uint32_t *vp = TS ? g_v + gbase + pC.get(15) : nullptr;
                            auto load8 = [&](int j) {
                                if (TS) {
#pragma unroll
                                    for (int m = 0; m < 8; ++m) {
                                        const int e0 = j + m * E, e1 = e0 + S;            // elements of layer 2
                                        const uint32_t w0 = vp[(size_t)(e0 >> 3) * 64], w1 = vp[(size_t)(e1 >> 3) * 64];
                                        a[m] = T2c[8 * e0 + ((w0 >> (4 * (e0 & 7))) & 7u)];
                                        b[m] = T2c[8 * e1 + ((w1 >> (4 * (e1 & 7))) & 7u)];
                                    }
                                } else if (in_is_ch) {
#pragma unroll
                                    for (int m = 0; m < 8; ++m) {
                                        const unsigned i0 = __brev((unsigned)(j + m * E)) >> (32 - n);
                                        a[m] = CH(in0, i0); b[m] = CH(in0, i0 + 1);
                                    }
                                } else {
                                    const int jj = j;
#pragma unroll
                                    for (int m = 0; m < 8; ++m) {
#ifdef POLAR_SADDR
                                        // (never the prefix buffer here: `deep` excludes it)
                                        const double *pa = reinterpret_cast<const double *>(inu + (size_t)(jj + m * E) * 512 + lin);
                                        const double *pb = reinterpret_cast<const double *>(inu + (size_t)(jj + m * E + S) * 512 + lin);
                                        if (NT & 1) { a[m] = __builtin_nontemporal_load(pa); b[m] = __builtin_nontemporal_load(pb); }
                                        else { a[m] = *pa; b[m] = *pb; }
#else
                                        if (NT & 1) { a[m] = __builtin_nontemporal_load(inp + (size_t)(jj + m * E) * istr); b[m] = __builtin_nontemporal_load(inp + (size_t)(jj + m * E + S) * istr); }
                                        else { a[m] = inp[(size_t)(jj + m * E) * istr]; b[m] = inp[(size_t)(jj + m * E + S) * istr]; }
#endif
                                    }
                                }
                            };
                            // (measured and dropped, round 3: an L2 prefetch of the NEXT pass's 16 source rows by ONE scattered
                            // global_load_dword touching their 64 lines — no VGPRs, no wait: -1.5 %; the kernel is on the bandwidth
                            // ceiling of its access pattern, asking earlier gains nothing)
                            for (int j = 0; j < E; ++j) {
                                load8(j);
                                if (odd && S > 32 && (j & 31) == 0) {
#pragma unroll
                                    for (int m = 0; m < 8; ++m) cw8[m] = cwp[(size_t)((j + m * E) >> 5) * 64];
                                }
#pragma unroll
                                for (int m = 0; m < 8; ++m) {
                                    if (odd) v[m] = (S > 32) ? GN(a[m], b[m], cw8[m], (j + m * E) & 31) : GN(a[m], b[m], cw8[0], j + m * E);
                                    else v[m] = FN(a[m], b[m]);
                                }
                                // (issuing the next pass's loads here, ahead of the 15 stores, was measured: -15 % — the
                                // double-buffered inputs do not fit the 128-VGPR budget)
                                const int js = j;
                                typedef std::integral_constant<bool, (GM & 1) != 0> G0; typedef std::integral_constant<bool, (GM & 2) != 0> G1;
                                typedef std::integral_constant<bool, (GM & 4) != 0> G2; typedef std::integral_constant<bool, (GM & 8) != 0> G3;
                                typedef std::integral_constant<bool, (NT & 2) != 0> N0; typedef std::integral_constant<bool, (NT & 4) != 0> N1;
                                typedef std::integral_constant<bool, (NT & 8) != 0> N2; typedef std::integral_constant<bool, false> N3;
#pragma unroll
                                for (int m = 0; m < 8; ++m) st(o0, G0{}, N0{}, (size_t)(js + m * E), v[m]);
#pragma unroll
                                for (int m = 0; m < 4; ++m) { v[m] = FN(v[m], v[m + 4]); st(o1, G1{}, N1{}, (size_t)(js + m * E), v[m]); }
#pragma unroll
                                for (int m = 0; m < 2; ++m) { v[m] = FN(v[m], v[m + 2]); st(o2, G2{}, N2{}, (size_t)(js + m * E), v[m]); }
                                v[0] = FN(v[0], v[1]);
                                st(o3, G3{}, N3{}, (size_t)j, v[0]);
                                leaf = v[0];            // (the leaf value when S/8 == 1)
                            }
                        };
#define POLAR_GROW(T) (g_llr + (size_t)((T) - 2 * SL) * 64 + lane)
#define POLAR_LROW(T) (lds_llr + (size_t)((T) - 1) * 64 + lane)
                        if (deep) {
                            const int Q = S / 4, E8 = S / 8;
                            // (also streaming the layer of size 32, so that only the layer of size 16 competes for the L2: -0.5 %)
#define POLAR_NTM(x) std::integral_constant<int, (x)>{}
                            typedef std::integral_constant<bool, false> TS0;
#ifdef POLAR_SADDR
#define POLAR_GOUT(T) (g_llr + (size_t)((T) - 2 * SL) * 64)
#define POLAR_GMK(x) std::integral_constant<int, (x)>{}
#else
#define POLAR_GOUT(T) POLAR_GROW(T)
#define POLAR_GMK(x) std::integral_constant<int, 0>{}
#endif
                            if (tsrc) fused4(gin, POLAR_GOUT(S), POLAR_GOUT(H), POLAR_GOUT(Q), POLAR_GOUT(E8), POLAR_NTM(6), std::integral_constant<bool, true>{}, POLAR_GMK(15));
                            else if (E8 > SL) fused4(gin, POLAR_GOUT(S), POLAR_GOUT(H), POLAR_GOUT(Q), POLAR_GOUT(E8), POLAR_NTM(7), TS0{}, POLAR_GMK(15));      // S >= 128
                            else if (Q > SL) fused4(gin, POLAR_GOUT(S), POLAR_GOUT(H), POLAR_GOUT(Q), POLAR_LROW(E8), POLAR_NTM(3), TS0{}, POLAR_GMK(7));   // S = 64
                            else if (H > SL) fused4(gin, POLAR_GOUT(S), POLAR_GOUT(H), POLAR_LROW(Q), POLAR_LROW(E8), POLAR_NTM(1), TS0{}, POLAR_GMK(3));   // S = 32: input 64
                            else if (S > SL) fused4(gin, POLAR_GOUT(S), POLAR_LROW(H), POLAR_LROW(Q), POLAR_LROW(E8), POLAR_NTM(0), TS0{}, POLAR_GMK(1));                 // S = 16: input 32
                            else fused4(gin, POLAR_LROW(S), POLAR_LROW(H), POLAR_LROW(Q), POLAR_LROW(E8), std::integral_constant<int, 0>{}, TS0{}, POLAR_GMK(0));        // S = 8: input 16
#undef POLAR_GOUT
#undef POLAR_GMK
#undef POLAR_NTM
                            pL.set(sh - 2, lig);
                            pL.set(sh - 3, lig);
                        }
                        else if (S <= SL) fused_body(gin, POLAR_LROW(S), POLAR_LROW(H));
                        else if (H <= SL) fused_body(gin, POLAR_GROW(S), POLAR_LROW(H));
                        else fused_body(gin, POLAR_GROW(S), POLAR_GROW(H));
#undef POLAR_GROW
#undef POLAR_LROW
                        pL.set(sh, lig);
                        pL.set(sh - 1, lig);
                    }
                    wave_mem_fence();
                    PROF(odd ? 1 : 2)
#ifdef POLAR_SLOTHIST
                    if (GS == 32 && p.pm_out && lane == 0 && actw) {      // f-visits of this pass fed from registers (the path's own values)
                        u64 *hb = reinterpret_cast<u64 *>(p.pm_out) + 64 + (size_t)24 * 40;
                        for (int d_ = 1; d_ <= (deep ? 3 : 1); ++d_) if ((S >> d_) > SL) atomicAdd(hb + (sh - d_), 1ull);
                    }
#endif
                    lam += deep ? 3 : 1;          // layers lam+1 (.. lam+3) are done
                    continue;
                }
                // ---- both layers in LDS (the bottom of the tree, visited at almost every leaf): plain ds_read /
                // ds_write on provably-LDS pointers, all inputs of the visit loaded before the first f
                if (lam > 1 && 2 * S <= SL) {
                    // the f-visits of the layers below follow immediately (phi is a multiple of S): they are taken
                    // from registers in the same pass — `below` more layers, down to the leaf or the rate-0 block
                    // (not at the resume point of the all-frozen prefix, where phi is not a multiple of S)
                    const int below = ((phi & (S - 1)) == 0) ? lam_stop - lam : 0;
                    if (active) {
                        LANE_CTX
                        const double *li = lds_llr + (size_t)(2 * S - 1) * 64 + gbase + pL.get(sh + 1);
                        double *lo = lds_llr + (size_t)(S - 1) * 64 + lane;
                        const uint32_t cb = odd ? (uint32_t)(clsmall >> S) : 0u;
                        auto small = [&](auto SS) {
                            constexpr int S_ = decltype(SS)::value;
                            double a[S_], b[S_], r[S_];
#pragma unroll
                            for (int j = 0; j < S_; ++j) { a[j] = li[(size_t)j * 64]; b[j] = li[(size_t)(j + S_) * 64]; }
#pragma unroll
                            for (int j = 0; j < S_; ++j) r[j] = odd ? GN(a[j], b[j], cb, j) : FN(a[j], b[j]);
                            // (the layer of size 1 — the leaf value — is consumed from the register: nobody reads it back)
                            if (S_ > 1 || !POLAR_SKIP_L1) {
#pragma unroll
                                for (int j = 0; j < S_; ++j) lo[(size_t)j * 64] = r[j];
                            }
                            // layer of size T = S_ >> d from the one above (registers), stored for its later g-visit
                            auto down = [&](auto TT) {
                                constexpr int T = decltype(TT)::value;
                                double *lt = lds_llr + (size_t)(T - 1) * 64 + lane;
#pragma unroll
                                for (int j = 0; j < T; ++j) { r[j] = FN(r[j], r[j + T]); if (T > 1 || !POLAR_SKIP_L1) lt[(size_t)j * 64] = r[j]; }
                            };
                            if constexpr (S_ >= 2) { if (below >= 1) down(std::integral_constant<int, S_ / 2>{}); }
                            if constexpr (S_ >= 4) { if (below >= 2) down(std::integral_constant<int, S_ / 4>{}); }
                            if constexpr (S_ >= 8) { if (below >= 3) down(std::integral_constant<int, S_ / 8>{}); }
                            if constexpr (S_ >= 16) { if (below >= 4) down(std::integral_constant<int, S_ / 16>{}); }
                            leaf = r[0];            // (the leaf LLR when the chain reached the layer of size 1)
                        };
                        if (S == 1) small(std::integral_constant<int, 1>{});
                        else if (S == 2) small(std::integral_constant<int, 2>{});
                        else if (S == 4) small(std::integral_constant<int, 4>{});
                        else if (S == 8) small(std::integral_constant<int, 8>{});
                        else small(std::integral_constant<int, 16>{});      // (lds_log = 5)
                        for (int d = 0; d <= below; ++d) if (sh - d > 0 || !POLAR_SKIP_L1) pL.set(sh - d, lig);
                    }
                    wave_mem_fence();
                    PROF(S >= 4 ? 3 : 4)
                    lam += below;
                    continue;
                }
                if (active) {
                    LANE_CTX
                    // input layer lam-1 (size 2S): 0 = channel LLRs, else scratch/LDS slot
                    const int pin = (lam > 1) ? pL.get(sh + 1) : 0;
                    const double *inp;   // element j at inp[j*istride]
                    size_t istride;
                    const bool in_is_ch = (lam == 1);
                    const bool in_pre = !in_is_ch && p.prefix_q > 0 && 2 * S >= p.prefix_q && phi < 2 * S;
                    const double *in0 = in_is_ch ? ch_row<ED>(p, cw_of_lane(lane), N) : nullptr;
                    const double *pre_cw = in_pre ? p.pre + cw_of_lane(lane) * (size_t)(N - p.prefix_q + 1) : nullptr;
                    constexpr bool in_lds = false;          // (LDS inputs were handled above)
                    istride = 64;
                    if (in_pre) { inp = pre_cw + 1 + (size_t)(N - 4 * S); istride = 1; }
                    else if (!in_is_ch) inp = g_llr + (size_t)(2 * S - 2 * SL) * 64 + gbase + pin;
                    else inp = nullptr;
                    const bool out_lds = (S <= SL);
                    auto generic_body = [&](double *outp) {
                    // partial sums for g (column 0 of C_lam)
                    uint32_t cbits = 0;
                    const uint32_t *cwp = nullptr;
                    if (odd) {
                        if (S <= 32) cbits = (uint32_t)(clsmall >> S);
                        else cwp = g_cl + (size_t)(S / 32 - 2) * 64 + gbase + pC.get(sh);
                    }
                    if (PIPE && S >= 16 && !in_lds) {
                        // source in HBM/L2 (channel LLRs or a scratch layer): software-pipelined,
                        // 8 elements (16 loads, 8 KiB per wave) in flight ahead of the compute
                        constexpr int U = 8;
                        double a0[U], b0[U], a1[U], b1[U];
                        auto load = [&](int j, double (&a)[U], double (&b)[U]) {
                            if (in_is_ch) {
#pragma unroll
                                for (int k = 0; k < U; ++k) {
                                    // position j <-> reference beta = bitrev_n(j); (j, j+N/2) <-> (2b', 2b'+1)
                                    unsigned idx = __brev((unsigned)(j + k)) >> (32 - n);
                                    a[k] = CH(in0, idx);
                                    b[k] = CH(in0, idx + 1);
                                }
                            } else {
#pragma unroll
                                for (int k = 0; k < U; ++k) {
                                    a[k] = inp[(size_t)(j + k) * istride];
                                    b[k] = inp[(size_t)(j + k + S) * istride];
                                }
                            }
                        };
                        auto comp = [&](int j, double (&a)[U], double (&b)[U]) {
                            double r[U];
                            if (odd) {
                                if (S > 32 && (j & 31) == 0) cbits = cwp[(size_t)(j >> 5) * 64];
#pragma unroll
                                for (int k = 0; k < U; ++k) r[k] = GN(a[k], b[k], cbits, (j + k) & 31);
                            } else {
#pragma unroll
                                for (int k = 0; k < U; ++k) r[k] = FN(a[k], b[k]);
                            }
#pragma unroll
                            for (int k = 0; k < U; ++k) outp[(size_t)(j + k) * 64] = r[k];
                        };
                        load(0, a0, b0);
                        for (int j = 0; j < S; j += 2 * U) {
                            load(j + U, a1, b1);
                            comp(j, a0, b0);
                            if (j + 2 * U < S) load(j + 2 * U, a0, b0);
                            comp(j + U, a1, b1);
                        }
                    } else if (S >= 4) {
                        for (int j = 0; j < S; j += 4) {
                            double a[4], b[4], r[4];
                            if (in_is_ch) {
#pragma unroll
                                for (int k = 0; k < 4; ++k) {
                                    unsigned idx = __brev((unsigned)(j + k)) >> (32 - n);
                                    a[k] = CH(in0, idx);
                                    b[k] = CH(in0, idx + 1);
                                }
                            } else {
#pragma unroll
                                for (int k = 0; k < 4; ++k) {
                                    a[k] = inp[(size_t)(j + k) * istride];
                                    b[k] = inp[(size_t)(j + k + S) * istride];
                                }
                            }
                            if (odd) {
                                if (S > 32 && (j & 31) == 0) cbits = cwp[(size_t)(j >> 5) * 64];
#pragma unroll
                                for (int k = 0; k < 4; ++k) r[k] = GN(a[k], b[k], cbits, (j + k) & 31);
                            } else {
#pragma unroll
                                for (int k = 0; k < 4; ++k) r[k] = FN(a[k], b[k]);
                            }
#pragma unroll
                            for (int k = 0; k < 4; ++k) outp[(size_t)(j + k) * 64] = r[k];
                        }
                    } else {
                        for (int j = 0; j < S; ++j) {
                            double a, b;
                            if (in_is_ch) {
                                unsigned idx = __brev((unsigned)j) >> (32 - n);
                                a = CH(in0, idx);
                                b = CH(in0, idx + 1);
                            } else {
                                a = inp[(size_t)j * istride];
                                b = inp[(size_t)(j + S) * istride];
                            }
                            double r = odd ? GN(a, b, cbits, j) : FN(a, b);
                            outp[(size_t)j * 64] = r;
                            leaf = r;
                        }
                    }
                    };
                    if (out_lds) generic_body(lds_llr + (size_t)(S - 1) * 64 + lane);
                    else generic_body(g_llr + (size_t)(S - 2 * SL) * 64 + lane);
                    pL.set(sh, lig);
                }
                wave_mem_fence();
                PROF(S > SL ? (odd ? 1 : 2) : (S >= 4 ? 3 : 4))
            }


            if (zb) {
                // ---- rate-0 block: Z = 2^zb consecutive frozen leaves whose subtree hangs off the layer of
                // size Z that was just computed. Every decision inside is the frozen 0, so the Z leaf LLRs
                // are a fixed f/g dataflow of that layer (g with u = 0): evaluated level by level in
                // registers (ILP Z/2) instead of Z sequential leaf steps; the path metric is then updated
                // leaf by leaf in order (PolarCode.cpp:475-487), and the block's partial sums (Z zeros) are
                // handed to the layer of size Z exactly as the last leaf's recursivelyUpdateC would.
                const int Z = 1 << zb;
                LANE_CTX
                // one 4-leaf sub-block: values v0..v3 of a size-4 node -> leaves (f,f) (f,g) (g,f) (g,g), then
                // the metric update of those four leaves in order
                auto block4 = [&](double v0, double v1, double v2, double v3) {
                    double lf[4] = {0, 0, 0, 0};
                    if (active) {
                        double a0, a1;
                        FN2(v0, v2, v1, v3, a0, a1);
                        const double b0 = GN(v0, v2, 0u, 0), b1 = GN(v1, v3, 0u, 0);
                        FN2(a0, a1, b0, b1, lf[0], lf[2]);
                        lf[1] = GN(a0, a1, 0u, 0); lf[3] = GN(b0, b1, 0u, 0);
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        bool ng; double alz, sneg, spos;
                        leaf_terms<ED>(lf[i], active, actw, tb, ng, alz, sneg, spos);
                        if (active) pm += ng ? spos : sneg;
                    }
                };
                // (one instantiation per address space of the source: a maybe-LDS-maybe-global pointer would
                // turn the loads into FLAT instructions that wait for every outstanding memory operation)
                auto rate0 = [&](const double *yp, auto STR_) {
                    constexpr int YS = decltype(STR_)::value;         // distance between the elements of the layer (LAT: GS)
                    if (zb == 3) {
                        double a[4] = {0, 0, 0, 0}, b[4] = {0, 0, 0, 0};
                        if (active) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                const double lo = yp[(size_t)j * YS], hi = yp[(size_t)(j + 4) * YS];
                                a[j] = FN(lo, hi);
                                b[j] = GN(lo, hi, 0u, 0);
                            }
                        }
                        block4(a[0], a[1], a[2], a[3]);
                        block4(b[0], b[1], b[2], b[3]);
                    } else {
                        double y[4] = {0, 0, 0, 0};
                        if (active) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) y[j] = yp[(size_t)j * YS];
                        }
                        block4(y[0], y[1], y[2], y[3]);
                    }
                };
                if constexpr (LAT) rate0(lat_a + (size_t)(Z - 1) * GS + lig, std::integral_constant<int, GS>{});
                else if (Z <= SL) rate0(lds_llr + (size_t)(Z - 1) * 64 + lane, std::integral_constant<int, 64>{});
                else rate0(g_llr + (size_t)(Z - 2 * SL) * 64 + lane, std::integral_constant<int, 64>{});
                const int nu = phi >> zb;                     // node index of the block at its layer
                if ((nu & 1) == 0) {
                    if (active) clsmall &= ~((((u64)1 << Z) - 1ull) << Z);   // column 0 of that layer := 0
                } else {
                    update_c(Z, 0u, nu);
                }
                wave_mem_fence();
                PROF(5)
                phi += Z - 1;
                continue;
            }
            // ---------------- leaf: frozen / unfrozen ----------------
            LANE_CTX
            const u64 below = (1ull << lig) - 1ull;
            const bool frozen = (ctl & 1u) != 0;      // wave-uniform
            unsigned ubit = 0;
            if (frozen) {
                // continuePaths_FrozenBit: PolarCode.cpp:475-487
                // PM += log(1+e^-llr): exactly 0 for llr >= 37, exactly |llr| (+0) for llr <= -37
                bool ng; double alz, sneg, spos;
                leaf_terms<ED>(leaf, active, actw, tb, ng, alz, sneg, spos);
                if (active) pm += ng ? spos : sneg;
            } else {
                // continuePaths_UnfrozenBit: PolarCode.cpp:489-607
                const u64 actm = (actw >> gbase) & gmask;
                const int nact = __popcll(actm);
                const int rho = (2 * nact < L) ? 2 * nact : L;
                // ---- fast path (exact): list full and every "good" fork (the bit the leaf LLR favours)
                // beats every "bad" fork of every path => the L survivors are the L good forks, nobody
                // is killed or cloned.  good metric = PM + log(1+e^-|llr|) (the same sum the general
                // path computes); bad metric = PM + log(1+e^|llr|) >= (PM + |llr|)(1 - 2^-40).
                bool lneg; double al;                      // llr < 0, |llr|
                double gm = -__builtin_inf(), bl = __builtin_inf();
                double sneg, spos;                         // log(1+e^-|llr|), log(1+e^|llr|)
                // (a logarithm-free lower bound of |llr| for this test — exponent and mantissa of E — was measured:
                // -1.5 %, the bound is short by up to 0.06 and sends more steps down the ranking path)
                leaf_terms<ED>(leaf, active, actw, tb, lneg, al, sneg, spos);
                if constexpr (ED) {
                    // a leaf the host marked as weak (control word bit 8: no construction for an ordinary channel leaves it
                    // unfrozen) that comes out below 1e-8: the reference decides on the rounding noise of its own arithmetic
                    // there, which the LLR-domain kernel follows much further down than this one -> fallback pass
                    if (POLAR_UNLIKELY2(ctl & 0x100u)) guard |= __ballot(active && fabs(leaf) > 0.99999999 && fabs(leaf) <= 1.0);
                }
                if (active) {
                    gm = pm + sneg;
                    bl = (pm + al) * 0.99999999999909050530;
                }
                // cheap sufficient test first: the metrics are non-negative doubles, so their HIGH words
                // order like unsigned integers; max / min of those over the group cost one DPP-fused integer
                // instruction per stage and no LDS traffic. Distinct high words decide gmax < bmin for
                // certain; only when they collide (|gmax - bmin| < 2^-20 relative) or the test fails are the
                // exact fp64 reductions run.
                bool fast;
                {
                    unsigned gh = active ? (unsigned)__double2hiint(gm) : 0u;
                    unsigned bh = active ? (unsigned)__double2hiint(bl) : 0xFFFFFFFFu;
                    group_max_min_u32<GS>(gh, bh);
                    const u64 m_ok = __builtin_amdgcn_sicmp(nact, 0, 32) | (__builtin_amdgcn_sicmp(nact, L, 32) & __builtin_amdgcn_uicmp(gh, bh, 36));   // EQ, EQ, ULT
                    fast = ((m_ok | ~group_result_rows<GS>()) == ~0ull);
                }
                double gmax = 0.0;
                if (!fast) {
                    gmax = group_reduce<GS, true>(gm, lane);
                    const double bmin = group_reduce<GS, false>(bl, lane);
                    fast = wave_all((nact == 0) || (nact == L && gmax < bmin));
                }
                PROF_CNT(8, 1)
                PROF(16)
                if (fast) {      // (marking this likely — the ranking path out of line — measured -0.5 %)
                    PROF_CNT(9, 1)
#ifdef POLAR_MARGIN
                    {
                        const double gx = group_reduce<GS, true>(active ? gm : -__builtin_inf(), lane);
                        const double bx = group_reduce<GS, false>(active ? pm + spos : __builtin_inf(), lane);
                        if (nact == L) mingap = __builtin_fmin(mingap, bx - gx);
                    }
#endif
                    if (active) {
                        ubit = lneg ? 1u : 0u;
                        pm = gm;
                        hword |= ubit << (t & 31);
                    }
                    PROF(17)
                } else {
                double pf0 = __builtin_nan(""), pf1 = __builtin_nan("");
                if (active) {
                    pf0 = -(pm + (lneg ? spos : sneg));     // -(PM + log(1+e^-llr)), PolarCode.cpp:505
                    pf1 = -(pm + (lneg ? sneg : spos));     // -(PM + log(1+e^llr)),  PolarCode.cpp:506
                }
                bool c0 = active, c1 = active;
                const bool need = (2 * nact > L);          // otherwise every fork continues
                const bool full = (nact == L);
                if (POLAR_LIKELY2(!wave_any(need && !full))) {
                    // List full (the usual case). Rank = number of better forks in the reference's order
                    // (metric desc = PM asc, fork index asc on ties, PolarCode.cpp:528-553). A bad fork
                    // whose lower bound is worse than every good fork (bl > gmax) can neither survive nor
                    // outrank a survivor, so only the L good forks and the few "competitive" bad forks
                    // are compared against: 32 LDS broadcasts + a short scalar loop instead of 64.
                    const bool goodbit = lneg;                       // bit the leaf LLR favours
                    const double mg = goodbit ? -pf1 : -pf0;                // PM of my good / bad fork
                    const double mb = goodbit ? -pf0 : -pf1;
                    const bool cbad = active && full && !(bl > gmax);
                    const u64 cbm = __ballot(cbad);
                    bool sg, sbd;
                    if constexpr (GS == 32) {
                        // Two groups of 32 lanes. (1) Every competitive bad fork is ranked on the SCALAR unit: its metric
                        // is read into SGPRs, four wave-wide compares give the masks of the good / competitive bad forks
                        // ordered before it (metric, then fork index 2*path + bit), a population count gives its rank.
                        // (2) The list stays at L, so k surviving bad forks displace the k WORST good forks of the
                        // group; those are peeled off by a DPP maximum over the metrics' high words (non-negative
                        // doubles order like their bit patterns), all lanes sharing the maximal high word at once
                        // when k allows, otherwise the low words and lane numbers decide, on the scalar unit.
                        const u64 actm64 = actw;
                        const u64 gbm = __ballot(goodbit);
                        u64 surv = 0;
                        PROF_CNT(10, 1)
                        PROF_CNT(11, __popcll(cbm))
                        for (u64 mi = cbm; mi; mi &= mi - 1) {
                            const int l_ = __builtin_ctzll(mi);
                            const double s_mb = readlane_d(mb, l_);
                            const u64 grp = 0xFFFFFFFFull << (l_ & 32);
                            const u64 blw = ((1ull << l_) - 1ull) & grp;              // paths below l_ in its group
                            const u64 self_first = ~gbm & (1ull << l_);              // own good fork is bit 0: lower index
                            const u64 lt_g = __builtin_amdgcn_fcmp(mg, s_mb, 4), eq_g = __builtin_amdgcn_fcmp(mg, s_mb, 1);
                            const u64 lt_b = __builtin_amdgcn_fcmp(mb, s_mb, 4), eq_b = __builtin_amdgcn_fcmp(mb, s_mb, 1);
                            const int r = __popcll((lt_g | (eq_g & (blw | self_first))) & grp & actm64) +
                                          __popcll((lt_b | (eq_b & blw)) & grp & cbm);
                            if (r < L) surv |= 1ull << l_;
                        }
                        PROF(18)
                        int k0 = __popcll(surv & 0xFFFFFFFFull), k1 = __popcll(surv >> 32);
                        u64 alive = actm64;
                        const unsigned khi = (unsigned)__double2hiint(mg) + 1u, klo = (unsigned)__double2loint(mg);
                        auto peel = [&](u64 e, int &k) {
                            const int ne = __popcll(e);
                            if (ne == 0) { k = 0; return; }
                            if (ne <= k) { alive &= ~e; k -= ne; return; }
                            for (; k > 0; --k) {                                       // equal high words: low word, then lane
                                int best = -1; unsigned blo = 0;
                                for (u64 q = e; q; q &= q - 1) {
                                    const int l_ = __builtin_ctzll(q);
                                    const unsigned lo_ = (unsigned)__builtin_amdgcn_readlane((int)klo, l_);
                                    if (best < 0 || lo_ >= blo) { best = l_; blo = lo_; }
                                }
                                e &= ~(1ull << best);
                                alive &= ~(1ull << best);
                            }
                        };
                        while (k0 | k1) {
                            const unsigned h0 = __builtin_amdgcn_inverse_ballot_w64(alive) ? khi : 0u;
                            unsigned h = h0;
                            group_max_u32<GS>(h);
                            const unsigned m0 = (unsigned)__builtin_amdgcn_readlane((int)h, 31), m1 = (unsigned)__builtin_amdgcn_readlane((int)h, 63);
                            const u64 eq = __ballot(h0 == (lane < 32 ? m0 : m1)) & alive;
                            if (k0) peel(eq & 0xFFFFFFFFull, k0);
                            if (k1) peel(eq & 0xFFFFFFFF00000000ull, k1);
                        }
                        sg = __builtin_amdgcn_inverse_ballot_w64(alive);
                        sbd = __builtin_amdgcn_inverse_ballot_w64(surv);
                        PROF_CNT(13, (__popcll(surv & 0xFFFFFFFFull) > __popcll(surv >> 32)) ? __popcll(surv & 0xFFFFFFFFull) : __popcll(surv >> 32))
                        PROF(19)
                    } else {
                        sortbuf[lane] = mg;
                        wave_mem_fence();
                        int rg = 0, rb = 0;
                        const double *sb = sortbuf + gbase;
                        // good fork of path i (index 2i + bit) vs my forks (indices 2*lig + ...): it precedes them
                        // on a tie exactly when i < lig (for i == lig see below), so the tie-break folds into the
                        // choice between "<=" and "<"; comparisons produce wave masks, combined on the scalar unit
#pragma unroll 8
                        for (int i = 0; i < GS; ++i) {
                            const double v = sb[i];
                            const u64 below_me = __ballot(lig > i);                    // lanes for which i < lig
                            const u64 lt_g = __builtin_amdgcn_fcmp(v, mg, 4), le_g = __builtin_amdgcn_fcmp(v, mg, 5);
                            const u64 lt_b = __builtin_amdgcn_fcmp(v, mb, 4), le_b = __builtin_amdgcn_fcmp(v, mb, 5);
                            rg += (int)__builtin_amdgcn_inverse_ballot_w64((le_g & below_me) | (lt_g & ~below_me));
                            rb += (int)__builtin_amdgcn_inverse_ballot_w64((le_b & below_me) | (lt_b & ~below_me));
                        }
                        // i == lig: my own good fork is never counted against itself (v < mg is false); against my
                        // bad fork the strict part (mg < mb) was counted above, a tie goes to the lower fork index
                        rb += (mg == mb && !goodbit) ? 1 : 0;
                        PROF(18)
                        // competitive bad forks (few at low SNR, up to all L when garbage paths fill the list):
                        // same scheme from the second half of the exchange buffer, iterations without a
                        // competitive bad fork in any group are skipped on the scalar unit
                        sortbuf[64 + lane] = mb;
                        wave_mem_fence();
                        const double *sbb = sortbuf + 64 + gbase;
                        u64 any_i = 0;                                       // bit i: some group has a competitive bad fork i
#pragma unroll
                        for (int g = 0; g < 64 / GS; ++g) any_i |= (cbm >> (g * GS)) & gmask;
                        PROF_CNT(10, 1)
                        PROF_CNT(11, __popcll(any_i))
#ifdef POLAR_PROFILE
                        {   // statistics only: good forks that some bad fork of their group could displace
                            const double bmin_true = group_reduce<GS, false>(active ? mb : __builtin_inf(), lane);
                            const u64 cgm = __ballot(active && !(mg < bmin_true));
                            u64 any_g = 0;
                            for (int g = 0; g < 64 / GS; ++g) any_g |= (cgm >> (g * GS)) & gmask;
                            PROF_CNT(12, __popcll(any_g))
                        }
#endif
                        for (u64 mi = any_i; mi; mi &= mi - 1) {
                            const int i = __builtin_ctzll(mi);
                            const double v = sbb[i];
                            const u64 mine = __ballot(((cbm >> gbase) >> i) & 1ull);      // lanes whose group's bad fork i competes
                            const u64 below_me = __ballot(lig > i);
                            const u64 lt_g = __builtin_amdgcn_fcmp(v, mg, 4), le_g = __builtin_amdgcn_fcmp(v, mg, 5);
                            const u64 lt_b = __builtin_amdgcn_fcmp(v, mb, 4), le_b = __builtin_amdgcn_fcmp(v, mb, 5);
                            rg += (int)__builtin_amdgcn_inverse_ballot_w64(((le_g & below_me) | (lt_g & ~below_me)) & mine);
                            rb += (int)__builtin_amdgcn_inverse_ballot_w64(((le_b & below_me) | (lt_b & ~below_me)) & mine);
                        }
                        // i == lig: my own bad fork against my good fork: mb < mg cannot hold, a tie goes to the
                        // lower fork index (the bad fork has the lower index when the good bit is 1)
                        rg += (cbad && mg == mb && goodbit) ? 1 : 0;
                        PROF(19)
#ifdef POLAR_PROFILE
                        {   // statistics only: surviving bad forks (= killed good forks) per group, max over the groups
                            const u64 sbm = __ballot(cbad && rb < L);
                            int kmax = 0, ktot = 0;
                            for (int g = 0; g < 64 / GS; ++g) { const int k_ = __popcll((sbm >> (g * GS)) & gmask); kmax = k_ > kmax ? k_ : kmax; ktot += k_; }
                            PROF_CNT(13, kmax)
                            PROF_CNT(14, kmax <= 4 ? 1 : 0)
                            PROF_CNT(15, kmax == 0 ? 1 : 0)
                        }
#endif
                        sg = active && (rg < L);
                        sbd = cbad && (rb < L);
                    }
                    if (full) {
                        c0 = goodbit ? sbd : sg;
                        c1 = goodbit ? sg : sbd;
                    }
#ifdef POLAR_MARGIN
                    {
                        const double ninf = -__builtin_inf(), pinf = __builtin_inf();
                        const double sv = __builtin_fmax(sg ? mg : ninf, sbd ? mb : ninf);
                        const double kv = __builtin_fmin((active && !sg) ? mg : pinf, (active && !sbd) ? mb : pinf);
                        const double sx = group_reduce<GS, true>(sv, lane), kx = group_reduce<GS, false>(kv, lane);
                        if (full) mingap = __builtin_fmin(mingap, kx - sx);
                    }
#endif
                    wave_mem_fence();
                } else {
                    sortbuf[2 * lane] = pf0;
                    sortbuf[2 * lane + 1] = pf1;
                    wave_mem_fence();
                    int r0 = 0, r1 = 0;
                    const double *sb = sortbuf + 2 * gbase;
                    const int i0 = 2 * lig;
                    // stable rank: value descending, fork index 2l+b ascending on ties
                    // (PolarCode.cpp:528-553: "> threshold" first, then "== threshold" in index order)
#pragma unroll 4
                    for (int i = 0; i < 2 * GS; ++i) {
                        const double v = sb[i];
                        r0 += (i < i0) ? (v >= pf0) : (v > pf0);
                        r1 += (i <= i0) ? (v >= pf1) : (v > pf1);
                    }
                    if (need) {
                        c0 = active && (r0 < rho);
                        c1 = active && (r1 < rho);
                    }
                    wave_mem_fence();
                }
                // kills (ascending l) push, then clones (ascending l) pop: PolarCode.cpp:555-570
                const bool kill = active && !c0 && !c1;
                const bool both = c0 && c1;
                const u64 km = (__ballot(kill) >> gbase) & gmask;
                const u64 bm = (__ballot(both) >> gbase) & gmask;
                srcof[lane] = (unsigned char)lig;
                if (POLAR_LIKELY2(wave_all(!active || full))) {
                    // list full before the step => #kills == #clones: the kills are pushed (ascending l) and
                    // popped right back (LIFO) by the clones in ascending l, i.e. the r-th cloner revives the
                    // r-th LARGEST killed index; stack pointer and the entries below are untouched. One LDS
                    // round trip: cloners post their index by rank, killed lanes pick theirs up.
                    if (both) stackv[gbase + __popcll(bm & below)] = (unsigned char)lig;           // rank r -> cloner
                    wave_mem_fence();
                    if (kill) srcof[lane] = stackv[gbase + __popcll(km >> 1 >> lig)];             // #killed above me = my rank from the top
                    wave_mem_fence();
                } else {
                    if (kill) stackv[gbase + sp + __popcll(km & below)] = (unsigned char)lig;
                    sp += __popcll(km);
                    wave_mem_fence();
                    if (both) {
                        int lp = stackv[gbase + sp - 1 - __popcll(bm & below)];
                        srcof[gbase + lp] = (unsigned char)lig;
                    }
                    sp -= __popcll(bm);
                    wave_mem_fence();
                }
                const int src = srcof[lane];
                const bool is_clone = (src != lig);
                PROF(20)
                // PM of the surviving forks: PM + log(1+exp(-+llr)) is the very sum whose negation
                // was ranked (PolarCode.cpp:580-582, 593, 601)
                double pm_new = c0 ? -pf0 : -pf1;
                ubit = c0 ? 0u : 1u;
                if (wave_any(is_clone)) {
                    const int sl = gbase + src;
                    double pm1 = shfl_d(-pf1, sl);
                    u64 a0 = shfl_u64(pL.lo, sl), a1 = shfl_u64(pL.hi, sl);
                    u64 b0 = shfl_u64(pC.lo, sl), b1 = shfl_u64(pC.hi, sl);
                    u64 cs = shfl_u64(clsmall, sl);
                    uint32_t hw = __shfl(hword, sl, 64);
                    int og = __shfl(origin, sl, 64);
                    if (is_clone) {
                        pm_new = pm1; ubit = 1u;
                        pL.lo = a0; pL.hi = a1; pC.lo = b0; pC.hi = b1;
                        clsmall = cs; hword = hw; origin = og;
                    }
                }
                active = (active && !kill) || is_clone;
                actw = __ballot(active);
                if (active) {
                    pm = pm_new;
                    hword |= ubit << (t & 31);
                } else {
                    pm = 0.0;   // killPath zeroes the metric (PolarCode.cpp:293-294)
                }
                PROF(21)
                }   // general path
                // every 32 unfrozen steps the decision word is stored together with the slot (`origin`) that
                // holds this path's previous word: a linked list per path, walked once at the end, so that
                // neither clones nor flushes ever copy history (the reference copies it on every clone,
                // PolarCode.cpp:574)
                if (POLAR_UNLIKELY2((t & 31) == 31)) {
                    const int w = (int)(t >> 5);
                    if (active) {
                        g_hist[(size_t)w * CST + POLAR_CL] = hword;
                        g_horg[(size_t)w * CST + POLAR_CL] = (uint32_t)origin;
                        origin = lig;
                        hword = 0;
                    }
                    wave_mem_fence();
                }
                ++t;
            }

            PROF(frozen ? 5 : 6)
            // ---------------- partial sums ----------------
            if ((phi & 1) == 0) {
                // left leaf: column 0 of C_n (size 1) lives at bit 1 of clsmall
                if (active) clsmall = (clsmall & ~2ull) | ((u64)ubit << 1);
            } else {
                update_c(1, ubit, phi);
            }
            PROF(7)
        }  // phi
        PROF_OUT

        // ---------------- last (partial) history word ----------------
        const int Wused = (int)((t + 31) >> 5);
        if ((t & 31) != 0 && active) {
            g_hist[(size_t)(Wused - 1) * CST + POLAR_CL] = hword;
            g_horg[(size_t)(Wused - 1) * CST + POLAR_CL] = (uint32_t)origin;
        }
        wave_mem_fence();

        // ---------------- findMostProbablePath + crc_check: PolarCode.cpp:609-644, 93-108 ----------------
        // crc_check walks the path's word list backwards: parity of (word & mask_i) per CRC row
        bool pass = true;
        if (p.crc > 0) {
            uint32_t acc = 0;
            if (active) {
                int cur = lig;
                for (int w = Wused - 1; w >= 0; --w) {
                    const uint32_t hw = g_hist[(size_t)w * CST + POLAR_CGB + cur];
                    cur = (int)(g_horg[(size_t)w * CST + POLAR_CGB + cur] & (GS - 1));
                    for (int i = 0; i < p.crc; ++i)
                        acc ^= (uint32_t)(__popc(hw & p.crc_mask[(size_t)i * p.W + w]) & 1) << i;
                }
            }
            pass = (acc == 0);
        }
        const u64 passm = (__ballot(active && pass) >> gbase) & gmask;
        const bool cand = active && (pass || passm == 0);      // :640-643 fall back to "no CRC"
        double key = (cand && pm < 1.7976931348623157e308) ? pm : __builtin_inf();
        int kidx = lig;
#pragma unroll
        for (int off = GS / 2; off >= 1; off >>= 1) {
            double ok = shfl_d(key, lane ^ off);
            int oi = __shfl(kidx, lane ^ off, 64);
            if (ok < key || (ok == key && oi < kidx)) { key = ok; kidx = oi; }
        }
        // no candidate with PM < DBL_MAX: the reference returns l_p = 0 (PolarCode.cpp:611,626)
        const int win = (key < __builtin_inf()) ? kidx : 0;
#ifdef POLAR_MARGIN
        double fingap;
        {   // final selection: runner-up candidate metric - winner's
            const double mine = (cand && pm < 1.7976931348623157e308 && lig != win) ? pm : __builtin_inf();
            fingap = group_reduce<GS, false>(mine, lane) - key;
        }
#endif
        const double pm_win = shfl_d(pm, gbase + win);
        // No candidate with a finite metric (every path met a frozen leaf with llr < -709.78: the reference's log(1+e^-llr)
        // is +inf there) AND the list never filled (more list entries than 2^K paths): the reference's l_p = 0 is a path that
        // was never activated, its info array still holds the zeros of initializeDataStructures (PolarCode.cpp:195-230).
        // (Found by tools/fuzz_parity.py; this read used to return whatever an earlier codeword left in the slot.)
        const bool win_active = __shfl((int)active, gbase + win, 64) != 0;
        if (valid) {
#if !defined(POLAR_PROFILE) && !defined(POLAR_SLOTHIST)
            if (p.pm_out && lig == 0) p.pm_out[cw] = pm_win;
#endif
        }
        {   // the winner's words, in order, into this lane's own column of g_tb (every lane of the group walks
            // the same list, so the loads are broadcasts), then the K info bits by unfrozen rank
            int cur = win;
            for (int w = Wused - 1; w >= 0; --w) {
                g_tb[(size_t)w * CST + POLAR_CL] = g_hist[(size_t)w * CST + POLAR_CGB + cur];
                cur = (int)(g_horg[(size_t)w * CST + POLAR_CGB + cur] & (GS - 1));   // (stale slots of idle groups stay in range)
            }
            wave_mem_fence();
        }
        if (valid) {
            for (int b = LAT ? lane : lig; b < K; b += LAT ? 64 : GS) {
                unsigned r = p.info_rank[b];
                uint32_t wd = g_tb[(size_t)(r >> 5) * CST + POLAR_CL];
                p.out[(size_t)cw * K + b] = win_active ? (uint8_t)((wd >> (r & 31)) & 1u) : (uint8_t)0;
            }
        }
        if constexpr (ED) {
            // codewords with an undecidable |x| < 40 test go to the LLR-domain kernel (host: fallback pass)
            guard |= __ballot(gacc <= ED_GACC_FLAG);
            if constexpr (LAT) { if (valid && lane == 0) p.flags[cw] = (guard != 0) ? 1 : 0; }      // (no conversion pass has cleared it)
            else if (valid && lig == 0 && ((guard >> gbase) & gmask) != 0) p.flags[cw] = 1;
        }
        wave_mem_fence();
#ifdef POLAR_MARGIN
        if (valid && lig == 0 && K >= 16) {      // (overwrites the first 16 info bytes: this build measures, it does not decode)
            double *o = reinterpret_cast<double *>(p.out + (size_t)cw * K);
            o[0] = mingap; o[1] = fingap;
        }
        wave_mem_fence();
#endif
        // next group
        if (p.work) {
            unsigned nxt = 0;
            if (lane == 0) nxt = atomicAdd(p.work, 1u);
            g0 = ((long)nwaves + (long)__builtin_amdgcn_readfirstlane((int)nxt)) * G;
        } else {
            g0 += (long)nwaves * G;
        }
    }  // codeword groups
}

// ------------------------------------------------------------------------------------------
// prefix_kernel — the all-frozen prefix [0, Pe) of every codeword, 32 lanes per codeword.
// Until the first unfrozen position one path exists and every decision is the frozen 0, so the leaf
// LLRs of the prefix are a fixed f/g dataflow of the channel LLRs (g with u = 0). The 32 lanes share
// the ELEMENTS: (1) f-chain of the layers above the prefix block (node 0 of the layers of size
// N/2 .. Q), written contiguously per codeword (pre[cw][1 + N - 2S + j]) for the decode kernel to
// read; (2) the Q block values expanded in registers by a log2(Q)-stage butterfly into the Q leaf
// LLRs; (3) the path metric accumulated over the first Pe leaves in order, same operations and order
// as continuePaths_FrozenBit (PolarCode.cpp:475-487) -> pre[cw][0].
template <bool ED>
// Round 4: (a) with ech_out (exp-domain, staged) the first pass reads the caller's RAW channel pairs, converts them (the input
// guard of ed_front_kernel included), writes the stored form for the decode kernel and feeds the f-node from registers: the separate
// conversion pass (a read and a write of the whole batch) is gone; (b) the layers below the first are computed from the staged copy
// in LDS, in place (element j and j + S belong to the same lane), instead of from what was just written to global memory: the
// write -> read round trip through the L2 between the layers of a codeword was the kernel's time.
__global__ __launch_bounds__(256) void prefix_kernel(PolarDecodeParams p, int staged, double *ech_out) {
    __shared__ double tabs[324];
    // staged: the first pass (channel -> layer N/2) reads the channel pairs in their own order — element j of the layer comes
    // from the pair at bitrev(j), a 16-B read from a different line for every lane when read in element order — and
    // turns the results into element order through LDS ([8 codewords][N/2] doubles, dynamic)
    extern __shared__ double pstage[];

    for (int i = threadIdx.x; i < 322; i += 256) tabs[i] = p.tabs[i];
    __syncthreads();
    const Tabs tb = {tabs, tabs + 64, tabs + 64 + 129};
    u64 guard = 0;
    auto FN = [&](double a, double b) -> double {
        if constexpr (ED) return f_node_e(a, b, guard); else return f_node(a, b, tb);
    };
    auto GN0 = [&](double a, double b) -> double {
        if constexpr (ED) return g_node_e(a, b, 0u, tb); else return g_node(a, b, 0u);
    };
    const int lane = threadIdx.x & 63, lig = lane & 31, gbase = lane & 32;
    const int n = p.n, N = p.N, Q = p.prefix_q, Pe = p.prefix_len;
    const int R = Q >> 5;
    const long per_block = 8;
    const long Bv = p.n_dev ? ((long)*p.n_dev < p.B ? (long)*p.n_dev : p.B) : p.B;
    for (long c0 = (long)blockIdx.x * per_block; c0 < Bv; c0 += (long)gridDim.x * per_block) {
        const long cw = c0 + (threadIdx.x >> 5);
        const bool valid = cw < Bv;
        const double *in0 = ch_row<ED>(p, (size_t)(valid ? cw : 0), N);
        double *pre = const_cast<double *>(p.pre) + (size_t)(valid ? cw : 0) * (size_t)(N - Q + 1);
        double x[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        bool bad_in = false;                 // (fused conversion: this codeword's input guard)
        for (int S = N / 2; S >= Q; S >>= 1) {
            if (valid && staged && 2 * S == N) {
                double *stg = pstage + (size_t)(threadIdx.x >> 5) * (size_t)S;
                double *outp = pre + 1;
                if (ED && ech_out) {
                    double *erow = ech_out + (size_t)cw * (size_t)N;
                    bool any = false, sized = false;
                    for (int m = lig; m < S; m += 32) {
                        const double x0 = CH(in0, 2 * m), x1 = CH(in0, 2 * m + 1);
                        bool f0, f1;
                        const double a = ed_from_channel(x0, tb, f0), b = ed_from_channel(x1, tb, f1);
                        any |= f0 | f1;
                        sized |= (fabs(x0) >= 0.1) | (fabs(x1) >= 0.1);
                        erow[2 * m] = a; erow[2 * m + 1] = b;
                        stg[__brev((unsigned)m) >> (33 - n)] = FN(a, b);       // element j = bitrev_{n-1}(m)
                    }
                    // input guard per codeword (ed_front_kernel): a non-finite or < 1e-9 value, or no value >= 0.1 at all
                    const u64 m_any = __builtin_amdgcn_ballot_w64(any), m_sz = __builtin_amdgcn_ballot_w64(sized);
                    const u64 half = gbase ? 0xFFFFFFFF00000000ull : 0x00000000FFFFFFFFull;
                    bad_in = ((m_any & half) != 0) || ((m_sz & half) == 0);
                } else {
                    for (int m = lig; m < S; m += 32) {
                        const double r = FN(CH(in0, 2 * m), CH(in0, 2 * m + 1));
                        stg[__brev((unsigned)m) >> (33 - n)] = r;              // element j = bitrev_{n-1}(m)
                    }
                }
                wave_mem_fence();                                            // (the 32 lanes of a codeword are in one wave)
                for (int j = lig; j < S; j += 32) {
                    const double r = stg[j];
                    outp[j] = r;
                    if (S == Q) {
#pragma unroll
                        for (int rr = 0; rr < 8; ++rr) if (rr == (j >> 5)) x[rr] = r;
                    }
                }
                wave_mem_fence();
            } else if (valid && staged) {
                // the layer above is still in the staging row (elements [0, 2S)): in place, lane-local
                double *stg = pstage + (size_t)(threadIdx.x >> 5) * (size_t)(N / 2);
                double *outp = pre + 1 + (size_t)(N - 2 * S);
                for (int j = lig; j < S; j += 32) {
                    const double r = FN(stg[j], stg[j + S]);
                    stg[j] = r;
                    outp[j] = r;
                    if (S == Q) {
#pragma unroll
                        for (int rr = 0; rr < 8; ++rr) if (rr == (j >> 5)) x[rr] = r;
                    }
                }
                wave_mem_fence();
            } else if (valid) {
                const bool from_ch = (2 * S == N);
                const double *inp = pre + 1 + (size_t)(N - 4 * S);      // layer of size 2S (unused when from_ch)
                double *outp = pre + 1 + (size_t)(N - 2 * S);
                // four elements per lane and pass: their eight loads are in flight together (S is a multiple of 128:
                // S >= Q >= 64 ... the tail loop takes what is left)
                int j = lig;
                for (; j + 96 < S; j += 128) {
                    double a[4], b[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        if (from_ch) {
                            const unsigned idx = __brev((unsigned)(j + 32 * k)) >> (32 - n);
                            a[k] = CH(in0, idx); b[k] = CH(in0, idx + 1);
                        } else {
                            a[k] = inp[j + 32 * k]; b[k] = inp[j + 32 * k + S];
                        }
                    }
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const double r = FN(a[k], b[k]);
                        outp[j + 32 * k] = r;
                        if (S == Q) {
#pragma unroll
                            for (int rr = 0; rr < 8; ++rr) if (rr == ((j + 32 * k) >> 5)) x[rr] = r;
                        }
                    }
                }
                for (; j < S; j += 32) {
                    double a, b;
                    if (from_ch) {
                        unsigned idx = __brev((unsigned)j) >> (32 - n);
                        a = CH(in0, idx); b = CH(in0, idx + 1);
                    } else {
                        a = inp[j]; b = inp[j + S];
                    }
                    const double r = FN(a, b);
                    outp[j] = r;
                    if (S == Q) {
#pragma unroll
                        for (int rr = 0; rr < 8; ++rr) if (rr == (j >> 5)) x[rr] = r;
                    }
                }
            }
            wave_mem_fence();
        }
        // butterfly: a stage with half-size h turns every node of size 2h into its f-child (lower half)
        // and its g-child (upper half, u = 0); value index i = r*32 + lig
        for (int h = Q / 2; h >= 1; h >>= 1) {
            if (h >= 32) {
                const int hr = h >> 5;
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    if (r < R && (r & hr) == 0) {
#pragma unroll
                        for (int r2 = 0; r2 < 8; ++r2) {
                            if (r2 == r + hr) {
                                const double lo = x[r], hi = x[r2];
                                x[r] = FN(lo, hi);
                                x[r2] = GN0(lo, hi);
                            }
                        }
                    }
                }
            } else {
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    if (r < R) {
                        const double mine = x[r];
                        const double other = shfl_d(mine, lane ^ h);
                        x[r] = (lig & h) ? GN0(other, mine) : FN(mine, other);
                    }
                }
            }
        }
        // path metric over the leaves 0..Pe-1 in order (leaf phi sits in x[phi>>5] of lane phi&31)
        double acc = 0.0;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            if (r < R && r * 32 < Pe) {
                bool ng; double al, sneg, spos;
                leaf_terms<ED>(x[r], true, __builtin_amdgcn_ballot_w64(true), tb, ng, al, sneg, spos);
                const double spv = ng ? spos : sneg;
                const int cnt = (Pe - r * 32 < 32) ? (Pe - r * 32) : 32;
                // (the sum must run in leaf order; unrolled, the 32 cross-lane reads are in flight together and only the
                // additions are serial)
#pragma unroll
                for (int c = 0; c < 32; ++c) {
                    const double t = shfl_d(spv, gbase + c);
                    acc = (c < cnt) ? acc + t : acc;
                }
            }
        }
        if (valid && lig == 0) pre[0] = acc;
        if constexpr (ED) {
            const bool fl = ((guard >> gbase) & 0xFFFFFFFFull) != 0;
            if (ech_out) { if (valid && lig == 0) p.flags[cw] = (fl || bad_in) ? 1 : 0; }       // (no conversion pass has set it)
            else if (valid && lig == 0 && fl) p.flags[cw] = 1;
            guard = 0;
        }
        wave_mem_fence();
    }
}

// This file is compiled three times (polar_amd/build.py): POLAR_ED_TU = 0 instantiates the LLR-domain kernels and
// the small helper kernels, POLAR_ED_TU = 1 the exp-domain kernels of the groups of 4, 8, 16 and 64 lanes, POLAR_ED_TU = 2 the
// exp-domain list of 32 — translation units that build in parallel, and the last one with its own scheduler options
// (max-memory-clause strategy + the AMDGPU register-pressure trackers: +2.2 ... 3.8 % on the headline kernel, -11 % on the
// groups of 8: build.py).
#ifndef POLAR_ED_TU
#define POLAR_ED_TU 0
#endif
#if POLAR_ED_TU != 2
#if POLAR_ED_TU
hipError_t polar_launch_prefix_ed1(const PolarDecodeParams &p, double *ech_out, hipStream_t st) {
#else
hipError_t polar_launch_prefix_ed0(const PolarDecodeParams &p, double *ech_out, hipStream_t st) {
#endif
    long blocks = (p.B + 7) / 8;
    if (blocks > 8192) blocks = 8192;
    const size_t stage = (size_t)8 * (size_t)(p.N / 2) * sizeof(double);        // 64 KiB at N = 2048: two blocks per CU
    const int staged = polar_prefix_is_staged(p.N);
    if (ech_out && !staged) return hipErrorInvalidValue;                        // (the fused conversion exists for the staged first pass)
    hipLaunchKernelGGL(prefix_kernel<POLAR_ED_TU != 0>, dim3((unsigned)blocks), dim3(256), staged ? stage : 0, st, p, staged, ech_out);
    return hipGetLastError();
}
#endif  // POLAR_ED_TU != 2

#if !POLAR_ED_TU
int polar_prefix_is_staged(int N) { return (N >= 64 && (size_t)8 * (size_t)(N / 2) * sizeof(double) <= 64 * 1024) ? 1 : 0; }
// ech_out (exp-domain only, staged block lengths only): p.llr are the caller's RAW rows; the kernel converts them, writes the stored
// form there and sets p.flags (0 / 1) for every codeword — no ed_front_kernel before it
hipError_t polar_launch_prefix(const PolarDecodeParams &p, bool ed, double *ech_out, hipStream_t st) {
    return ed ? polar_launch_prefix_ed1(p, ech_out, st) : polar_launch_prefix_ed0(p, nullptr, st);
}
// ------------------------------------------------------------------------------------------
// ed_front_kernel — channel LLRs -> stored form of the exp-domain kernel (p.llr -> p.ech), plus the
// input guard: flags[cw] = 1 when the codeword holds a non-finite LLR or one below 1e-9 (the
// reference's f-node results are then its own rounding noise), else 0.
template <typename TIN>
__global__ __launch_bounds__(256) void ed_front_kernel(const TIN *llr, double *ech, uint8_t *flags, const double *tabs_g, int N, long B, const unsigned *n_dev) {
    if (n_dev && (long)*n_dev < B) B = (long)*n_dev;
    __shared__ double tabs[324];
    for (int i = threadIdx.x; i < 322; i += 256) tabs[i] = tabs_g[i];
    __syncthreads();
    const Tabs tb = {tabs, tabs + 64, tabs + 64 + 129};
    const int lane = threadIdx.x & 63;
    for (long cw = (long)blockIdx.x * 4 + (threadIdx.x >> 6); cw < B; cw += (long)gridDim.x * 4) {
        const TIN *src = llr + (size_t)cw * N;
        double *dst = ech + (size_t)cw * N;
        bool any = false, sized = false;
        for (int i = lane; i < N; i += 64) {
            bool f;
            const double x = (double)src[i];
            dst[i] = ed_from_channel(x, tb, f);
            any |= f;
            sized |= fabs(x) >= 0.1;
        }
        // A row whose WHOLE channel is tiny (no |llr| >= 0.1: e.g. LLRs scaled by 1e-3, whose largest values are 0.02 ... 0.03;
        // an ordinary row has hundreds of values above 1) is decided by the reference at the rounding noise of its own f-chains from the second layer on —
        // the same class as the per-value guard above, which looks at one value at a time and lets such rows pass. They go to
        // the LLR-domain kernel, which follows the reference's arithmetic further down (fuzz: 4 of 3 600 such rows differed).
        const bool bad = wave_any(any) || !wave_any(sized);
        if (lane == 0) flags[cw] = bad ? 1 : 0;
    }
}
hipError_t polar_launch_ed_front(const void *llr, int llr_f32, double *ech, uint8_t *flags, const double *tabs, int N, long B, const unsigned *n_dev, hipStream_t st) {
    long blocks = (B + 3) / 4;
    if (blocks > 16384) blocks = 16384;
    if (llr_f32) hipLaunchKernelGGL(ed_front_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, st, (const float *)llr, ech, flags, tabs, N, B, n_dev);
    else hipLaunchKernelGGL(ed_front_kernel<double>, dim3((unsigned)blocks), dim3(256), 0, st, (const double *)llr, ech, flags, tabs, N, B, n_dev);
    return hipGetLastError();
}
// flagged codewords -> work list of the fallback pass (order irrelevant: every codeword is independent)
__global__ __launch_bounds__(256) void ed_collect_kernel(const uint8_t *flags, long B, const unsigned *n_dev, uint32_t *list, unsigned *count) {
    if (n_dev && (long)*n_dev < B) B = (long)*n_dev;
    for (long cw = (long)blockIdx.x * 256 + threadIdx.x; cw < B; cw += (long)gridDim.x * 256)
        if (flags[cw]) list[atomicAdd(count, 1u)] = (uint32_t)cw;
}
hipError_t polar_launch_ed_collect(const uint8_t *flags, long B, const unsigned *n_dev, uint32_t *list, unsigned *count, hipStream_t st) {
    long blocks = (B + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(ed_collect_kernel, dim3((unsigned)blocks), dim3(256), 0, st, flags, B, n_dev, list, count);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
int polar_decode_waves_per_block(int pipe) { return pipe ? 1 : 4; }
size_t polar_decode_lds_bytes(int lds_log, int pipe) {
    return 324 * 8 + (size_t)polar_decode_waves_per_block(pipe) * ((size_t)((2u << lds_log) - 1) * 64 * 8 + 128 * 8 + 128);
}

#endif  // !POLAR_ED_TU

template <int GS, bool ED>
static hipError_t launch_gs(const PolarDecodeParams &p, int lds_log, int pipe, int grid, hipStream_t st) {
    // `grid` counts WAVES; blocks = grid / waves-per-block (the host rounds grid to a multiple)
    size_t lds = polar_decode_lds_bytes(lds_log, pipe);
    const int wpb = polar_decode_waves_per_block(pipe);
#define POLAR_LAUNCH(LL, PP) hipLaunchKernelGGL((scl_decode_llr_kernel<GS, LL, PP, ED>), dim3(grid / wpb), dim3(64 * wpb), lds, st, p)
#ifndef POLAR_NO_FIXED_N
    if constexpr (GS == 32 && ED) {
        // the headline shapes (N = 2048 / 1024, list of 17..32, default tuning) have instantiations of their own
        if (lds_log == 3 && !pipe && (p.n == 11 || p.n == 10)) {
            if (p.n == 11) hipLaunchKernelGGL((scl_decode_llr_kernel<GS, 3, 0, ED, 11>), dim3(grid / wpb), dim3(64 * wpb), lds, st, p);
            else hipLaunchKernelGGL((scl_decode_llr_kernel<GS, 3, 0, ED, 10>), dim3(grid / wpb), dim3(64 * wpb), lds, st, p);
            return hipGetLastError();
        }
    }
#endif
    switch (lds_log * 2 + (pipe ? 1 : 0)) {
        case 6: POLAR_LAUNCH(3, 0); break;
#ifndef POLAR_DEV_ONE    // (development builds: the default tuning only)
        case 4: POLAR_LAUNCH(2, 0); break;
        case 7: POLAR_LAUNCH(3, 1); break;
        case 8: POLAR_LAUNCH(4, 0); break;
        case 9: POLAR_LAUNCH(4, 1); break;
        case 10: POLAR_LAUNCH(5, 0); break;
        case 11: POLAR_LAUNCH(5, 1); break;
#endif
        default: return hipErrorInvalidValue;
    }
#undef POLAR_LAUNCH
    return hipGetLastError();
}

// LAT instantiations (one codeword per wave, state in LDS): exp-domain arithmetic for the groups of 4 and 8 lanes (this
// file compiled with POLAR_ED_TU = 1), LLR-domain for the groups of 2 (POLAR_ED_TU = 0)
template <int GS, bool ED>
static hipError_t launch_lat(const PolarDecodeParams &p, int blocks, hipStream_t st) {
    const size_t lds = polar_decode_lat_lds_bytes(p.N, GS, p.W);
    // (per launch, not once per process: the attribute belongs to the function ON THE CURRENT DEVICE, and the multi-device Monte-Carlo
    // driver launches from one thread per device)
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&scl_decode_llr_kernel<GS, 3, 1, ED, 0, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;        // (the host checks `lds` against the device's limit before it chooses this kernel)
    hipLaunchKernelGGL((scl_decode_llr_kernel<GS, 3, 1, ED, 0, 1>), dim3(blocks), dim3(64), lds, st, p);
    return hipGetLastError();
}
#if POLAR_ED_TU == 1
hipError_t polar_launch_decode_lat_ed1(const PolarDecodeParams &p, int gs, int blocks, hipStream_t st) {
    switch (gs) {
        case 2: return launch_lat<2, true>(p, blocks, st);
        case 4: return launch_lat<4, true>(p, blocks, st);
        case 8: return launch_lat<8, true>(p, blocks, st);
        default: return hipErrorInvalidValue;
    }
}
#elif POLAR_ED_TU == 0
hipError_t polar_launch_decode_lat_ed1(const PolarDecodeParams &p, int gs, int blocks, hipStream_t st);
hipError_t polar_launch_decode_lat(const PolarDecodeParams &p, int gs, bool ed, int blocks, hipStream_t st) {
    if (ed) return polar_launch_decode_lat_ed1(p, gs, blocks, st);
    if (gs == 2) return launch_lat<2, false>(p, blocks, st);
    return hipErrorInvalidValue;
}
size_t polar_decode_lat_lds_bytes(int N, int gs, int W) {
    const int cwords = (N >= 128) ? (N / 32 - 2) : 0;
    // tables + exchange buffers + converted channel + layers + partial-sum and history words (one column per path)
    return 324 * 8 + (128 * 8 + 128) + (size_t)N * 8 + (size_t)N * gs * 8 + ((size_t)2 * cwords + (size_t)3 * W) * gs * 4 + 64;
}
#endif

#if POLAR_ED_TU == 2
hipError_t polar_launch_decode_llr_ed1_gs32(const PolarDecodeParams &p, int lds_log, int pipe, int grid, hipStream_t st) {
    return launch_gs<32, true>(p, lds_log, pipe, grid, st);
}
#elif POLAR_ED_TU == 1
hipError_t polar_launch_decode_llr_ed1_gs32(const PolarDecodeParams &p, int lds_log, int pipe, int grid, hipStream_t st);
hipError_t polar_launch_decode_llr_ed1(const PolarDecodeParams &p, int gs, int lds_log, int pipe, int grid, hipStream_t st) {
    switch (gs) {
#ifndef POLAR_DEV_GS32
        case 4: return launch_gs<4, true>(p, lds_log, pipe, grid, st);
        case 8: return launch_gs<8, true>(p, lds_log, pipe, grid, st);
        case 16: return launch_gs<16, true>(p, lds_log, pipe, grid, st);
        case 64: return launch_gs<64, true>(p, lds_log, pipe, grid, st);
#endif
        case 32: return polar_launch_decode_llr_ed1_gs32(p, lds_log, pipe, grid, st);
        default: return hipErrorInvalidValue;
    }
}
#else
hipError_t polar_launch_decode_llr_ed0(const PolarDecodeParams &p, int gs, int lds_log, int pipe, int grid, hipStream_t st) {
    switch (gs) {
        case 1: return launch_gs<1, false>(p, lds_log, pipe, grid, st);
#ifndef POLAR_DEV_GS32
        case 2: return launch_gs<2, false>(p, lds_log, pipe, grid, st);
        case 4: return launch_gs<4, false>(p, lds_log, pipe, grid, st);
        case 8: return launch_gs<8, false>(p, lds_log, pipe, grid, st);
        case 16: return launch_gs<16, false>(p, lds_log, pipe, grid, st);
        case 64: return launch_gs<64, false>(p, lds_log, pipe, grid, st);
#endif
        case 32: return launch_gs<32, false>(p, lds_log, pipe, grid, st);
        default: return hipErrorInvalidValue;
    }
}
hipError_t polar_launch_decode_llr(const PolarDecodeParams &p, int gs, int lds_log, int pipe, int grid, bool ed, hipStream_t st) {
    return ed ? polar_launch_decode_llr_ed1(p, gs, lds_log, pipe, grid, st) : polar_launch_decode_llr_ed0(p, gs, lds_log, pipe, grid, st);
}
#endif
