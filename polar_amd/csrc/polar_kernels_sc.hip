// polar_kernels_sc.hip — successive-cancellation decoder for list size 1 (gfx950).
//
// decode_scl_llr(llr, 1) of the reference (PolarCode.cpp:130-190 with _list_size = 1) keeps ONE path: the path metric
// (:475-487, 505-506) never decides anything, findMostProbablePath (:609-644) returns that path whatever the CRC says,
// and every unfrozen decision is the sign of the leaf LLR (:586-604: the fork that survives is the one with the larger
// probability, i.e. u = (llr < 0)). What is left is the f/g recursion (:422-455) and the partial sums (:457-473), and
// for those two classic identities hold bit for bit in the reference's own arithmetic:
//   * a subtree whose leaves are all frozen decides all zeros: nothing below its root is evaluated;
//   * a subtree whose leaves are all unfrozen decides x = hard(alpha) at its root (sign(f(a,b)) = sign(a) sign(b) and
//     sign(g(a,b,u)) = sign(b) for u = hard(f(a,b)), exact f or min-sum alike), so its S leaf decisions are the
//     polar transform of the S sign bits: no f/g below the root either.
// (Both need a non-zero LLR wherever a sign is taken; codewords with |llr| < 1e-9 or non-finite inputs are flagged by
// the front kernel and decoded by the general kernel — same guard as the exp-domain list kernel. One more case goes
// there: the reference's path metric reaching +inf, see the all-frozen bound below.)
//
// Layout: EIGHT LANES PER CODEWORD, 8 codewords per wave: element j of a layer lives in sublane j & 7, row j >> 3
// (rows of 64 doubles = 512 B, as everywhere). Every codeword follows the same host-built schedule (it depends on the
// frozen set only), so the wave never diverges. Layers of size <= 32 in LDS, the larger ones in a per-wave global
// scratch; the channel values arrive permuted into the kernel's element order (sc8_front_kernel, through LDS: both
// sides coalesced). Node arithmetic: exp-domain stored form (polar_edom.h), i.e. one division per f or g and no
// transcendental anywhere (there is no path metric to feed). Partial sums and decisions are bit words in LDS.
// (A one-lane-per-codeword layout of the same schedule was built first: 10 M cw/s at batch 65 536 — a quarter of the
// machine's wave slots — against 18 M for this one.)
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "polar_kernels.h"
#include "polar_device.h"
// (the block-placement hints of polar_edom.h pay where the code is several times the instruction cache — the list kernels,
// +5 % — and cost 2 % here, where it fits)
#define POLAR_NO_COLD_HINTS
#include "polar_edom.h"

namespace {


__device__ __forceinline__ bool ed_is_neg(double v) { return (__double2hiint(v) < 0) && fabs(v) != 1.0; }

// polar transform of the low S bits of x (S <= 64): leaf decisions u from the root's hard bits
__device__ __forceinline__ u64 bits_transform(u64 x, int S) {
    if (S > 32) x ^= (x >> 32) & 0x00000000FFFFFFFFull;
    if (S > 16) x ^= (x >> 16) & 0x0000FFFF0000FFFFull;
    if (S > 8) x ^= (x >> 8) & 0x00FF00FF00FF00FFull;
    if (S > 4) x ^= (x >> 4) & 0x0F0F0F0F0F0F0F0Full;
    if (S > 2) x ^= (x >> 2) & 0x3333333333333333ull;
    if (S > 1) x ^= (x >> 1) & 0x5555555555555555ull;
    return x;
}

}  // namespace

// ------------------------------------------------------------------------------------------
// sc8_*: nodes of size >= 8 are lane-local (j and j + S share the sublane), the three narrowest layers exchange across
// sublanes; partial sums and decisions are 32-bit words [word][codeword] in LDS. Measured: layers <= 32 in LDS with 20
// waves/CU beat layers <= 256 with 4 waves/CU by 2.5x (the narrow part of the tree is pure latency).
#ifndef SC8_SL_DEF
#define SC8_SL_DEF 32
#endif
constexpr int SC8_SL = SC8_SL_DEF;             // layers of size <= SC8_SL live in LDS
#ifndef SC8_U
#define SC8_U 4
#endif
#ifndef SC8_WPB_DEF
#define SC8_WPB_DEF 4
#endif
constexpr int SC8_WPB = SC8_WPB_DEF;           // waves per block (they share the 2.6 KiB of exp/log tables)
__device__ __forceinline__ int sc8_rowbase(int S) { return S < 8 ? (S == 1 ? 0 : (S == 2 ? 1 : 2)) : (S / 8 + 2); }   // rows: 1,1,1,1,2,4,...
constexpr int SC8_ROWS = 3 + (2 * SC8_SL / 8 - 1);     // sizes 1, 2, 4 | 8 .. SC8_SL
// per-wave global scratch: the layers larger than the LDS-resident ones + the decision words [ceil(N/32)][8] (uint32),
// rounded to whole 512-B rows
__host__ __device__ static inline size_t sc8_scratch_doubles(int N) {
    const size_t layers = (N > 2 * SC8_SL) ? (size_t)(N - 2 * SC8_SL) / 8 * 64 : 0;
    const size_t uwords = (size_t)((N + 31) / 32) * 8;
    return layers + ((uwords * 4 + 511) / 512) * 64;
}

template <typename TIN>
__global__ __launch_bounds__(256) void sc8_front_kernel(const TIN *llr, double *ech_p, unsigned int *flag_words, const double *tabs_g,
                                                        int n, long B, const unsigned *n_dev, int staged) {
    __shared__ double tabs[324];
    for (int i = threadIdx.x; i < 322; i += 256) tabs[i] = tabs_g[i];
    __syncthreads();
    const Tabs tb = {tabs, tabs + 64, tabs + 64 + 129};
    if (n_dev && (long)*n_dev < B) B = (long)*n_dev;
    const int N = 1 << n;
    extern __shared__ double row[];                  // N doubles when the row fits (staged = 1): both sides coalesced
    for (long cw = blockIdx.x; cw < B; cw += gridDim.x) {
        bool any = false;
        // element e of the kernel's order is channel position bitrev_n(e)
        for (int i = threadIdx.x; i < N; i += 256) {
            bool f;
            const double v = ed_from_channel((double)llr[(size_t)cw * N + i], tb, f);
            any |= f;
            if (staged) row[i] = v;
            else ech_p[(size_t)cw * N + (__brev((unsigned)i) >> (32 - n))] = v;
        }
        if (staged) {
            __syncthreads();
            for (int e = threadIdx.x; e < N; e += 256) ech_p[(size_t)cw * N + e] = row[__brev((unsigned)e) >> (32 - n)];
            __syncthreads();
        }
        if (any) atomicOr(&flag_words[cw >> 5], 1u << (cw & 31));
    }
}

__global__ __launch_bounds__(64 * SC8_WPB) void sc8_decode_kernel(PolarScParams p) {
    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int wave_id = blockIdx.x * SC8_WPB + wib;
    const int nwaves = gridDim.x * SC8_WPB;
    const int N = p.N, K = p.K;
    const int sub = lane & 7, cws = lane >> 3;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double *tabs = reinterpret_cast<double *>(smem);
    for (int i = threadIdx.x; i < 322; i += SC8_WPB * 64) tabs[i] = p.tabs[i];
    __syncthreads();
    const Tabs tb = {tabs, tabs + 64, tabs + 64 + 129};
    const int words = (N + 31) / 32;
    const size_t wave_lds = (size_t)SC8_ROWS * 64 * 8 + (size_t)words * 8 * 4;
    unsigned char *wb = smem + 324 * 8 + (size_t)wib * wave_lds;
    double *lds = reinterpret_cast<double *>(wb);                                  // [SC8_ROWS][64]
    uint32_t *bw = reinterpret_cast<uint32_t *>(wb + (size_t)SC8_ROWS * 64 * 8);   // partial sums [words][8]
    const int big_rows = (N > 2 * SC8_SL) ? (N - 2 * SC8_SL) / 8 : 0;              // rows of the layers 2 SC8_SL .. N/2
    double *g_a = p.a_scr + (size_t)wave_id * sc8_scratch_doubles(N);
    // decisions [words][8]: produced in leaf order, read once at the end -> the wave's global scratch, not the LDS (which
    // caps the waves per CU); the word under construction is a register of the codeword's first sublane
    uint32_t *uw = reinterpret_cast<uint32_t *>(g_a + (size_t)big_rows * 64);
    long Bv = p.B;
    if (p.n_dev && (long)*p.n_dev < Bv) Bv = (long)*p.n_dev;
    const long groups = (Bv + 7) / 8;
    typedef const uint32_t __attribute__((address_space(4))) *kconst_u32;
    const kconst_u32 ops = (kconst_u32)(uintptr_t)p.ops;
    u64 guard = 0;

    for (long g = wave_id; g < groups;) {
        const long cw = g * 8 + cws;
        const bool valid = cw < Bv;
        const double *ch = p.ech_t + (size_t)(valid ? cw : 0) * N;               // this codeword's channel values, kernel order
        for (int i = lane; i < words * 8; i += 64) bw[i] = 0u;
        guard = 0;
        uint32_t ucur = 0;                 // decisions of word `ucur_w` so far (sublane 0 of each codeword)
        int ucur_w = -1;                   // wave-uniform
        auto uflush = [&](int w_new) {
            if (ucur_w >= 0 && sub == 0) uw[(size_t)ucur_w * 8 + cws] = ucur;
            ucur = 0;
            ucur_w = w_new;
        };
        wave_mem_fence();
        for (int io = 0; io < p.n_ops; ++io) {
            const uint32_t op = ops[io];
            const int type = (int)(op & 7u), sh = (int)((op >> 3) & 15u), base = (int)((op >> 8) & 0xFFFFu);
            const int extra = (int)((op >> 24) & 3u);                  // F steps of the child chain folded into this visit
            const int S = 1 << sh;
            if (type <= 1) {
                if (S >= 8) {
                    const int R = S / 8;
                    // source rows r and r + R (element j = 8r + sub and j + S), destination row r
                    auto visit = [&](const double *src, size_t sstr, double *dst) {
                        for (int r = 0; r < R; r += SC8_U) {
                            double a[SC8_U], b[SC8_U];
#pragma unroll
                            for (int k = 0; k < SC8_U; ++k) if (r + k < R) { a[k] = src[(size_t)(r + k) * sstr]; b[k] = src[(size_t)(r + k + R) * sstr]; }
#pragma unroll
                            for (int k = 0; k < SC8_U; ++k) if (r + k < R) {
                                double y;
                                if (type == 0) y = f_node_e(a[k], b[k], guard);
                                else {
                                    const int j = base + 8 * (r + k) + sub;
                                    const uint32_t w = bw[(size_t)(j >> 5) * 8 + cws];
                                    y = g_node_e(a[k], b[k], w << (31 - (j & 31)), tb);
                                }
                                dst[(size_t)(r + k) * 64] = y;
                            }
                        }
                    };
                    // A visit whose output is the input of the F visit that follows (the left descent F F F..., or G then F
                    // of the right child) evaluates that F - and the next - on its results while they are in registers: the
                    // layers are still written (their G visits read them later) but not read back (every visit of an
                    // HBM-resident layer is bandwidth: 62 GB per 262 144 codewords at 5.4 TB/s before this).
                    // One iteration = one row of the LAST layer of the chain = 2^(D-1) row pairs of the source.
                    // top = true (depth 3 only): the source is the CHANNEL, read where the caller put it. Element e of the kernel's
                    // order is channel position bitrev_n(e); the eight source values of one iteration — rows r + k RS (+ R) — are
                    // eight CONSECUTIVE channel positions (the three top bits of the row index are the three low bits of the
                    // position), so a lane reads one whole 64-byte sector (32 B of floats) and no permuted copy of the batch is
                    // ever written: sc8_front_kernel cost 12 % of the time and 32 KB of traffic per codeword. F converts the
                    // eight values to stored form; G adds in the LLR domain as the reference does (PolarCode.cpp:449-450) and
                    // converts its four results.
                    auto visit_chain = [&](const double *src, size_t sstr, auto depth, auto last_on_chip, auto top_) {
                        constexpr int D = decltype(depth)::value, NP = 1 << (D - 1);
                        constexpr bool LL = decltype(last_on_chip)::value;   // the last layer of the chain is the LDS-resident SC8_SL
                        constexpr bool TOP = decltype(top_)::value;
                        const int RS = R / NP;                               // rows of the last layer
                        double *d0 = g_a + (size_t)((S - 2 * SC8_SL) / 8) * 64 + lane;
                        double *d1g = g_a + (size_t)((LL && D == 2 ? 0 : S / 2 - 2 * SC8_SL) / 8) * 64 + lane;
                        double *d2g = g_a + (size_t)((LL || D < 3 ? 0 : S / 4 - 2 * SC8_SL) / 8) * 64 + lane;
                        double *dll = lds + (size_t)sc8_rowbase(SC8_SL) * 64 + lane;
                        for (int it = 0; it < RS; ++it) {
                            // (top: rows r and r + RS/2 are the two 64-byte halves of one 128-byte line — visited back to back, the
                            // second half is an L2 hit instead of a second fetch of the line)
                            const int r = (TOP && D == 3) ? (it >> 1) + (it & 1) * (RS / 2) : it;
                            double a[NP], b[NP], y[NP];
                            if constexpr (TOP && D == 3) {
                                // position = bitrev3(sub) N/8 + bitrev(r) 8 + 4 k0 + 2 k1 + (b ? 1 : 0)
                                const size_t pos = (size_t)(valid ? cw : 0) * N + (size_t)(__brev((unsigned)sub) >> 29) * (size_t)(N / 8)
                                                   + (size_t)(__brev((unsigned)r) >> (32 - (p.n - 6))) * 8;
                                double v[8];
                                if (p.llr_f32) {
                                    const float4 *q = reinterpret_cast<const float4 *>(reinterpret_cast<const float *>(p.llr) + pos);
                                    const float4 x0 = q[0], x1 = q[1];
                                    v[0] = x0.x; v[1] = x0.y; v[2] = x0.z; v[3] = x0.w; v[4] = x1.x; v[5] = x1.y; v[6] = x1.z; v[7] = x1.w;
                                } else {
                                    const double2 *q = reinterpret_cast<const double2 *>(reinterpret_cast<const double *>(p.llr) + pos);
#pragma unroll
                                    for (int i = 0; i < 4; ++i) { const double2 x = q[i]; v[2 * i] = x.x; v[2 * i + 1] = x.y; }
                                }
                                bool fl = false;          // the input guard of sc8_front_kernel (ed_from_channel)
#pragma unroll
                                for (int i = 0; i < 8; ++i) { const double fx = fabs(v[i]); fl |= !(fx < __builtin_inf()) || fx < 1e-9; }
                                guard |= __builtin_amdgcn_ballot_w64(fl);
#pragma unroll
                                for (int k = 0; k < NP; ++k) { a[k] = v[4 * (k & 1) + 2 * (k >> 1)]; b[k] = v[4 * (k & 1) + 2 * (k >> 1) + 1]; }
                                if (type == 0) {
#pragma unroll
                                    for (int k = 0; k < NP; ++k) { a[k] = ed_from_llr(a[k], tb); b[k] = ed_from_llr(b[k], tb); }
                                }
                            } else {
#pragma unroll
                                for (int k = 0; k < NP; ++k) { a[k] = src[(size_t)(r + k * RS) * sstr]; b[k] = src[(size_t)(r + k * RS + R) * sstr]; }
                            }
#pragma unroll
                            for (int k = 0; k < NP; ++k) {
                                if (type == 0) y[k] = f_node_e(a[k], b[k], guard);
                                else {
                                    const int j = base + 8 * (r + k * RS) + sub;
                                    const uint32_t w = bw[(size_t)(j >> 5) * 8 + cws];
                                    if constexpr (TOP && D == 3) y[k] = ed_from_llr(b[k] + (((w >> (j & 31)) & 1u) ? -a[k] : a[k]), tb);
                                    else y[k] = g_node_e(a[k], b[k], w << (31 - (j & 31)), tb);
                                }
                                d0[(size_t)(r + k * RS) * 64] = y[k];
                            }
#pragma unroll
                            for (int k = 0; k < NP / 2; ++k) {
                                y[k] = f_node_e(y[k], y[k + NP / 2], guard);
                                if constexpr (LL && D == 2) dll[(size_t)(r + k * RS) * 64] = y[k];
                                else d1g[(size_t)(r + k * RS) * 64] = y[k];
                            }
                            if constexpr (D > 2) {
                                y[0] = f_node_e(y[0], y[1], guard);
                                if constexpr (LL) dll[(size_t)r * 64] = y[0];
                                else d2g[(size_t)r * 64] = y[0];
                            }
                        }
                    };
                    if (extra) {
                        typedef std::integral_constant<int, 2> D2;
                        typedef std::integral_constant<int, 3> D3;
                        const bool ll = (S >> extra) <= SC8_SL;
                        typedef std::false_type NT;
                        if (2 * S == N && p.llr) {               // (host: only schedules whose top visits are depth-3 chains into HBM layers)
                            visit_chain(nullptr, 0, D3(), std::false_type(), std::true_type());
                        } else if (2 * S == N) {
                            const double *sc = ch + sub;
                            if (extra == 1) { if (ll) visit_chain(sc, 8, D2(), std::true_type(), NT()); else visit_chain(sc, 8, D2(), std::false_type(), NT()); }
                            else { if (ll) visit_chain(sc, 8, D3(), std::true_type(), NT()); else visit_chain(sc, 8, D3(), std::false_type(), NT()); }
                        } else {
                            const double *sg = g_a + (size_t)((2 * S - 2 * SC8_SL) / 8) * 64 + lane;
                            if (extra == 1) { if (ll) visit_chain(sg, 64, D2(), std::true_type(), NT()); else visit_chain(sg, 64, D2(), std::false_type(), NT()); }
                            else { if (ll) visit_chain(sg, 64, D3(), std::true_type(), NT()); else visit_chain(sg, 64, D3(), std::false_type(), NT()); }
                        }
                        wave_mem_fence();
                        continue;
                    }
                    double *dl = lds + (size_t)sc8_rowbase(S) * 64 + lane;
                    double *dg = g_a + (size_t)((S - 2 * SC8_SL) / 8) * 64 + lane;
                    if (2 * S == N) {                      // source = channel: 8 consecutive doubles per codeword and row
                        if (S <= SC8_SL) visit(ch + sub, 8, dl); else visit(ch + sub, 8, dg);
                    } else if (2 * S <= SC8_SL) visit(lds + (size_t)sc8_rowbase(2 * S) * 64 + lane, 64, dl);
                    else if (S <= SC8_SL) visit(g_a + (size_t)((2 * S - 2 * SC8_SL) / 8) * 64 + lane, 64, dl);
                    else visit(g_a + (size_t)((2 * S - 2 * SC8_SL) / 8) * 64 + lane, 64, dg);
                } else {
                    // S = 4, 2, 1: the source (size 2S <= 8) is one row; element j + S sits S sublanes to the right
                    const double v = (2 * S == N) ? ch[sub & (2 * S - 1)] : lds[(size_t)sc8_rowbase(2 * S) * 64 + lane];
                    const double vb0 = __shfl(v, lane + S, 64);
                    const double va = (sub < S) ? v : 0.5, vb = (sub < S) ? vb0 : 0.5;      // (idle sublanes: harmless operands)
                    double y;
                    if (type == 0) y = f_node_e(va, vb, guard);
                    else {
                        const int j = base + sub;
                        const uint32_t w = bw[(size_t)(j >> 5) * 8 + cws];
                        y = g_node_e(va, vb, w << (31 - (j & 31)), tb);
                    }
                    lds[(size_t)sc8_rowbase(S) * 64 + lane] = y;
                }
                wave_mem_fence();
            } else if (type == 3) {
                // ---- all-unfrozen subtree: hard decisions of its root (exact zero -> general kernel), their polar transform
                const double *src = (S == N) ? nullptr : (S <= SC8_SL ? lds + (size_t)sc8_rowbase(S) * 64 + lane
                                                                      : g_a + (size_t)((S - 2 * SC8_SL) / 8) * 64 + lane);
                const int R = S >= 8 ? S / 8 : 1;
                bool zero = false;
                // The reference does not decide by the sign of the leaf LLR but by comparing PM + log(1+e^-llr) with
                // PM + log(1+e^llr) (PolarCode.cpp:505-506): below the rounding granularity of PM (<= 1e-11 for any PM this
                // decoder can reach) the two are EQUAL and the tie goes to bit 0. The smallest leaf magnitude of an
                // all-unfrozen node is its first leaf's, 2 atanh(prod tanh(|x_i| / 2)) over the root values; tanh(|x|/2) =
                // (1-E)/(1+E) >= 1 - 2E. Codewords whose product falls below 1e-8 (a node in the worst channels: only with
                // unfrozen sets no construction produces) go to the general kernel, which keeps the metric.
                double q = 1.0;
                for (int r4 = 0; r4 < R; r4 += 4) {
                    uint32_t acc = 0;
                    for (int k = 0; k < 4 && r4 + k < R; ++k) {
                        const double v = (S == N) ? ch[(size_t)(r4 + k) * 8 + sub] : src[(size_t)(r4 + k) * 64];
                        const bool in = (S >= 8) || sub < S;
                        zero |= in && fabs(v) == 1.0;
                        {
                            const double m = fabs(v);
                            const double t = in ? __builtin_fmax(1.0 - 2.0 * ((m > 1.0) ? 0.0 : m), 0.0) : 1.0;
                            q = __builtin_fmax(q * t, 1e-300);
                        }
                        const u64 bal = __builtin_amdgcn_ballot_w64(in && ed_is_neg(v));
                        acc |= (uint32_t)((bal >> (8 * cws)) & 0xFFull) << (8 * k);
                    }
                    if (sub == 0) {
                        const int j0 = base + 8 * r4;
                        if (S >= 32) bw[(size_t)(j0 >> 5) * 8 + cws] = acc;
                        else bw[(size_t)(j0 >> 5) * 8 + cws] |= acc << (j0 & 31);
                    }
                }
                guard |= __builtin_amdgcn_ballot_w64(zero);
#pragma unroll
                for (int off = 1; off < 8; off <<= 1) q = __builtin_fmax(q * __shfl_xor(q, off, 64), 1e-300);
                if (__builtin_amdgcn_ballot_w64(q < 1e-8)) {
                    // the cheap bound failed somewhere in the wave: the product itself (a division per root value)
                    double T = 1.0;
                    for (int r = 0; r < R; ++r) {
                        const double v = (S == N) ? ch[(size_t)r * 8 + sub] : src[(size_t)r * 64];
                        const bool in = (S >= 8) || sub < S;
                        const double m = __builtin_fmin(fabs(v), 1.0);
                        const double t = (in && fabs(v) <= 1.0) ? ed_div(1.0 - m, 1.0 + m) : 1.0;
                        T = __builtin_fmax(T * t, 1e-300);
                    }
#pragma unroll
                    for (int off = 1; off < 8; off <<= 1) T = __builtin_fmax(T * __shfl_xor(T, off, 64), 1e-300);
                    guard |= __builtin_amdgcn_ballot_w64(T < 1e-8);
                }
                wave_mem_fence();
                if (S <= 32) {
                    if ((base >> 5) != ucur_w) uflush(base >> 5);
                    if (sub == 0) {
                        const int sft = base & 31;
                        const uint32_t m = (S == 32) ? 0xFFFFFFFFu : ((1u << S) - 1u);
                        const uint32_t x = (bw[(size_t)(base >> 5) * 8 + cws] >> sft) & m;
                        ucur |= (uint32_t)bits_transform((u64)x, S) << sft;
                    }
                } else if (sub == 0) {
                    // whole words: straight to the scratch (the word under construction is an earlier one)
                    const int nw = S / 32, w0b = base >> 5;
                    for (int b0 = 0; b0 < nw; ++b0) {
                        uint32_t x = 0;
                        for (int b1 = b0; b1 < nw; ++b1) if ((b1 & b0) == b0) x ^= bw[(size_t)(w0b + b1) * 8 + cws];
                        uw[(size_t)(w0b + b0) * 8 + cws] = (uint32_t)bits_transform((u64)x, 32);
                    }
                }
                wave_mem_fence();
            } else if (type == 4) {
                // ---- combine: left half ^= right half (child size S)
                if (S >= 32) {
                    for (int w = sub; w < S / 32; w += 8)
                        bw[(size_t)((base >> 5) + w) * 8 + cws] ^= bw[(size_t)(((base + S) >> 5) + w) * 8 + cws];
                } else if (sub == 0) {
                    const int sft = base & 31;
                    uint32_t x = bw[(size_t)(base >> 5) * 8 + cws];
                    x ^= ((x >> (sft + S)) & ((1u << S) - 1u)) << sft;
                    bw[(size_t)(base >> 5) * 8 + cws] = x;
                }
                wave_mem_fence();
            } else if (type == 6) {
                // ---- all-frozen subtree: its leaves are not evaluated, but the reference's path metric would become +inf at
                // a frozen leaf with llr < -709.78 (PolarCode.cpp:483) and from then on every unfrozen decision is the
                // tie-break's 0, not the sign. No leaf exceeds the sum of the root's |x| (= -log of the product of the
                // stored values): beyond 690 the codeword goes to the general kernel.
                const double *src = S <= SC8_SL ? lds + (size_t)sc8_rowbase(S) * 64 + lane : g_a + (size_t)((S - 2 * SC8_SL) / 8) * 64 + lane;
                const int R = S >= 8 ? S / 8 : 1;
                double P = 1.0;
                bool bad = false;
                for (int r = 0; r < R; ++r) {
                    const double m = fabs(src[(size_t)r * 64]);
                    const bool in = (S >= 8) || sub < S;
                    bad |= in && m > 1.0;
                    P *= in ? __builtin_fmin(m, 1.0) : 1.0;
                    bad |= P < 1e-300;
                    P = __builtin_fmax(P, 1e-300);
                }
#pragma unroll
                for (int off = 1; off < 8; off <<= 1) {
                    P *= __shfl_xor(P, off, 64);
                    bad |= P < 1e-300;
                    P = __builtin_fmax(P, 1e-300);
                }
                guard |= __builtin_amdgcn_ballot_w64(bad);
            }
        }
        uflush(-1);
        wave_mem_fence();
        if (valid && sub == 0 && ((guard >> (8 * cws)) & 0xFFull)) atomicOr(&p.flag_words[cw >> 5], 1u << (cw & 31));
        // ---- info bits, one codeword at a time across the wave
        for (int c = 0; c < 8; ++c) {
            const long cwc = g * 8 + c;
            if (cwc >= Bv) break;
            for (int b = lane; b < K; b += 64) {
                const unsigned pos = p.order[b];
                p.out[(size_t)cwc * K + b] = (uint8_t)((uw[(size_t)(pos >> 5) * 8 + c] >> (pos & 31)) & 1u);
            }
        }
        wave_mem_fence();
        if (p.work) {
            unsigned nxt = 0;
            if (lane == 0) nxt = atomicAdd(p.work, 1u);
            g = (long)nwaves + (long)__builtin_amdgcn_readfirstlane((int)nxt);
        } else {
            g += nwaves;
        }
    }
}
size_t polar_sc8_lds_bytes(int N) { return 324 * 8 + (size_t)SC8_WPB * ((size_t)SC8_ROWS * 64 * 8 + (size_t)((N + 31) / 32) * 8 * 4); }
int polar_sc8_waves_per_block() { return SC8_WPB; }
// folding needs: n - 6 >= 1 row-index bits below the chain's three, and the chain's last layer (N/8) HBM-resident
int polar_sc8_fold_min_log() { int l = 7; while ((1 << l) / 8 < 2 * SC8_SL) ++l; return l; }
int polar_sc8_min_global_log() { int l = 0; while ((1 << l) < 2 * SC8_SL) ++l; return l; }
int polar_sc8_waves_per_cu(int N) { const int w = (int)((160 * 1024) / polar_sc8_lds_bytes(N)) * SC8_WPB; return w > 32 ? 32 : w; }
size_t polar_sc8_scratch_doubles_per_wave(int N) { return sc8_scratch_doubles(N); }
hipError_t polar_launch_sc8_front(const void *llr, int llr_f32, double *ech_p, unsigned int *flag_words, const double *tabs,
                                  int n, long B, const unsigned *n_dev, hipStream_t st) {
    const unsigned blocks = (unsigned)(B < 65536 ? (B ? B : 1) : 65536);
    const int staged = (n <= 12) ? 1 : 0;                      // the row (<= 32 KiB) is permuted through LDS
    const size_t sh = staged ? ((size_t)8 << n) : 0;
    if (llr_f32) hipLaunchKernelGGL(sc8_front_kernel<float>, dim3(blocks), dim3(256), sh, st, (const float *)llr, ech_p, flag_words, tabs, n, B, n_dev, staged);
    else hipLaunchKernelGGL(sc8_front_kernel<double>, dim3(blocks), dim3(256), sh, st, (const double *)llr, ech_p, flag_words, tabs, n, B, n_dev, staged);
    return hipGetLastError();
}
hipError_t polar_launch_sc8_decode(const PolarScParams &p, int grid_waves, hipStream_t st) {
    hipLaunchKernelGGL(sc8_decode_kernel, dim3((grid_waves + SC8_WPB - 1) / SC8_WPB), dim3(64 * SC8_WPB), polar_sc8_lds_bytes(p.N), st, p);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// sc_lat_kernel — list size 1, ONE CODEWORD PER WAVE: the latency form of the pruned SC decoder, for the reference's own
// call pattern (one codeword per call: PolarCode.cpp:756, PolarM/main_MC_CC_Comparison.m:96) and small batches.
// sc8_decode_kernel packs eight codewords into a wave and keeps the big layers in a global scratch: right for throughput,
// but a lone wave then pays a full memory round trip (~2 us) per dependent access of the top layers, and B = 1 costs
// 0.85 ms against 0.33 ms on one host core. Here the 64 lanes share the ELEMENTS of one codeword (element j in lane
// j & 63), the whole state lives in LDS (layer of size S at a[S .. 2S), the converted channel at a[N .. 2N): 16 N bytes,
// 32 KiB at N = 2048) and nothing but the channel row and the K result bytes touches HBM. Same host-built schedule (the
// folded F steps of an op are unrolled again), same node arithmetic (f_node_e / g_node_e: results bit-identical to
// sc8_decode_kernel's), same guards -> flag word -> the general kernel in the same call. Nodes of size >= 64 are
// lane-local (j and j + S share a lane); narrower ones read their partner straight from LDS.
constexpr int SCLAT_U = 4;
// A whole subtree of S <= 8 leaves decoded in registers (every lane holds the same S root values: one broadcast LDS read
// instead of an LDS round trip per node of the subtree): frozen pattern `fz` (bit j = leaf j frozen), returns the partial
// sums of the S leaf positions, `ubits` their decisions. Same arithmetic, pruning identities and guards as the ops of the
// schedule (types 0, 1, 3, 4, 6 of sc8_decode_kernel), evaluated depth first.
template <int S>
__device__ __forceinline__ uint32_t sc_block(const double (&v)[S], uint32_t fz, const Tabs &tb, u64 &guard, uint32_t &ubits) {
    constexpr uint32_t ALL = (1u << S) - 1u;
    if (fz == ALL) {                       // all frozen: zeros; the +inf path-metric bound (type 6)
        double P = 1.0;
        bool bad = false;
#pragma unroll
        for (int j = 0; j < S; ++j) {
            const double m = fabs(v[j]);
            bad |= m > 1.0;
            P *= __builtin_fmin(m, 1.0);
            bad |= P < 1e-300;
            P = __builtin_fmax(P, 1e-300);
        }
        guard |= __builtin_amdgcn_ballot_w64(bad);
        ubits = 0u;
        return 0u;
    }
    if (fz == 0u) {                        // all unfrozen: hard decisions of the root, their polar transform (type 3)
        uint32_t hbits = 0u;
        bool zero = false;
        double q = 1.0;
#pragma unroll
        for (int j = 0; j < S; ++j) {
            const double m = fabs(v[j]);
            zero |= m == 1.0;
            q = __builtin_fmax(q * __builtin_fmax(1.0 - 2.0 * ((m > 1.0) ? 0.0 : m), 0.0), 1e-300);
            hbits |= (ed_is_neg(v[j]) ? 1u : 0u) << j;
        }
        guard |= __builtin_amdgcn_ballot_w64(zero);
        if (__builtin_amdgcn_ballot_w64(q < 1e-8)) {
            double T = 1.0;
#pragma unroll
            for (int j = 0; j < S; ++j) {
                const double m = __builtin_fmin(fabs(v[j]), 1.0);
                T = __builtin_fmax(T * ((fabs(v[j]) <= 1.0) ? ed_div(1.0 - m, 1.0 + m) : 1.0), 1e-300);
            }
            guard |= __builtin_amdgcn_ballot_w64(T < 1e-8);
        }
        ubits = (uint32_t)bits_transform((u64)hbits, S);
        return hbits;
    }
    if constexpr (S > 1) {
        constexpr int H = S / 2;
        double l[H], r[H];
#pragma unroll
        for (int j = 0; j < H; ++j) l[j] = f_node_e(v[j], v[j + H], guard);
        uint32_t ul, ur;
        const uint32_t xl = sc_block<H>(l, fz & ((1u << H) - 1u), tb, guard, ul);
#pragma unroll
        for (int j = 0; j < H; ++j) r[j] = g_node_e(v[j], v[j + H], xl << (31 - j), tb);
        const uint32_t xr = sc_block<H>(r, fz >> H, tb, guard, ur);
        ubits = ul | (ur << H);
        return (xl ^ xr) | (xr << H);
    } else {
        ubits = 0u;                        // (S = 1 is all frozen or all unfrozen: handled above)
        return 0u;
    }
}
__global__ __launch_bounds__(64) void sc_lat_kernel(PolarScParams p) {
    const int lane = threadIdx.x;
    const int N = p.N, K = p.K;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double *tabs = reinterpret_cast<double *>(smem);
    double *a = tabs + 324;                                           // [2 N]
    const int words = (N + 31) / 32;
    uint32_t *bw = reinterpret_cast<uint32_t *>(a + 2 * (size_t)N);   // partial sums, one bit per leaf position
    uint32_t *uw = bw + words;                                        // decisions
    uint32_t *lops = uw + words;                                      // the schedule, staged once per block: a scalar load per op
                                                                      // misses the constant cache every 16 ops (~1 us each for a lone wave)
    for (int i = lane; i < 322; i += 64) tabs[i] = p.tabs[i];
    for (int i = lane; i < p.n_ops; i += 64) lops[i] = p.ops[i];
    if (lane == 0) lops[p.n_ops] = 0u;
    wave_mem_fence();
    const Tabs tb = {tabs, tabs + 64, tabs + 64 + 129};
    long Bv = p.B;
    if (p.n_dev && (long)*p.n_dev < Bv) Bv = (long)*p.n_dev;
    for (long cw = blockIdx.x; cw < Bv; cw += gridDim.x) {
#ifdef SCLAT_PROF
        const u64 kstart = __builtin_readcyclecounter();
#endif
        u64 guard = 0;
        for (int i = lane; i < words; i += 64) { bw[i] = 0u; uw[i] = 0u; }
        {   // channel row -> stored form, kernel element order (element e = channel position bitrev_n(e)); input guard as
            // sc8_front_kernel (ed_from_channel). The loads are coalesced and eight of them are in flight per lane.
            bool any = false;
            for (int i0 = 0; i0 < N; i0 += 64 * 8) {
                double x[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int i = i0 + 64 * k + lane;
                    x[k] = (i < N) ? (p.llr_f32 ? (double)reinterpret_cast<const float *>(p.llr)[(size_t)cw * N + i]
                                                : reinterpret_cast<const double *>(p.llr)[(size_t)cw * N + i]) : 1.0;
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int i = i0 + 64 * k + lane;
                    bool f;
                    const double v = ed_from_channel(x[k], tb, f);
                    any |= f;
                    if (i < N) a[N + (__brev((unsigned)i) >> (32 - p.n))] = v;
                }
            }
            guard |= __builtin_amdgcn_ballot_w64(any);
        }
        wave_mem_fence();
        // one layer visit: layer of size S = 2^sh from the layer above it (f), or with the partial sums at [base, base + S) (g)
        auto visit = [&](int type, int sh, int base) {
            const int S = 1 << sh;
            const double *src = a + 2 * (size_t)S;
            double *dst = a + S;
            if (S >= 64) {
                for (int j0 = lane; j0 < S; j0 += 64 * SCLAT_U) {
                    double va[SCLAT_U], vb[SCLAT_U];
#pragma unroll
                    for (int k = 0; k < SCLAT_U; ++k) if (j0 + 64 * k < S) { va[k] = src[j0 + 64 * k]; vb[k] = src[j0 + 64 * k + S]; }
#pragma unroll
                    for (int k = 0; k < SCLAT_U; ++k) if (j0 + 64 * k < S) {
                        double y;
                        if (type == 0) y = f_node_e(va[k], vb[k], guard);
                        else {
                            const int jj = base + j0 + 64 * k;
                            y = g_node_e(va[k], vb[k], bw[jj >> 5] << (31 - (jj & 31)), tb);
                        }
                        dst[j0 + 64 * k] = y;
                    }
                }
            } else {
                const bool in = lane < S;
                const double va = in ? src[lane] : 0.5, vb = in ? src[lane + S] : 0.5;      // (idle lanes: harmless operands)
                double y;
                if (type == 0) y = f_node_e(va, vb, guard);
                else {
                    const int jj = base + (in ? lane : 0);
                    y = g_node_e(va, vb, bw[jj >> 5] << (31 - (jj & 31)), tb);
                }
                if (in) dst[lane] = y;
            }
            wave_mem_fence();
        };
#ifdef SCLAT_PROF
        u64 pacc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, pcnt[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        u64 pt0 = __builtin_readcyclecounter();
        const u64 pstart = pt0;
#define SCLAT_TICK(k) { const u64 t_ = __builtin_readcyclecounter(); pacc[k] += t_ - pt0; pcnt[k] += 1; pt0 = t_; }
#else
#define SCLAT_TICK(k)
#endif
        // partial sums of a completed right child (node [b, b + s)) folded into its left sibling, c levels up: the host counts
        // the combine steps that follow an op instead of scheduling them as ops of their own (a quarter of the schedule)
        auto combine_chain = [&](int b, int s, int c) {
            for (; c > 0; --c) {
                const int lb = b - s;
                if (s >= 32) {
                    for (int w = lane; w < s / 32; w += 64) bw[(lb >> 5) + w] ^= bw[(b >> 5) + w];
                } else if (lane == 0) {
                    const int sft = lb & 31;
                    uint32_t x = bw[lb >> 5];
                    x ^= ((x >> (sft + s)) & ((1u << s) - 1u)) << sft;
                    bw[lb >> 5] = x;
                }
                wave_mem_fence();
                b = lb; s *= 2;
            }
        };
        uint32_t op_next = (uint32_t)__builtin_amdgcn_readfirstlane((int)lops[0]);
        for (int io = 0; io < p.n_ops; ++io) {
            const uint32_t op = op_next;
            const uint32_t opv = lops[io + 1];                           // (LDS, one op ahead; LDS operations return in order:
                                                                         //  no wait beyond the one the visit's own reads need)
            const int type = (int)(op & 7u), sh = (int)((op >> 3) & 15u), base = (int)((op >> 8) & 0xFFFFu);
#ifdef SCLAT_PROF
            const int pk_ = (type <= 1) ? (sh >= 6 ? 8 : type) : type;     // 0 f (S < 64), 1 g (S < 64), 8 f/g (S >= 64), 3, 4, 6, 7
#endif
            const int extra = (int)((op >> 24) & 3u);                   // F steps of the child chain the host folded into this op
            const int S = 1 << sh;
            if (type <= 1) {
                visit(type, sh, base);
                for (int d = 1; d <= extra; ++d) visit(0, sh - d, base);
            } else if (type == 3) {
                // ---- all-unfrozen subtree (see sc8_decode_kernel): hard decisions of its root, the exact-zero and
                // smallest-leaf-magnitude guards, then the polar transform of the sign bits
                const double *src = a + S;
                const int R = S >= 64 ? S / 64 : 1;
                const int sft = base & 31;
                bool zero = false;
                double q = 1.0;
                for (int r = 0; r < R; ++r) {
                    const int j = 64 * r + lane;
                    const bool in = j < S;
                    const double v = in ? src[j] : 0.5;
                    zero |= in && fabs(v) == 1.0;
                    const double m = fabs(v);
                    const double t = in ? __builtin_fmax(1.0 - 2.0 * ((m > 1.0) ? 0.0 : m), 0.0) : 1.0;
                    q = __builtin_fmax(q * t, 1e-300);
                    const u64 bal = __builtin_amdgcn_ballot_w64(in && ed_is_neg(v));
                    if (lane == 0) {
                        if (S >= 64) { bw[(base + 64 * r) >> 5] = (uint32_t)bal; bw[((base + 64 * r) >> 5) + 1] = (uint32_t)(bal >> 32); }
                        else if (S == 32) bw[base >> 5] = (uint32_t)bal;
                        else bw[base >> 5] |= ((uint32_t)bal & ((1u << S) - 1u)) << sft;
                    }
                }
                guard |= __builtin_amdgcn_ballot_w64(zero);
                // (every lane's own product >= 0.75 => the product over the 64 lanes >= 0.75^64 > 1e-8: the usual case — the root
                // values of an all-unfrozen node are large — decided by one compare instead of six cross-lane stages)
                if (__builtin_amdgcn_ballot_w64(q < 0.75)) {
#pragma unroll
                    for (int off = 1; off < 64; off <<= 1) q = __builtin_fmax(q * __shfl_xor(q, off, 64), 1e-300);
                }
                if (__builtin_amdgcn_ballot_w64(q < 1e-8)) {
                    double T = 1.0;                                        // the cheap bound failed: the product itself
                    for (int r = 0; r < R; ++r) {
                        const int j = 64 * r + lane;
                        const bool in = j < S;
                        const double v = in ? src[j] : 0.5;
                        const double m = __builtin_fmin(fabs(v), 1.0);
                        const double t = (in && fabs(v) <= 1.0) ? ed_div(1.0 - m, 1.0 + m) : 1.0;
                        T = __builtin_fmax(T * t, 1e-300);
                    }
#pragma unroll
                    for (int off = 1; off < 64; off <<= 1) T = __builtin_fmax(T * __shfl_xor(T, off, 64), 1e-300);
                    guard |= __builtin_amdgcn_ballot_w64(T < 1e-8);
                }
                wave_mem_fence();
                if (S <= 32) {
                    if (lane == 0) {
                        const uint32_t m = (S == 32) ? 0xFFFFFFFFu : ((1u << S) - 1u);
                        const uint32_t x = (bw[base >> 5] >> sft) & m;
                        uw[base >> 5] |= (uint32_t)bits_transform((u64)x, S) << sft;
                    }
                } else {
                    const int nw = S / 32, w0b = base >> 5;
                    for (int b0 = lane; b0 < nw; b0 += 64) {
                        uint32_t x = 0;
                        for (int b1 = b0; b1 < nw; ++b1) if ((b1 & b0) == b0) x ^= bw[w0b + b1];
                        uw[w0b + b0] = (uint32_t)bits_transform((u64)x, 32);
                    }
                }
                wave_mem_fence();
                combine_chain(base, S, extra | ((int)((op >> 26) & 3u) << 2));
            } else if (type == 7) {
                // ---- mixed node of size 8, decoded in registers (sc_block): its eight root values are one broadcast read
#ifndef SCLAT_BLOCK_LANES
#define SCLAT_BLOCK_LANES 64
#endif
                if (lane < SCLAT_BLOCK_LANES) {          // (every lane would compute the same thing)
                    double v[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = a[8 + j];
                    uint32_t ub;
                    const uint32_t xb = sc_block<8>(v, (op >> 24) & 0xFFu, tb, guard, ub);
                    if (lane == 0) {
                        bw[base >> 5] |= xb << (base & 31);
                        uw[base >> 5] |= ub << (base & 31);
                    }
                }
                wave_mem_fence();
                combine_chain(base, 8, sh);                              // (type 7: the size is fixed, the field carries the combine count)
            } else if (type == 4) {
                // ---- combine: left half ^= right half (child size S)
                if (S >= 32) {
                    for (int w = lane; w < S / 32; w += 64) bw[(base >> 5) + w] ^= bw[((base + S) >> 5) + w];
                } else if (lane == 0) {
                    const int sft = base & 31;
                    uint32_t x = bw[base >> 5];
                    x ^= ((x >> (sft + S)) & ((1u << S) - 1u)) << sft;
                    bw[base >> 5] = x;
                }
                wave_mem_fence();
            } else if (type == 6) {
                // ---- all-frozen subtree: the +inf path-metric bound of sc8_decode_kernel (sum of the root's |x| beyond 690)
                const double *src = a + S;
                const int R = S >= 64 ? S / 64 : 1;
                double P = 1.0;
                bool bad = false;
                for (int r = 0; r < R; ++r) {
                    const int j = 64 * r + lane;
                    const bool in = j < S;
                    const double m = in ? fabs(src[j]) : 1.0;
                    bad |= in && m > 1.0;
                    P *= __builtin_fmin(m, 1.0);
                    bad |= P < 1e-300;
                    P = __builtin_fmax(P, 1e-300);
                }
                // (every lane's own product >= 1e-4 => the product over the 64 lanes >= 1e-256: no cross-lane stage in the usual case)
                if (__builtin_amdgcn_ballot_w64(P < 1e-4)) {
#pragma unroll
                    for (int off = 1; off < 64; off <<= 1) {
                        P *= __shfl_xor(P, off, 64);
                        bad |= P < 1e-300;
                        P = __builtin_fmax(P, 1e-300);
                    }
                }
                guard |= __builtin_amdgcn_ballot_w64(bad);
                combine_chain(base, S, extra | ((int)((op >> 26) & 3u) << 2));
            }
            op_next = (uint32_t)__builtin_amdgcn_readfirstlane((int)opv);
#ifdef SCLAT_PROF
            SCLAT_TICK(pk_)
#endif
        }
        wave_mem_fence();
#ifdef SCLAT_PROF
        if (lane == 0 && p.a_scr) {
            u64 *o_ = reinterpret_cast<u64 *>(p.a_scr);
            for (int k = 0; k < 10; ++k) { atomicAdd(o_ + k, pacc[k]); atomicAdd(o_ + 10 + k, pcnt[k]); }
            atomicAdd(o_ + 20, pstart - kstart);                              // front pass
            atomicAdd(o_ + 21, __builtin_readcyclecounter() - kstart);       // everything but the output
        }
#endif
        if (guard != 0 && lane == 0) atomicOr(&p.flag_words[cw >> 5], 1u << (cw & 31));
        if (p.flag_bytes && lane == 0) p.flag_bytes[cw] = (guard != 0) ? 1 : 0;
        for (int b = lane; b < K; b += 64) {
            const unsigned pos = p.order[b];
            p.out[(size_t)cw * K + b] = (uint8_t)((uw[pos >> 5] >> (pos & 31)) & 1u);
        }
        wave_mem_fence();
    }
}
size_t polar_sc_lat_lds_bytes(int N, int n_ops) { return 324 * 8 + (size_t)2 * N * 8 + (size_t)2 * ((N + 31) / 32) * 4 + ((size_t)n_ops + 2) * 4; }
int polar_sc_lat_max_log() { return 12; }       // 16 N bytes of LDS per wave: 64 KiB at N = 4096
hipError_t polar_launch_sc_lat(const PolarScParams &p, int blocks, hipStream_t st) {
    const size_t lds = polar_sc_lat_lds_bytes(p.N, p.n_ops);
    if (lds > 48 * 1024) {        // (per launch: the attribute belongs to the function on the current device; the host only comes
                                  // here when `lds` fits the device's limit — a refusal is an error, not something to launch through)
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(sc_lat_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(sc_lat_kernel, dim3(blocks), dim3(64), lds, st, p);
    return hipGetLastError();
}

// flag bit words -> work list of the fallback pass (order irrelevant: every codeword is independent)
__global__ __launch_bounds__(256) void sc_collect_kernel(const unsigned int *flag_words, long B, const unsigned *n_dev, uint32_t *list, unsigned *count) {
    if (n_dev && (long)*n_dev < B) B = (long)*n_dev;
    const long nw = (B + 31) / 32;
    for (long w = (long)blockIdx.x * 256 + threadIdx.x; w < nw; w += (long)gridDim.x * 256) {
        unsigned int m = flag_words[w];
        while (m) {
            const int b = __builtin_ctz(m);
            m &= m - 1u;
            const long cw = w * 32 + b;
            if (cw < B) list[atomicAdd(count, 1u)] = (uint32_t)cw;
        }
    }
}
hipError_t polar_launch_sc_collect(const unsigned int *flag_words, long B, const unsigned *n_dev, uint32_t *list, unsigned *count, hipStream_t st) {
    long blocks = ((B + 31) / 32 + 255) / 256;
    if (blocks > 256) blocks = 256;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(sc_collect_kernel, dim3((unsigned)blocks), dim3(256), 0, st, flag_words, B, n_dev, list, count);
    return hipGetLastError();
}
