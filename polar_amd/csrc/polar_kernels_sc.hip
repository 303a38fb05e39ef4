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
// Layout: ONE LANE PER CODEWORD, 64 codewords per wave, every codeword follows the same host-built schedule
// (it depends on the frozen set only), so the wave never diverges. alpha arrays [layer][element][lane] (512-byte
// rows: coalesced), layers of size <= 8 in LDS, the rest in a per-wave global scratch; the channel values arrive
// TRANSPOSED in that same layout (sc_front_kernel). Node arithmetic: exp-domain stored form (polar_edom.h), i.e. one
// division per f or g and no transcendental anywhere (there is no path metric to feed).
// Partial sums and decisions are bits: a 64-leaf window in two 64-bit registers, flushed to [word][lane] arrays.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "polar_kernels.h"
#include "polar_device.h"
#include "polar_edom.h"

namespace {

constexpr int SC_SL = 8;                       // layers of size <= SC_SL live in LDS
#ifndef SC_UNROLL
#define SC_UNROLL 8
#endif
constexpr int SC_WPB = 4;                      // waves per block (share the exp/log tables of the rare paths)

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
// sc_front_kernel — channel LLRs [B][N] (double or float) -> stored form, bit-reversed element order, transposed to
// [group of 64 codewords][element][lane]; flags[cw] = 1 for codewords the SC kernel must not decide.
// One block = one group of 64 codewords x 32 consecutive input positions: 256-byte reads, 512-byte writes.
template <typename TIN>
__global__ __launch_bounds__(256) void sc_front_kernel(const TIN *llr, double *ech_t, unsigned int *flag_words, const double *tabs_g,
                                                       int n, long B, const unsigned *n_dev) {
    __shared__ double tabs[324];
    __shared__ double tile[32][65];
    for (int i = threadIdx.x; i < 322; i += 256) tabs[i] = tabs_g[i];
    __syncthreads();
    const Tabs tb = {tabs, tabs + 64, tabs + 64 + 129};
    if (n_dev && (long)*n_dev < B) B = (long)*n_dev;
    const int N = 1 << n;
    const long groups = (B + 63) / 64;
    const int chunks = N / 32;
    for (long t = blockIdx.x; t < groups * chunks; t += gridDim.x) {
        const long g = t / chunks;
        const int i0 = (int)(t % chunks) * 32;
        unsigned bad = 0;
        // thread (r, c): codeword g*64 + r + 8k, position i0 + c   (r = tid / 32, c = tid % 32)
        const int c = threadIdx.x & 31, r = threadIdx.x >> 5;
        for (int k = 0; k < 8; ++k) {
            const long cw = g * 64 + r + 8 * k;
            double v = 1.0;
            if (cw < B) {
                bool f;
                v = ed_from_channel((double)llr[(size_t)cw * N + i0 + c], tb, f);
                bad |= f ? 1u : 0u;
                if (f) atomicOr(&flag_words[cw >> 5], 1u << (cw & 31));
            }
            tile[c][r + 8 * k] = v;
        }
        __syncthreads();
        // element e = bitrev_n(i0 + cc): row of 64 lanes
        const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
        for (int k = 0; k < 8; ++k) {
            const int cc = q + 4 * k;
            const unsigned e = __brev((unsigned)(i0 + cc)) >> (32 - n);
            ech_t[((size_t)g * N + e) * 64 + lane] = tile[cc][lane];
        }
        __syncthreads();
        (void)bad;
    }
}

// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64 * SC_WPB, 4) void sc_decode_kernel(PolarScParams p) {
    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int wave_id = blockIdx.x * SC_WPB + wib;
    const int nwaves = gridDim.x * SC_WPB;
    const int n = p.n, N = p.N, K = p.K;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double *tabs = reinterpret_cast<double *>(smem);
    for (int i = threadIdx.x; i < 322; i += SC_WPB * 64) tabs[i] = p.tabs[i];
    __syncthreads();
    const Tabs tb = {tabs, tabs + 64, tabs + 64 + 129};
    double *lds = reinterpret_cast<double *>(smem + 324 * 8) + (size_t)wib * (2 * SC_SL - 1) * 64;   // [(2*SL-1)][64]
    const size_t big = (N > 2 * SC_SL) ? (size_t)(N - 2 * SC_SL) : 0;
    double *g_a = p.a_scr + (size_t)wave_id * big * 64;
    const int words = (N + 31) / 32;
    uint32_t *g_b = p.bits_scr + (size_t)wave_id * 2 * (size_t)words * 64;      // partial sums [words][64]
    uint32_t *g_u = g_b + (size_t)words * 64;                                    // decisions    [words][64]
    u64 guard = 0;
    long Bv = p.B;
    if (p.n_dev && (long)*p.n_dev < Bv) Bv = (long)*p.n_dev;
    const long groups = (Bv + 63) / 64;
    typedef const uint32_t __attribute__((address_space(4))) *kconst_u32;
    const kconst_u32 ops = (kconst_u32)(uintptr_t)p.ops;

    for (long g = wave_id; g < groups;) {
        const long cw = g * 64 + lane;
        const bool valid = cw < Bv;
        const double *ch = p.ech_t + (size_t)g * N * 64 + lane;          // element e at ch[e*64]
        u64 bcur = 0, ucur = 0;
        guard = 0;
        for (int io = 0; io < p.n_ops; ++io) {
            const uint32_t op = ops[io];
            const int type = (int)(op & 7u), sh = (int)((op >> 3) & 15u), base = (int)(op >> 8);
            const int S = 1 << sh;
            const int off = base & 63;
            if (type <= 1) {
                // ---- F / G: layer of size S from the layer of size 2S (the channel when 2S == N)
                // U elements per pass, 2U loads in flight before the first node (measured: U = 8 best; 16 spills)
                auto visit_u = [&](const double *src, double *dst, auto UU) {
                    constexpr int U = decltype(UU)::value;
                    if (type == 0) {
                        for (int j = 0; j < S; j += U) {
                            double a[U], b[U];
#pragma unroll
                            for (int k = 0; k < U; ++k) { a[k] = src[(size_t)(j + k) * 64]; b[k] = src[(size_t)(j + k + S) * 64]; }
#pragma unroll
                            for (int k = 0; k < U; ++k) dst[(size_t)(j + k) * 64] = f_node_e(a[k], b[k], guard);
                        }
                    } else {
                        uint32_t w = (uint32_t)(bcur >> off);
                        for (int j = 0; j < S; j += U) {
                            if (S >= 64 && (j & 31) == 0) w = g_b[(size_t)((base + j) >> 5) * 64 + lane];
                            double a[U], b[U];
#pragma unroll
                            for (int k = 0; k < U; ++k) { a[k] = src[(size_t)(j + k) * 64]; b[k] = src[(size_t)(j + k + S) * 64]; }
#pragma unroll
                            for (int k = 0; k < U; ++k) dst[(size_t)(j + k) * 64] = g_node_e(a[k], b[k], w << (31 - ((j + k) & 31)), tb);
                        }
                    }
                };
                auto visit = [&](const double *src, double *dst) {
                    if (S >= SC_UNROLL) visit_u(src, dst, std::integral_constant<int, SC_UNROLL>{});
                    else if (S == 8) visit_u(src, dst, std::integral_constant<int, 8>{});
                    else if (S == 4) visit_u(src, dst, std::integral_constant<int, 4>{});
                    else if (S == 2) visit_u(src, dst, std::integral_constant<int, 2>{});
                    else visit_u(src, dst, std::integral_constant<int, 1>{});
                };
                // (one instantiation per address-space combination: no FLAT accesses)
                if (2 * S == N) {
                    if (S <= SC_SL) visit(ch, lds + (size_t)(S - 1) * 64 + lane);
                    else visit(ch, g_a + (size_t)(S - 2 * SC_SL) * 64 + lane);
                } else if (2 * S <= SC_SL) visit(lds + (size_t)(2 * S - 1) * 64 + lane, lds + (size_t)(S - 1) * 64 + lane);
                else if (S <= SC_SL) visit(g_a + (size_t)(2 * S - 2 * SC_SL) * 64 + lane, lds + (size_t)(S - 1) * 64 + lane);
                else visit(g_a + (size_t)(2 * S - 2 * SC_SL) * 64 + lane, g_a + (size_t)(S - 2 * SC_SL) * 64 + lane);
                wave_mem_fence();
            } else if (type == 6) {
                // ---- all-frozen subtree: its leaves are not evaluated, but the reference's path metric would become
                // +inf at a frozen leaf with llr < -709.78 (PolarCode.cpp:483) and from then on every unfrozen decision
                // is the tie-break's 0, not the sign. No leaf of the subtree exceeds the sum of the root's |x|
                // (= -log of the product of the stored values): beyond 690 the codeword goes to the general kernel.
                auto bound = [&](const double *src) {
                    double P = 1.0;
                    bool bad = false;
                    for (int j = 0; j < S; ++j) {
                        const double m = fabs(src[(size_t)j * 64]);
                        bad |= m > 1.0;
                        P *= __builtin_fmin(m, 1.0);
                        bad |= P < 1e-300;
                        P = __builtin_fmax(P, 1e-300);
                    }
                    guard |= __builtin_amdgcn_ballot_w64(bad);
                };
                if (S <= SC_SL) bound(lds + (size_t)(S - 1) * 64 + lane);
                else bound(g_a + (size_t)(S - 2 * SC_SL) * 64 + lane);
            } else if (type == 2) {
                // ---- all-frozen subtree of >= 128 leaves: zeros (up to 64 leaves they are zeros of the window)
                for (int w = 0; w < S / 32; ++w) {
                    g_b[(size_t)((base >> 5) + w) * 64 + lane] = 0u;
                    g_u[(size_t)((base >> 5) + w) * 64 + lane] = 0u;
                }
            } else if (type == 3) {
                // ---- all-unfrozen subtree: hard decisions of its root layer, leaf decisions = their polar transform
                // (a root value that is exactly zero — e.g. 1 - 1 from +-1 inputs — breaks the identity: the reference
                // then decides through f(a, 0) = 0; such codewords go to the general kernel)
                auto hard = [&](const double *src) {
                    bool zero = false;
                    for (int j = 0; j < S; ++j) zero |= fabs(src[(size_t)j * 64]) == 1.0;
                    guard |= __builtin_amdgcn_ballot_w64(zero);
                    if (S <= 64) {
                        u64 x = 0;
                        for (int j = 0; j < S; ++j) x |= (u64)(ed_is_neg(src[(size_t)j * 64]) ? 1u : 0u) << j;
                        bcur |= x << off;
                        ucur |= bits_transform(x, S) << off;
                    } else {
                        for (int w = 0; w < S / 64; ++w) {
                            u64 x = 0;
                            for (int j = 0; j < 64; ++j) x |= (u64)(ed_is_neg(src[(size_t)(w * 64 + j) * 64]) ? 1u : 0u) << j;
                            g_b[(size_t)((base >> 5) + 2 * w) * 64 + lane] = (uint32_t)x;
                            g_b[(size_t)((base >> 5) + 2 * w + 1) * 64 + lane] = (uint32_t)(x >> 32);
                        }
                        wave_mem_fence();
                        // transform across 64-bit blocks first (block b ^= block b + h), then inside each block
                        const int nb = S / 64;
                        for (int b0 = 0; b0 < nb; ++b0) {
                            // u-block b0 = XOR of the x-blocks b with (b & b0) == b0 (Pascal-mod-2 rows of the transform)
                            u64 x = 0;
                            for (int b1 = b0; b1 < nb; ++b1) {
                                if ((b1 & b0) != b0) continue;
                                const u64 lo = g_b[(size_t)((base >> 5) + 2 * b1) * 64 + lane], hi = g_b[(size_t)((base >> 5) + 2 * b1 + 1) * 64 + lane];
                                x ^= lo | (hi << 32);
                            }
                            x = bits_transform(x, 64);
                            g_u[(size_t)((base >> 5) + 2 * b0) * 64 + lane] = (uint32_t)x;
                            g_u[(size_t)((base >> 5) + 2 * b0 + 1) * 64 + lane] = (uint32_t)(x >> 32);
                        }
                    }
                };
                if (S == N) hard(ch);
                else if (S <= SC_SL) hard(lds + (size_t)(S - 1) * 64 + lane);
                else hard(g_a + (size_t)(S - 2 * SC_SL) * 64 + lane);
            } else if (type == 4) {
                // ---- partial sums of a node from its children (PolarCode.cpp:457-473): left half ^= right half
                if (S < 64) {
                    const u64 m = (S == 32) ? 0xFFFFFFFFull : ((1ull << S) - 1ull);
                    bcur ^= ((bcur >> (off + S)) & m) << off;
                } else {
                    for (int w = 0; w < S / 32; ++w)
                        g_b[(size_t)((base >> 5) + w) * 64 + lane] ^= g_b[(size_t)(((base + S) >> 5) + w) * 64 + lane];
                    wave_mem_fence();
                }
            } else {
                // ---- the 64-leaf window is complete: flush (base = first leaf of the window)
                g_b[(size_t)(base >> 5) * 64 + lane] = (uint32_t)bcur;
                g_u[(size_t)(base >> 5) * 64 + lane] = (uint32_t)ucur;
                if (N > 32) {
                    g_b[(size_t)((base >> 5) + 1) * 64 + lane] = (uint32_t)(bcur >> 32);
                    g_u[(size_t)((base >> 5) + 1) * 64 + lane] = (uint32_t)(ucur >> 32);
                }
                bcur = 0; ucur = 0;
                wave_mem_fence();
            }
        }
        wave_mem_fence();
        // codewords whose |x| < 40 decisions were too close to call go to the general kernel as well
        if (valid && ((guard >> lane) & 1ull)) atomicOr(&p.flag_words[cw >> 5], 1u << (cw & 31));
        // ---- info bits: out[cw][b] = u[order[b]] (PolarCode.cpp:172-174); the 64 lanes write one codeword at a time
        for (int c = 0; c < 64; ++c) {
            const long cwc = g * 64 + c;
            if (cwc >= Bv) break;
            for (int b = lane; b < K; b += 64) {
                const unsigned pos = p.order[b];
                const uint32_t wd = g_u[(size_t)(pos >> 5) * 64 + c];
                p.out[(size_t)cwc * K + b] = (uint8_t)((wd >> (pos & 31)) & 1u);
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

// flag bit words -> byte flags of the fallback list builder (ed_collect_kernel reads bytes)
__global__ __launch_bounds__(256) void sc_flags_expand_kernel(const unsigned int *flag_words, uint8_t *flags, long B) {
    for (long cw = (long)blockIdx.x * 256 + threadIdx.x; cw < B; cw += (long)gridDim.x * 256)
        flags[cw] = (uint8_t)((flag_words[cw >> 5] >> (cw & 31)) & 1u);
}

size_t polar_sc_lds_bytes() { return 324 * 8 + (size_t)SC_WPB * (2 * SC_SL - 1) * 64 * 8; }
int polar_sc_waves_per_block() { return SC_WPB; }
int polar_sc_lds_layer() { return SC_SL; }

hipError_t polar_launch_sc_front(const void *llr, int llr_f32, double *ech_t, unsigned int *flag_words, const double *tabs,
                                 int n, long B, const unsigned *n_dev, hipStream_t st) {
    const long tiles = ((B + 63) / 64) * ((1L << n) / 32);
    const unsigned blocks = (unsigned)(tiles < 65536 ? (tiles ? tiles : 1) : 65536);
    if (llr_f32) hipLaunchKernelGGL(sc_front_kernel<float>, dim3(blocks), dim3(256), 0, st, (const float *)llr, ech_t, flag_words, tabs, n, B, n_dev);
    else hipLaunchKernelGGL(sc_front_kernel<double>, dim3(blocks), dim3(256), 0, st, (const double *)llr, ech_t, flag_words, tabs, n, B, n_dev);
    return hipGetLastError();
}
hipError_t polar_launch_sc_decode(const PolarScParams &p, int grid_waves, hipStream_t st) {
    hipLaunchKernelGGL(sc_decode_kernel, dim3((grid_waves + SC_WPB - 1) / SC_WPB), dim3(64 * SC_WPB), polar_sc_lds_bytes(), st, p);
    return hipGetLastError();
}
hipError_t polar_launch_sc_flags_expand(const unsigned int *flag_words, uint8_t *flags, long B, hipStream_t st) {
    long blocks = (B + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(sc_flags_expand_kernel, dim3((unsigned)blocks), dim3(256), 0, st, flag_words, flags, B);
    return hipGetLastError();
}
