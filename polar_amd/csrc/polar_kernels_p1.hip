// polar_kernels_p1.hip — probability-domain decoders of the class surface:
//   * scl_decode_p1_kernel : PolarCode::decode_scl_p1 (PolarCode.cpp:110-128) -> decode_scl with
//     recursivelyCalcP (PolarCode.cpp:375-420, cross-path max-normalisation per layer),
//     probability forks (:509-511) and the probability final select (:631-637).
//   * sc_p1_kernel : PolarM decode_sc_p1 -> polar_decode / cnop / vnop (PolarCode.m:290-295, 870-895).
// No driver of the reference calls these (decode_scl_p1 is commented out at PolarCode.cpp:755), so
// they are built for completeness of the class surface with the same lane-per-path layout as the
// LLR kernel but without its tuning: every layer lives in the per-wave global scratch.
// -ffp-contract=off is required (the reference's products and sums are separately rounded).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "polar_kernels.h"
#include "polar_device.h"

template <int GS>
__global__ __launch_bounds__(64) void scl_decode_p1_kernel(PolarDecodeParams p) {
    constexpr int G = 64 / GS;
    const int lane = threadIdx.x;
    const int lig = lane & (GS - 1);
    const int gbase = lane & ~(GS - 1);
    const int grp = lane / GS;
    const int n = p.n, N = p.N, K = p.K, L = p.L;
    const u64 gmask = (GS == 64) ? ~0ull : ((1ull << GS) - 1ull);
    const u64 below = (1ull << lig) - 1ull;

    __shared__ double sortbuf[128];
    __shared__ unsigned char stack_s[64], srcof_s[64];
    volatile unsigned char *stackv = stack_s, *srcof = srcof_s;

    // per-wave scratch: layer of size S starts at element S (sizes 1..N/2), pairs (p0,p1)
    double2 *g_p = reinterpret_cast<double2 *>(p.llr_scr) + (size_t)blockIdx.x * (size_t)N * 64;
    const int cwords = (N >= 128) ? (N / 32 - 2) : 0;
    uint32_t *g_cl = p.c_scr + (size_t)blockIdx.x * 2 * (size_t)cwords * 64;
    uint32_t *g_cr = g_cl + (size_t)cwords * 64;
    uint32_t *g_hist = p.hist_scr + (size_t)blockIdx.x * (size_t)p.W * 64;

    for (long g0 = (long)blockIdx.x * G; g0 < p.B; g0 += (long)gridDim.x * G) {
        const long cw = g0 + grp;
        const bool valid = (cw < p.B);
        const double *in_p0 = p.p0 + (size_t)(valid ? cw : 0) * N;
        const double *in_p1 = p.llr + (size_t)(valid ? cw : 0) * N;     // p1 travels in the `llr` field

        bool active = valid && (lig == L - 1);
        int sp = L - 1;
        if (lig < L - 1) stackv[gbase + lig] = (unsigned char)lig;
        P16 pL = {0, 0}, pC = {0, 0};
        u64 clsmall = 0;
        uint32_t hword = 0;
        int origin = lig;
        unsigned t = 0;
        double lp0 = 0.0, lp1 = 0.0;      // leaf pair of this path (after normalisation)
        unsigned ubit = 0;
        wave_mem_fence();

        for (int phi = 0; phi < N; ++phi) {
            // ---------------- recursivelyCalcP(n, phi): PolarCode.cpp:375-420 ----------------
            const int lam_top = phi ? (n - __builtin_ctz((unsigned)phi)) : 1;
            for (int lam = lam_top; lam <= n; ++lam) {
                const int sh = n - lam;
                const int S = 1 << sh;
                const bool odd = (phi >> sh) & 1;
                double sig = 0.0;
                double2 *outp = g_p + (size_t)S * 64 + lane;
                // Round 6: U elements per pass with all their loads issued before the first use (the element-by-element loop paid a
                // memory round trip per element), and layers of size <= 8 are normalised in registers and written ONCE (they were
                // written, fenced, re-read, divided and written again). Same operations on the same operands: max is
                // order-independent, every product, sum and quotient is the one the loop computed.
                auto visit = [&](auto UU) {
                    constexpr int U = decltype(UU)::value;
                    const bool in_regs = (S <= U);                 // the whole layer is this one pass
                    double r0[U], r1[U];
                    if (active) {
                        const int pin = (lam > 1) ? pL.get(sh + 1) : 0;
                        const double2 *inp = g_p + (size_t)(2 * S) * 64 + gbase + pin;
                        uint32_t cbits = 0;
                        const uint32_t *cwp = nullptr;
                        if (odd) {
                            if (S <= 32) cbits = (uint32_t)(clsmall >> S);
                            else cwp = g_cl + (size_t)(S / 32 - 2) * 64 + gbase + pC.get(sh);
                        }
                        for (int j0 = 0; j0 < S; j0 += U) {
                            double a0[U], a1[U], b0[U], b1[U];
#pragma unroll
                            for (int k = 0; k < U; ++k) {
                                const int j = j0 + k;
                                if (lam == 1) {
                                    const unsigned idx = __brev((unsigned)j) >> (32 - n);
                                    a0[k] = in_p0[idx]; a1[k] = in_p1[idx]; b0[k] = in_p0[idx + 1]; b1[k] = in_p1[idx + 1];
                                } else {
                                    const double2 a = inp[(size_t)j * 64], b = inp[(size_t)(j + S) * 64];
                                    a0[k] = a.x; a1[k] = a.y; b0[k] = b.x; b1[k] = b.y;
                                }
                            }
                            if (odd && S > 32 && (j0 & 31) == 0) cbits = cwp[(size_t)(j0 >> 5) * 64];
#pragma unroll
                            for (int k = 0; k < U; ++k) {
                                const int j = j0 + k;
                                double o0, o1;
                                if (!odd) {                                   // :392-395
                                    o0 = 0.5 * (a0[k] * b0[k] + a1[k] * b1[k]);
                                    o1 = 0.5 * (a1[k] * b0[k] + a0[k] * b1[k]);
                                } else {                                      // :398-400
                                    const unsigned u = (cbits >> (j & 31)) & 1u;
                                    o0 = (0.5 * (u ? a1[k] : a0[k])) * b0[k];
                                    o1 = (0.5 * (u ? a0[k] : a1[k])) * b1[k];
                                }
                                sig = (sig < o0) ? o0 : sig;                  // :402-403
                                sig = (sig < o1) ? o1 : sig;
                                r0[k] = o0; r1[k] = o1;
                            }
                            if (!in_regs) {
#pragma unroll
                                for (int k = 0; k < U; ++k) outp[(size_t)(j0 + k) * 64] = make_double2(r0[k], r1[k]);
                            }
                        }
                        pL.set(sh, lig);
                    }
                    // sigma over ALL active paths of the codeword, then normalise (:409-419)
                    sig = group_reduce<GS, true>(sig, lane);
                    if (in_regs) {
                        if (active) {
#pragma unroll
                            for (int k = 0; k < U; ++k) {
                                if (sig != 0) { r0[k] = r0[k] / sig; r1[k] = r1[k] / sig; }
                                outp[(size_t)k * 64] = make_double2(r0[k], r1[k]);
                            }
                            lp0 = r0[0]; lp1 = r1[0];             // S == 1 at the last layer: the leaf pair
                        }
                    } else {
                        wave_mem_fence();
                        if (active && sig != 0) {
                            for (int j0 = 0; j0 < S; j0 += U) {
                                double2 v[U];
#pragma unroll
                                for (int k = 0; k < U; ++k) v[k] = outp[(size_t)(j0 + k) * 64];
#pragma unroll
                                for (int k = 0; k < U; ++k) { v[k].x = v[k].x / sig; v[k].y = v[k].y / sig; outp[(size_t)(j0 + k) * 64] = v[k]; }
                            }
                        }
                    }
                    wave_mem_fence();
                };
                if (S == 1) visit(std::integral_constant<int, 1>{});
                else if (S == 2) visit(std::integral_constant<int, 2>{});
                else if (S == 4) visit(std::integral_constant<int, 4>{});
                else visit(std::integral_constant<int, 8>{});
            }

            const bool frozen = p.frozen[phi] != 0;
            ubit = 0;
            if (!frozen) {
                // continuePaths_UnfrozenBit, probability forks: PolarCode.cpp:489-607 (:509-511)
                double pf0 = __builtin_nan(""), pf1 = __builtin_nan("");
                if (active) { pf0 = lp0; pf1 = lp1; }
                const u64 actm = (__ballot(active) >> gbase) & gmask;
                const int nact = __popcll(actm);
                const int rho = (2 * nact < L) ? 2 * nact : L;
                bool c0 = active, c1 = active;
                const bool need = (2 * nact > L);
                // Exact fast path (round 6; the LLR kernel's, in the probability domain): the list is full and every path's better
                // fork strictly beats every path's worse fork => the L survivors are the L better forks in whatever order they
                // rank: nobody is killed or cloned and the 2 L-entry ranking loop (half of this kernel's time at L = 32) is
                // skipped. A tie anywhere (pf0 == pf1 in a path, or a better fork equal to some worse fork) fails the strict test
                // and takes the ranking below, where the reference's index order decides.
                bool fastp;
                {
                    const double gd = active ? ((pf0 > pf1) ? pf0 : pf1) : __builtin_inf();
                    const double bd = active ? ((pf0 > pf1) ? pf1 : pf0) : -__builtin_inf();
                    const double gmin = group_reduce<GS, false>(gd, lane), bmax = group_reduce<GS, true>(bd, lane);
                    fastp = __all((nact == 0) || (nact == L && gmin > bmax));
                }
                if (fastp) {
                    c0 = active && (pf0 > pf1);
                    c1 = active && !c0;
                } else if (__any(need)) {
                    sortbuf[2 * lane] = pf0;
                    sortbuf[2 * lane + 1] = pf1;
                    wave_mem_fence();
                    int r0 = 0, r1 = 0;
                    const double *sb = sortbuf + 2 * gbase;
                    const int i0 = 2 * lig;
                    for (int i = 0; i < 2 * GS; ++i) {
                        const double v = sb[i];
                        r0 += (i < i0) ? (v >= pf0) : (v > pf0);
                        r1 += (i <= i0) ? (v >= pf1) : (v > pf1);
                    }
                    if (need) { c0 = active && (r0 < rho); c1 = active && (r1 < rho); }
                    wave_mem_fence();
                }
                const bool kill = active && !c0 && !c1;
                const bool both = c0 && c1;
                const u64 km = (__ballot(kill) >> gbase) & gmask;
                const u64 bm = (__ballot(both) >> gbase) & gmask;
                srcof[lane] = (unsigned char)lig;
                if (kill) stackv[gbase + sp + __popcll(km & below)] = (unsigned char)lig;
                sp += __popcll(km);
                wave_mem_fence();
                if (both) {
                    int lp = stackv[gbase + sp - 1 - __popcll(bm & below)];
                    srcof[gbase + lp] = (unsigned char)lig;
                }
                sp -= __popcll(bm);
                wave_mem_fence();
                const int src = srcof[lane];
                const bool is_clone = (src != lig);
                ubit = c0 ? 0u : 1u;
                if (__any(is_clone)) {
                    const int sl = gbase + src;
                    u64 a0 = shfl_u64(pL.lo, sl), a1 = shfl_u64(pL.hi, sl);
                    u64 b0 = shfl_u64(pC.lo, sl), b1 = shfl_u64(pC.hi, sl);
                    u64 cs = shfl_u64(clsmall, sl);
                    uint32_t hw = __shfl(hword, sl, 64);
                    int og = __shfl(origin, sl, 64);
                    double q0 = shfl_d(lp0, sl), q1 = shfl_d(lp1, sl);
                    if (is_clone) {
                        ubit = 1u;
                        pL.lo = a0; pL.hi = a1; pC.lo = b0; pC.hi = b1;
                        clsmall = cs; hword = hw; origin = og; lp0 = q0; lp1 = q1;
                    }
                }
                active = (active && !kill) || is_clone;
                if (active) hword |= ubit << (t & 31);
                if ((t & 31) == 31) {
                    const int w = (int)(t >> 5);
                    if (__any(active && origin != lig)) {
                        for (int wi = 0; wi < w; ++wi) {
                            uint32_t v = 0;
                            if (active) v = g_hist[(size_t)wi * 64 + gbase + origin];
                            wave_mem_fence();
                            if (active) g_hist[(size_t)wi * 64 + lane] = v;
                        }
                    }
                    if (active) { g_hist[(size_t)w * 64 + lane] = hword; origin = lig; hword = 0; }
                    wave_mem_fence();
                }
                ++t;
            }

            // partial sums: same as the LLR kernel (recursivelyUpdateC, PolarCode.cpp:457-473)
            if ((phi & 1) == 0) {
                if (active) clsmall = (clsmall & ~2ull) | ((u64)ubit << 1);
            } else {
                int S = 1, ph = phi;
                uint32_t X = ubit;
                for (;;) {
                    if (4 * S > N) break;
                    const int psi = ph >> 1;
                    const bool to_right = (psi & 1);
                    const int sh = __builtin_ctz((unsigned)S);
                    if (S <= 16) {
                        uint32_t cl = (uint32_t)(clsmall >> S) & ((1u << S) - 1u);
                        uint32_t nw = (cl ^ X) | (X << S);
                        if (!to_right) {
                            const int S2 = 2 * S;
                            const u64 m = ((S2 == 32) ? 0xFFFFFFFFull : ((1ull << S2) - 1ull)) << S2;
                            if (active) clsmall = (clsmall & ~m) | ((u64)nw << S2);
                        }
                        X = nw;
                    } else if (S == 32) {
                        uint32_t cl = (uint32_t)(clsmall >> 32);
                        uint32_t *dst = (to_right ? g_cr : g_cl) + lane;
                        if (active) { dst[0] = cl ^ X; dst[64] = X; }
                        if (!to_right && active) pC.set(sh + 1, lig);
                    } else {
                        const int nwd = S / 32;
                        const uint32_t *cl = g_cl + (size_t)(nwd - 2) * 64 + gbase + pC.get(sh);
                        const uint32_t *cr = g_cr + (size_t)(nwd - 2) * 64 + lane;
                        uint32_t *dst = (to_right ? g_cr : g_cl) + (size_t)(2 * nwd - 2) * 64 + lane;
                        if (active) {
                            for (int w = 0; w < nwd; ++w) {
                                uint32_t r = cr[(size_t)w * 64];
                                uint32_t l = cl[(size_t)w * 64];
                                dst[(size_t)w * 64] = l ^ r;
                                dst[(size_t)(w + nwd) * 64] = r;
                            }
                            if (!to_right) pC.set(sh + 1, lig);
                        }
                    }
                    wave_mem_fence();
                    if (!to_right) break;
                    S *= 2;
                    ph = psi;
                }
            }
        }  // phi

        {   // final flush
            const int w = (int)(t >> 5);
            if (__any(active && origin != lig)) {
                for (int wi = 0; wi < w; ++wi) {
                    uint32_t v = 0;
                    if (active) v = g_hist[(size_t)wi * 64 + gbase + origin];
                    wave_mem_fence();
                    if (active) g_hist[(size_t)wi * 64 + lane] = v;
                }
            }
            if ((t & 31) != 0 && active) g_hist[(size_t)w * 64 + lane] = hword;
            wave_mem_fence();
        }
        const int Wused = (int)((t + 31) >> 5);
        bool pass = true;
        if (p.crc > 0) {
            uint32_t acc = 0;
            if (active) {
                for (int w = 0; w < Wused; ++w) {
                    uint32_t hw = g_hist[(size_t)w * 64 + lane];
                    for (int i = 0; i < p.crc; ++i)
                        acc ^= (uint32_t)(__popc(hw & p.crc_mask[(size_t)i * p.W + w]) & 1) << i;
                }
            }
            pass = (acc == 0);
        }
        const u64 passm = (__ballot(active && pass) >> gbase) & gmask;
        const bool cand = active && (pass || passm == 0);
        // findMostProbablePath, probability branch (PolarCode.cpp:631-637): p_m[c_m[1]] strictly
        // greater than the running maximum (initially 0), lowest index on ties
        double key = cand ? (ubit ? lp1 : lp0) : -1.0;
        int kidx = lig;
#pragma unroll
        for (int off = GS / 2; off >= 1; off >>= 1) {
            double ok = shfl_d(key, lane ^ off);
            int oi = __shfl(kidx, lane ^ off, 64);
            if (ok > key || (ok == key && oi < kidx)) { key = ok; kidx = oi; }
        }
        const int win = (key > 0.0) ? kidx : 0;
        if (valid) {
            for (int b = lig; b < K; b += GS) {
                unsigned r = p.info_rank[b];
                uint32_t wd = g_hist[(size_t)(r >> 5) * 64 + gbase + win];
                p.out[(size_t)cw * K + b] = (uint8_t)((wd >> (r & 31)) & 1u);
            }
        }
        wave_mem_fence();
    }
}

// ---- PolarM decode_sc_p1: one lane per codeword, natural recursion made iterative -----------
// y layers and "hard" partial sums are doubles as in MATLAB (a leaf with y == 0.5 yields 0.5).
__global__ __launch_bounds__(64) void sc_p1_kernel(PolarScP1Params p) {
    const int lane = threadIdx.x;
    const int n = p.n, N = p.N, K = p.K;
    // per-wave scratch [elem][lane]: y layers (size S at offset S), xl / xr (same offsets), u[N]
    double *gy = p.scr + (size_t)blockIdx.x * (size_t)(4 * N) * 64;
    double *gxl = gy + (size_t)N * 64;
    double *gxr = gxl + (size_t)N * 64;
    double *gu = gxr + (size_t)N * 64;
    for (long c0 = (long)blockIdx.x * 64; c0 < p.B; c0 += (long)gridDim.x * 64) {
        const long cw = c0 + lane;
        const bool valid = cw < p.B;
        const double *y0 = p.p1 + (size_t)(valid ? cw : 0) * N;
        if (valid) {
            for (int phi = 0; phi < N; ++phi) {
                const int lam_top = phi ? (n - __builtin_ctz((unsigned)phi)) : 1;
                double leaf = 0.0;
                for (int lam = lam_top; lam <= n; ++lam) {
                    const int sh = n - lam, S = 1 << sh;
                    const bool odd = (phi >> sh) & 1;
                    for (int j = 0; j < S; ++j) {
                        double a, b;
                        if (lam == 1) {
                            unsigned idx = __brev((unsigned)j) >> (32 - n);
                            a = y0[idx]; b = y0[idx + 1];
                        } else {
                            a = gy[(size_t)(2 * S + j) * 64 + lane];
                            b = gy[(size_t)(2 * S + j + S) * 64 + lane];
                        }
                        double r;
                        if (!odd) r = a * (1 - b) + b * (1 - a);                       // cnop, PolarCode.m:889-891
                        else {
                            const double x = gxl[(size_t)(S + j) * 64 + lane];
                            const double w1 = x * (1 - a) + a * (1 - x);                // cnop(u1hardprev, y_odd)
                            r = w1 * b / (w1 * b + (1 - w1) * (1 - b));                 // vnop, :893-895
                        }
                        gy[(size_t)(S + j) * 64 + lane] = r;
                        leaf = r;
                    }
                }
                double x;
                if (p.frozen[phi]) x = 0.0;                                              // :875-876
                else { const double tt = 1 - 2 * leaf; x = (1 - (double)((tt > 0) - (tt < 0))) / 2; }   // :873
                gu[(size_t)phi * 64 + lane] = x;
                if ((phi & 1) == 0) gxl[(size_t)1 * 64 + lane] = x;
                else {
                    gxr[(size_t)1 * 64 + lane] = x;
                    int S = 1, ph = phi;
                    for (;;) {
                        if (4 * S > N) break;
                        const int psi = ph >> 1;
                        const bool to_right = psi & 1;
                        double *dst = (to_right ? gxr : gxl) + (size_t)(2 * S) * 64 + lane;
                        for (int j = 0; j < S; ++j) {
                            const double x1 = gxl[(size_t)(S + j) * 64 + lane], x2 = gxr[(size_t)(S + j) * 64 + lane];
                            dst[(size_t)j * 64] = x1 * (1 - x2) + x2 * (1 - x1);         // cnop(u1hard,u2hard) :885
                            dst[(size_t)(j + S) * 64] = x2;
                        }
                        if (!to_right) break;
                        S *= 2; ph = psi;
                    }
                }
            }
            for (int b = 0; b < K; ++b) p.out[(size_t)cw * K + b] = gu[(size_t)p.order[b] * 64 + lane];
        }
    }
}

// ---- the same decoder for SMALL batches: ONE codeword per wave, the elements of a layer spread over the 64 lanes, the whole
// state in LDS (round 6). PolarM's loop calls decode_sc_p1 once per codeword (main_MC_CC_Comparison.m:96): with one LANE per
// codeword a single call is 22 528 node evaluations in a row, each behind a round trip to the HBM scratch (7.6 ms at N = 2048).
// Same expressions per node, same element order, no reductions: the doubles are those of sc_p1_kernel bit for bit.
// LDS: y layers (size S at offset S: N doubles), xl / xr (N each), u (N): 4 N doubles = 64 KiB at N = 2048.
__global__ __launch_bounds__(64) void sc_p1_lat_kernel(PolarScP1Params p) {
    extern __shared__ double lds_p1[];
    const int lane = threadIdx.x;
    const int n = p.n, N = p.N, K = p.K;
    double *ly = lds_p1, *lxl = ly + N, *lxr = lxl + N, *lu = lxr + N;
    // A lone wave pays 4 cycles per instruction, ~ 50 per LDS round trip and a memory round trip (~ 1 us) for anything it reads from
    // global memory inside the leaf loop (tools/lone_wave_microbench.hip): the frozen flags are staged in LDS once per block, the
    // leaf's own node is evaluated by every lane from a broadcast read (not stored and read back), and the first level of the
    // partial-sum update takes its two bits from registers.
    unsigned char *lfz = reinterpret_cast<unsigned char *>(lu + N);
    for (int i = lane; i < N; i += 64) lfz[i] = p.frozen[i];
    wave_mem_fence();
    for (long cw = blockIdx.x; cw < p.B; cw += gridDim.x) {
        const double *y0 = p.p1 + (size_t)cw * N;
        double xprev = 0.0;                                                          // decision of the pair's left leaf
        for (int phi = 0; phi < N; ++phi) {
            const bool fz = lfz[phi] != 0;
            const int lam_top = phi ? (n - __builtin_ctz((unsigned)phi)) : 1;
            const int lam_hi = (n > 1) ? n - 1 : n;
            for (int lam = lam_top; lam <= lam_hi; ++lam) {
                const int sh = n - lam, S = 1 << sh;
                const bool odd = (phi >> sh) & 1;
                for (int j = lane; j < S; j += 64) {
                    double a, b;
                    if (lam == 1) {
                        const unsigned idx = __brev((unsigned)j) >> (32 - n);
                        a = y0[idx]; b = y0[idx + 1];
                    } else {
                        a = ly[2 * S + j]; b = ly[2 * S + j + S];
                    }
                    double r;
                    if (!odd) r = a * (1 - b) + b * (1 - a);                       // cnop, PolarCode.m:889-891
                    else {
                        const double x = lxl[S + j];
                        const double w1 = x * (1 - a) + a * (1 - x);                // cnop(u1hardprev, y_odd)
                        r = w1 * b / (w1 * b + (1 - w1) * (1 - b));                 // vnop, :893-895
                    }
                    ly[S + j] = r;
                }
                wave_mem_fence();
            }
            double leaf;
            if (n > 1) {
                // the leaf's node, by every lane: the two values of the layer of size 2 (a broadcast read), the left leaf's decision
                const double a = ly[2], b = ly[3];
                if ((phi & 1) == 0) leaf = a * (1 - b) + b * (1 - a);
                else {
                    const double w1 = xprev * (1 - a) + a * (1 - xprev);
                    leaf = w1 * b / (w1 * b + (1 - w1) * (1 - b));
                }
            } else leaf = ly[1];
            double x;
            if (fz) x = 0.0;                                                         // :875-876
            else { const double tt = 1 - 2 * leaf; x = (1 - (double)((tt > 0) - (tt < 0))) / 2; }   // :873
            if (lane == 0) lu[phi] = x;
            if ((phi & 1) == 0) {
                xprev = x;
                if (n == 1) { if (lane == 0) lxl[1] = x; wave_mem_fence(); }
            } else if (4 <= N) {
                // first level (S = 1) from registers: cnop(u1hard, u2hard) :885, then the generic walk from S = 2
                int ph = phi >> 1;
                bool to_right = ph & 1;
                if (lane == 0) {
                    double *dst = (to_right ? lxr : lxl) + 2;
                    dst[0] = xprev * (1 - x) + x * (1 - xprev);
                    dst[1] = x;
                }
                wave_mem_fence();
                int S = 2;
                while (to_right && 4 * S <= N) {
                    const int psi = ph >> 1;
                    to_right = psi & 1;
                    double *dst = (to_right ? lxr : lxl) + 2 * S;
                    for (int j = lane; j < S; j += 64) {
                        const double x1 = lxl[S + j], x2 = lxr[S + j];
                        dst[j] = x1 * (1 - x2) + x2 * (1 - x1);                         // cnop(u1hard, u2hard) :885
                        dst[j + S] = x2;
                    }
                    wave_mem_fence();
                    S *= 2; ph = psi;
                }
            }
        }
        wave_mem_fence();
        for (int b = lane; b < K; b += 64) p.out[(size_t)cw * K + b] = lu[p.order[b]];
        wave_mem_fence();
    }
}
size_t polar_sc_p1_lat_lds_bytes(int N) { return (size_t)4 * N * sizeof(double) + (size_t)N; }   // y, xl, xr, u + the frozen flags
hipError_t polar_launch_sc_p1_lat(const PolarScP1Params &p, int grid, hipStream_t st) {
    const size_t lds = polar_sc_p1_lat_lds_bytes(p.N);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(sc_p1_lat_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(sc_p1_lat_kernel, dim3(grid), dim3(64), lds, st, p);
    return hipGetLastError();
}

hipError_t polar_launch_decode_p1(const PolarDecodeParams &p, int gs, int grid, hipStream_t st) {
    switch (gs) {
        case 1: hipLaunchKernelGGL(scl_decode_p1_kernel<1>, dim3(grid), dim3(64), 0, st, p); break;
        case 2: hipLaunchKernelGGL(scl_decode_p1_kernel<2>, dim3(grid), dim3(64), 0, st, p); break;
        case 4: hipLaunchKernelGGL(scl_decode_p1_kernel<4>, dim3(grid), dim3(64), 0, st, p); break;
        case 8: hipLaunchKernelGGL(scl_decode_p1_kernel<8>, dim3(grid), dim3(64), 0, st, p); break;
        case 16: hipLaunchKernelGGL(scl_decode_p1_kernel<16>, dim3(grid), dim3(64), 0, st, p); break;
        case 32: hipLaunchKernelGGL(scl_decode_p1_kernel<32>, dim3(grid), dim3(64), 0, st, p); break;
        case 64: hipLaunchKernelGGL(scl_decode_p1_kernel<64>, dim3(grid), dim3(64), 0, st, p); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}
hipError_t polar_launch_sc_p1(const PolarScP1Params &p, int grid, hipStream_t st) {
    hipLaunchKernelGGL(sc_p1_kernel, dim3(grid), dim3(64), 0, st, p);
    return hipGetLastError();
}
