// polar_construct.hip — Monte-Carlo code construction on the device:
// PolarM `monte_carlo`, receiver 'bicm' (PolarCode.m:143-196) with the genie-aided SC decoder
// `polar_decode_monte` (PolarCode.m:897-914, cnop/vnop :889-895).
//
// Per run: N random message bits (no frozen positions), polar transform, constellation mapping,
// AWGN at the design SNR, BICM demapper -> p1 = P(bit = 1) per position, then SC decoding in the
// probability domain where every partial sum comes from the KNOWN message (genie) and a position
// counts an error when the hard decision of its leaf disagrees with the message bit. The sum of
// those flags over the runs is the table the reference stores in CodeConstructionData/*.txt.
//
// Two kernels per batch of runs:
//   mc_front_kernel  — one wavefront per run, message/codeword bytes in LDS (same butterfly as the
//                      encoder of polar_channel.hip), writes p1[run][N] and the packed message;
//   mc_genie_kernel  — one LANE per run (64 runs per wavefront walk the N leaves in lockstep, so the
//                      per-position error count of a wave is one ballot + one atomic), y-layers and
//                      partial sums in a per-wave global scratch laid out [element][lane].
// Layer elements are kept in bit-reversed order as in the decoder kernels (a node combines
// elements j and j+S). HBM-bound double/byte work, no MFMA. Build with -ffp-contract=off: the
// channel front end must agree bit for bit with the host evaluation of include/polar_synth.h.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "polar_kernels.h"
#include "polar_synth.h"

namespace {

__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

__global__ __launch_bounds__(64) void mc_front_kernel(PolarConstructParams p) {
    extern __shared__ uint8_t sm[];
    volatile uint8_t *u = sm;                 // [N]
    const int lane = threadIdx.x;
    const int n = p.n, N = p.N;
    const int nb = polar_const_nbits(p.constellation);
    const int nsym = N / nb;
    const int words = (N + 31) / 32;
    for (long b = blockIdx.x; b < p.B; b += gridDim.x) {
        const uint64_t trial = p.trial0 + (uint64_t)b;
        // dummy_info = rand(1, N) < 0.5 (PolarCode.m:153): 32 message bits per lane and Philox word
        for (int q = lane; q < words; q += 64) {
            uint32_t r[4];
            polar_synth_mc_info_word(p.seed, trial, (uint32_t)(q >> 2), r);
            uint32_t w = r[q & 3];
            if (N < 32) w &= (1u << N) - 1u;
            p.info[(size_t)b * words + q] = w;
            for (int k = 0; k < 32 && 32 * q + k < N; ++k) u[32 * q + k] = (uint8_t)((w >> k) & 1u);
        }
        wave_sync();
        // polar_encode (PolarCode.m:855-867): [enc(u_odd xor u_even) enc(u_even)] recursively ==
        // in-place XOR butterfly over every stride, read out in bit-reversed order
        for (int it = 0; it < n; ++it) {
            const int inc = 1 << it;
            for (int q = lane; q < N / 2; q += 64) {
                int a = ((q >> it) << (it + 1)) | (q & (inc - 1));
                u[a] = (uint8_t)(u[a] ^ u[a + inc]);
            }
            wave_sync();
        }
        double *dp = p.p1 + (size_t)b * N;
        for (int i = nsym * nb + lane; i < N; i += 64) dp[i] = 0.5;                  // PolarCode.m:176
        for (int i = lane; i < nsym; i += 64) {
            int sym = 0;                                                              // Constellation.m:84-93
            for (int j = 0; j < nb; ++j) sym += (1 << j) * (int)u[__brev((unsigned)(i * nb + j)) >> (32 - n)];
            const double x = polar_const_point(p.constellation, sym) / p.cnorm;
            const double y = x + p.sigma * polar_synth_symbol_noise(p.seed, trial, (uint32_t)i);   // :170-172
            double q4[4];
            polar_synth_bicm_demap2(p.constellation, p.cnorm, y, p.n0, nullptr, q4);                // :178
            for (int j = 0; j < nb; ++j) dp[(size_t)i * nb + j] = q4[j];
        }
        wave_sync();
    }
}

__global__ __launch_bounds__(64) void mc_genie_kernel(PolarConstructParams p) {
    const int lane = threadIdx.x;
    const int n = p.n, N = p.N;
    const int words = (N + 31) / 32;
    // per-wave scratch: y layers [N][64] doubles (layer of size S at element offset S);
    // genie partial sums xl / xr [N][64] bytes (same offsets: left / right child of a node)
    double *gy = p.y_scr + (size_t)blockIdx.x * (size_t)N * 64;
    uint8_t *gxl = p.x_scr + (size_t)blockIdx.x * (size_t)(2 * N) * 64;
    uint8_t *gxr = gxl + (size_t)N * 64;
    for (long c0 = (long)blockIdx.x * 64; c0 < p.B; c0 += (long)gridDim.x * 64) {
        const long run = c0 + lane;
        const bool valid = run < p.B;
        const double *y0 = p.p1 + (size_t)(valid ? run : 0) * N;
        const uint32_t *inf = p.info + (size_t)(valid ? run : 0) * words;
        uint32_t iw = 0;
        for (int phi = 0; phi < N; ++phi) {
            if ((phi & 31) == 0) iw = inf[phi >> 5];
            const unsigned ubit = (iw >> (phi & 31)) & 1u;
            const int lam_top = phi ? (n - __builtin_ctz((unsigned)phi)) : 1;
            double leaf = 0.0;
            for (int lam = lam_top; lam <= n; ++lam) {
                const int sh = n - lam, S = 1 << sh;
                const bool odd = (phi >> sh) & 1;
                for (int j = 0; j < S; ++j) {
                    double a, b;
                    if (lam == 1) {
                        const unsigned idx = __brev((unsigned)j) >> (32 - n);     // pair (y(2k-1), y(2k)) of the recursion
                        a = y0[idx]; b = y0[idx + 1];
                    } else {
                        a = gy[(size_t)(2 * S + j) * 64 + lane];
                        b = gy[(size_t)(2 * S + j + S) * 64 + lane];
                    }
                    double r;
                    if (!odd) {
                        r = a * (1 - b) + b * (1 - a);                            // cnop, PolarCode.m:889-891
                    } else {
                        // cnop(u1hardprev, y_odd) with a hard bit is y_odd or 1 - y_odd exactly
                        const double w1 = gxl[(size_t)(S + j) * 64 + lane] ? 1 - a : a;
                        r = w1 * b / (w1 * b + (1 - w1) * (1 - b));                 // vnop, :893-895
                    }
                    gy[(size_t)(S + j) * 64 + lane] = r;
                    leaf = r;
                }
            }
            // PolarCode.m:899-905: correct iff (y > 0.5 and bit 1) or (y <= 0.5 and bit 0); NaN counts as <= 0.5 is false
            const bool ok = (leaf > 0.5 && ubit == 1u) || (leaf <= 0.5 && ubit == 0u);
            const unsigned long long em = __ballot(valid && !ok);
            if (lane == 0 && em) atomicAdd(p.num_err + phi, (unsigned long long)__popcll(em));
            // x = dummy_info (:906); parents: x = [cnop(x1, x2); x2] interleaved (:912)
            if ((phi & 1) == 0) gxl[(size_t)1 * 64 + lane] = (uint8_t)ubit;
            else {
                gxr[(size_t)1 * 64 + lane] = (uint8_t)ubit;
                int S = 1, ph = phi;
                for (;;) {
                    if (4 * S > N) break;
                    const int psi = ph >> 1;
                    const bool to_right = psi & 1;
                    uint8_t *dst = (to_right ? gxr : gxl) + (size_t)(2 * S) * 64 + lane;
                    for (int j = 0; j < S; ++j) {
                        const uint8_t x1 = gxl[(size_t)(S + j) * 64 + lane], x2 = gxr[(size_t)(S + j) * 64 + lane];
                        dst[(size_t)j * 64] = (uint8_t)(x1 ^ x2);
                        dst[(size_t)(j + S) * 64] = x2;
                    }
                    if (!to_right) break;
                    S *= 2; ph = psi;
                }
            }
        }
    }
}

}  // namespace

hipError_t polar_launch_mc_front(const PolarConstructParams &p, int grid, hipStream_t st) {
    hipLaunchKernelGGL(mc_front_kernel, dim3(grid), dim3(64), (size_t)p.N, st, p);
    return hipGetLastError();
}
hipError_t polar_launch_mc_genie(const PolarConstructParams &p, int grid, hipStream_t st) {
    hipLaunchKernelGGL(mc_genie_kernel, dim3(grid), dim3(64), 0, st, p);
    return hipGetLastError();
}
