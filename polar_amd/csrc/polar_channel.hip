// polar_channel.hip — the front of the hot path on the device: encoder
// (PolarCode::encode, PolarCode.cpp:60-91), the synthetic BPSK/AWGN channel + LLR
// (PolarCode.cpp:703-716, 744-753 with the counter-based generator of
// include/polar_synth.h) and the error counter (PolarCode.cpp:758-769).
//
// One wavefront per codeword, the N message bits staged as bytes in LDS; HBM-bound byte work
// (N*8 B of LLR written per codeword), no MFMA.  Must be built with -ffp-contract=off: the
// LLRs have to be bit-identical to the host evaluation of polar_synth.h.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "polar_kernels.h"
#include "polar_synth.h"

namespace {

__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// u[order[i]] = info[i]; crc bits; XOR butterfly: PolarCode.cpp:65-83. `u` is N bytes of LDS.
__device__ void encode_in_lds(volatile uint8_t *u, const uint8_t *info_lds, const PolarEncodeParams &p, int lane) {
    const int n = p.n, N = p.N, K = p.K;
    for (int i = lane; i < N; i += 64) u[i] = 0;
    wave_sync();
    for (int i = lane; i < K; i += 64) u[p.order[i]] = info_lds[i];
    for (int r = 0; r < p.crc; ++r) {
        unsigned par = 0;
        for (int j = lane; j < K; j += 64) par ^= (unsigned)(p.crcm[(size_t)r * K + j] & info_lds[j]);
        unsigned long long b = __ballot(par & 1u);
        if (lane == 0) u[p.order[K + r]] = (uint8_t)(__popcll(b) & 1);
    }
    wave_sync();
    for (int it = 0; it < n; ++it) {
        const int inc = 1 << it;
        for (int q = lane; q < N / 2; q += 64) {
            int a = ((q >> it) << (it + 1)) | (q & (inc - 1));
            u[a] = (uint8_t)(u[a] ^ u[a + inc]);
        }
        wave_sync();
    }
}

__global__ __launch_bounds__(64) void encode_kernel(PolarEncodeParams p) {
    extern __shared__ uint8_t sm[];
    volatile uint8_t *u = sm;          // [N]
    uint8_t *inf = sm + p.N;           // [K]
    const int lane = threadIdx.x;
    const long Bv = p.n_dev ? ((long)*p.n_dev < p.B ? (long)*p.n_dev : p.B) : p.B;
    for (long b = blockIdx.x; b < Bv; b += gridDim.x) {
        for (int i = lane; i < p.K; i += 64) inf[i] = p.info[(size_t)b * p.K + i];
        wave_sync();
        encode_in_lds(u, inf, p, lane);
        // coded[i] = u[bitrev[i]]: PolarCode.cpp:85-87
        for (int i = lane; i < p.N; i += 64) p.coded[(size_t)b * p.N + i] = u[__brev((unsigned)i) >> (32 - p.n)];
        wave_sync();
    }
}

__global__ __launch_bounds__(64) void synth_kernel(PolarEncodeParams p) {
    extern __shared__ uint8_t sm[];
    volatile uint8_t *u = sm;
    uint8_t *inf = sm + p.N;
    const int lane = threadIdx.x;
    const long Bv = p.n_dev ? ((long)*p.n_dev < p.B ? (long)*p.n_dev : p.B) : p.B;
    for (long b = blockIdx.x; b < Bv; b += gridDim.x) {
        const uint64_t trial = p.sel ? p.sel[b] : (p.trial0 + (uint64_t)b * (uint64_t)p.stride);
        const uint64_t block = trial / (uint64_t)p.info_block_div;   // 100: info refreshed every 100 runs, PolarCode.cpp:703-707
        for (int i = lane; i < p.K; i += 64) {
            uint32_t r[4];
            polar_synth_info_word(p.seed, block, (uint32_t)(i >> 7), r);
            int k = i & 127;
            inf[i] = (uint8_t)((r[(k >> 5) & 3] >> (k & 31)) & 1u);
        }
        wave_sync();
        if (p.info_out)
            for (int i = lane; i < p.K; i += 64) p.info_out[(size_t)b * p.K + i] = inf[i];
        encode_in_lds(u, inf, p, lane);
        if (p.coded)
            for (int i = lane; i < p.N; i += 64) p.coded[(size_t)b * p.N + i] = u[__brev((unsigned)i) >> (32 - p.n)];
        if (p.constellation != 0) {
            // ASK Gray + BICM demapper: Constellation.m:84-93, 123-144 (include/polar_synth.h)
            const int nb = polar_const_nbits(p.constellation);
            const int nsym = p.N / nb;
            double *dl = p.llr + (size_t)b * p.N;
            for (int i = nsym * nb + lane; i < p.N; i += 64) dl[i] = 0.0;
            for (int i = lane; i < nsym; i += 64) {
                int sym = 0;
                for (int j = 0; j < nb; ++j) sym += (1 << j) * (int)u[__brev((unsigned)(i * nb + j)) >> (32 - p.n)];
                const double x = polar_const_point(p.constellation, sym) / p.cnorm;
                const double y = x + polar_synth_symbol_noise(p.seed, trial, (uint32_t)i) * p.sigma;
                double l4[4];
                polar_synth_bicm_demap(p.constellation, p.cnorm, y, p.n0, l4);
                for (int j = 0; j < nb; ++j) dl[(size_t)i * nb + j] = l4[j];
            }
            wave_sync();
            continue;
        }
        double2 *dst = reinterpret_cast<double2 *>(p.llr + (size_t)b * p.N);
        for (int pr = lane; pr < p.N / 2; pr += 64) {
            double z0, z1;
            polar_synth_noise_pair(p.seed, trial, (uint32_t)pr, &z0, &z1);
            int c0 = u[__brev((unsigned)(2 * pr)) >> (32 - p.n)];
            int c1 = u[__brev((unsigned)(2 * pr + 1)) >> (32 - p.n)];
            double2 v;
            v.x = polar_synth_llr(p.s, c0, z0);
            v.y = polar_synth_llr(p.s, c1, z1);
            dst[pr] = v;
        }
        wave_sync();
    }
}

// one wave per codeword: any differing info bit => block error (PolarCode.cpp:758-769)
__global__ __launch_bounds__(64) void count_errors_kernel(const uint8_t *a, const uint8_t *b, long B, int K,
                                                           unsigned long long *err, uint8_t *flags) {
    const int lane = threadIdx.x;
    for (long c = blockIdx.x; c < B; c += gridDim.x) {
        bool diff = false;
        for (int i = lane; i < K; i += 64) diff |= (a[(size_t)c * K + i] != b[(size_t)c * K + i]);
        bool any = __any(diff);
        if (lane == 0) {
            if (any && err) atomicAdd(err, 1ull);
            if (flags) flags[c] = any ? 1 : 0;
        }
    }
}

// ---- Monte-Carlo round, device side (no host round trip between the (L, Eb/N0) points) ----
__global__ __launch_bounds__(256) void mc_init_alive_kernel(uint64_t *alive, unsigned *n, uint64_t t0, long stride, long T) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < T; i += (long)gridDim.x * 256) alive[i] = t0 + (uint64_t)i * (uint64_t)stride;
    if (blockIdx.x == 0 && threadIdx.x == 0) *n = (unsigned)T;
}
__global__ __launch_bounds__(64) void mc_count_compact_kernel(const uint8_t *dec, const uint8_t *sent, long B, int K,
                                                               const uint64_t *alive_in, const unsigned *n_in, uint64_t *alive_out,
                                                               unsigned *n_out, unsigned long long *ctr) {
    const int lane = threadIdx.x;
    if (n_in && (long)*n_in < B) B = (long)*n_in;          // (n_in == nullptr: exactly B rows)
    for (long c = blockIdx.x; c < B; c += gridDim.x) {
        unsigned nd = 0;
        for (int i = lane; i < K; i += 64) nd += (dec[(size_t)c * K + i] != sent[(size_t)c * K + i]) ? 1u : 0u;
        for (int off = 32; off >= 1; off >>= 1) nd += __shfl_xor(nd, off, 64);
        if (lane == 0 && nd) {
            atomicAdd(ctr, 1ull);
            atomicAdd(ctr + 1, (unsigned long long)nd);
            alive_out[atomicAdd(n_out, 1u)] = alive_in[c];
        }
    }
}

}  // namespace

static int grid_for(long B) { return (int)(B < 8192 ? (B > 0 ? B : 1) : 8192); }

hipError_t polar_launch_encode(const PolarEncodeParams &p, hipStream_t st) {
    hipLaunchKernelGGL(encode_kernel, dim3(grid_for(p.B)), dim3(64), (size_t)p.N + p.K, st, p);
    return hipGetLastError();
}
hipError_t polar_launch_synth(const PolarEncodeParams &p, hipStream_t st) {
    hipLaunchKernelGGL(synth_kernel, dim3(grid_for(p.B)), dim3(64), (size_t)p.N + p.K, st, p);
    return hipGetLastError();
}
hipError_t polar_launch_mc_init_alive(uint64_t *alive, unsigned *n, uint64_t t0, long stride, long T, hipStream_t st) {
    const long blocks = (T + 255) / 256;
    hipLaunchKernelGGL(mc_init_alive_kernel, dim3((unsigned)(blocks < 1024 ? (blocks ? blocks : 1) : 1024)), dim3(256), 0, st, alive, n, t0, stride, T);
    return hipGetLastError();
}
hipError_t polar_launch_mc_count_compact(const uint8_t *decoded, const uint8_t *sent, long B, int K,
                                         const uint64_t *alive_in, const unsigned *n_in, uint64_t *alive_out, unsigned *n_out,
                                         unsigned long long *ctr, hipStream_t st) {
    hipLaunchKernelGGL(mc_count_compact_kernel, dim3(grid_for(B)), dim3(64), 0, st, decoded, sent, B, K, alive_in, n_in, alive_out, n_out, ctr);
    return hipGetLastError();
}
hipError_t polar_launch_count_errors(const uint8_t *a, const uint8_t *b, long B, int K,
                                     unsigned long long *err, uint8_t *flags, hipStream_t st) {
    hipLaunchKernelGGL(count_errors_kernel, dim3(grid_for(B)), dim3(64), 0, st, a, b, B, K, err, flags);
    return hipGetLastError();
}
