// tools/traffic_replay.hip — the headline kernel's HBM-side access pattern WITHOUT its arithmetic: what bandwidth does the memory
// system give this pattern? (the saturation record on the kernel's real touched footprint and read : write
// mix, not on a 4-GiB stand-in.)
//
// 4096 persistent waves (16 per CU, four-wave blocks, as the kernel), each with the kernel's per-wave scratch layout — layer of
// size S at row (S - 16) of a (N - 16)-row region, 512-B rows — replay the layer visits of ONE list-of-32 decode of N = 2048 in
// table mode, leaf by leaf in steps of 16 leaves (the layers of size <= 8 live in LDS and move no HBM bytes):
//   * chain at leaf phi: g-visit of the layer of size S = lowbit(phi) reading 2 S rows of the layer above (from the per-codeword
//     layer-2 value table when that layer is a table: 8 B per lane out of a 64-B line per element), then the f-visits below it,
//     four layers per pass: 16 row loads in flight, then 8 + 4 + 2 + 1 row stores; the lowest output of a pass is re-read by the
//     next pass; layers of size >= 64 non-temporal, 16 and 32 cacheable (the kernel's hints);
//   * phi = N/4, N/2, 3N/4: the table builds (channel rows / prefix rows in, X[N/2][2] and T2[N/4][2 / 4 / 8] out, per codeword);
//   * the layers of size 512 and 1024 are never touched (table mode): the touched footprint is what the kernel's is.
// Partial-sum words (1 bit per element against 64) and the decision history are left out (< 3 % of the bytes).
// Every "value" is a sum of what was loaded, so nothing is optimised away; no fp64 node arithmetic, no LDS, no fork / prune.
// Output: bytes per wave-decode and in total, time, TB/s — to be put next to the kernel's own 1.29 TB per launch at 5.15 TB/s.
// build+run on the GPU box: hipcc --offload-arch=gfx950 -O3 tools/traffic_replay.hip -o /tmp/tr && /tmp/tr [rounds]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

constexpr int N = 2048, SLMAX = 16;          // smallest HBM-resident layer
constexpr int TOPS = N / 8;                  // largest per-path layer in table mode (256)

__device__ __forceinline__ double ld(const double *p, bool nt) { return nt ? __builtin_nontemporal_load(p) : *p; }
__device__ __forceinline__ void st(double *p, double v, bool nt) { if (nt) __builtin_nontemporal_store(v, p); else *p = v; }

// one four-layer pass: source rows src[0 .. 2S) (or the table), outputs S, S/2, S/4, S/8 rows (those >= 16 go to HBM)
__device__ __forceinline__ double pass4(double *reg, const double *tab, int S, bool from_table, int lane, double acc) {
    const int E = S / 8 > 0 ? S / 8 : 1;
    const double *src = reg + (size_t)(2 * S - SLMAX) * 64 + lane;
    const bool nt_in = 2 * S >= 64;
    for (int j = 0; j < E; ++j) {
        double a[8], b[8];
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            const int e0 = j + m * E, e1 = e0 + S;
            if (from_table) {          // T2c[8 e + variant]: the 32 lanes of a codeword inside one 64-B line
                a[m] = tab[(size_t)(lane >> 5) * 3 * N + N + 8 * (size_t)e0 + ((lane * 5 + e0) & 7)];
                b[m] = tab[(size_t)(lane >> 5) * 3 * N + N + 8 * (size_t)(e1 & (N / 4 - 1)) + ((lane * 3 + e1) & 7)];
            } else {
                a[m] = ld(src + (size_t)e0 * 64, nt_in);
                b[m] = ld(src + (size_t)e1 * 64, nt_in);
            }
        }
        double v[8];
#pragma unroll
        for (int m = 0; m < 8; ++m) v[m] = a[m] + b[m] + acc;
        int T = S;
#pragma unroll
        for (int lvl = 0; lvl < 4; ++lvl) {
            const int cnt = 8 >> lvl;                 // rows this level stores for this j
            if (T >= SLMAX) {
                double *o = reg + (size_t)(T - SLMAX) * 64 + lane;
#pragma unroll
                for (int m = 0; m < 8; ++m) if (m < cnt) st(o + (size_t)(j + m * E) * 64, v[m], T >= 64);
            }
#pragma unroll
            for (int m = 0; m < 4; ++m) if (m < cnt / 2) v[m] += v[m + cnt / 2];
            T >>= 1;
        }
        acc = v[0];
    }
    return acc;
}

__global__ __launch_bounds__(256, 4) void replay(double *scr, double *tabs, const double *chan, const double *pre, int rounds, int phi0) {
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    double *reg = scr + (size_t)wave * (size_t)(N - SLMAX) * 64;
    double *tab = tabs + (size_t)wave * 2 * 3 * N;                      // two codewords per wave: X[N/2][2], T2[N/4][8]
    double acc = 0.0;
    for (int r = 0; r < rounds; ++r) {
        const double *ch = chan + ((size_t)(r * gridDim.x * 4 + wave) * 2 + (lane >> 5)) * N;      // this round's two codewords
        const double *pr = pre + ((size_t)(r * gridDim.x * 4 + wave) * 2 + (lane >> 5)) * (N - 256 + 1);
        for (int phi = phi0; phi < N; phi += SLMAX) {
            int S = phi & -phi;
            if (S >= N / 4) {
                // table build: 32 lanes per codeword share the elements
                const int lig = lane & 31;
                double *Xc = tab + (size_t)(lane >> 5) * 3 * N, *T2c = Xc + N;
                if (phi == N / 4) {
                    for (int j = lig; j < N / 4; j += 32) { const double a = pr[1 + j], b = pr[1 + j + N / 4]; T2c[8 * j] = a + b + acc; T2c[8 * j + 1] = a - b; }
                } else {
                    if (phi == N / 2)
                        for (int e = lig; e < N / 2; e += 32) { const double a = ch[2 * e], b = ch[2 * e + 1]; Xc[2 * e] = a + b + acc; Xc[2 * e + 1] = a - b; }
                    const int nv = phi == N / 2 ? 4 : 8;
                    for (int j = lig; j < N / 4; j += 32) {
                        const double a0 = Xc[2 * j], a1 = Xc[2 * j + 1], b0 = Xc[2 * (j + N / 4)], b1 = Xc[2 * (j + N / 4) + 1];
                        for (int v = 0; v < nv; ++v) T2c[8 * j + v] = ((v & 1) ? a1 : a0) + ((v & 2) ? b1 : b0) + v;
                    }
                }
                S = TOPS;                                            // the chain goes on at layer 3 with the table as its source
                acc = pass4(reg, tab, S, true, lane, acc);
            } else {
                acc = pass4(reg, tab, S, false, lane, acc);          // g-visit of layer S (source: the layer of size 2 S) + three f-layers
            }
            // the f-chain below, four layers per pass, while its layers are HBM-resident (S/8 >= 16 was written by the pass above:
            // the next pass re-reads it as the source of the layer of size S/16)
            for (int T = S / 16; T >= SLMAX / 2; T /= 16) {
                if (2 * T < SLMAX * 2 && T < SLMAX) {                 // source (size 2T = 16) is read, outputs live in LDS
                    const double *src = reg + (size_t)(2 * T - SLMAX) * 64 + lane;
                    for (int j = 0; j < 2 * T; ++j) acc += src[(size_t)j * 64];
                    break;
                }
                acc = pass4(reg, tab, T, false, lane, acc);
            }
        }
    }
    if (acc == 12345.678) scr[0] = acc;
}

int main(int argc, char **argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 8;
    const int waves = 4096;
    double *scr, *tabs, *chan, *pre;
    const size_t scr_b = (size_t)waves * (N - SLMAX) * 64 * 8, tab_b = (size_t)waves * 2 * 3 * N * 8;
    const size_t cw = (size_t)waves * 2 * rounds, chan_b = cw * N * 8, pre_b = cw * (N - 256 + 1) * 8;
    if (hipMalloc(&scr, scr_b) != hipSuccess || hipMalloc(&tabs, tab_b) != hipSuccess || hipMalloc(&chan, chan_b) != hipSuccess || hipMalloc(&pre, pre_b) != hipSuccess) return 1;
    (void)hipMemset(scr, 0, scr_b); (void)hipMemset(tabs, 0, tab_b); (void)hipMemset(chan, 0, chan_b); (void)hipMemset(pre, 0, pre_b);
    // bytes of one wave-decode under this model (host replica of the loops above)
    double rd = 0, wr = 0;
    auto pass = [&](int S, bool tbl) {
        const int E = S / 8 > 0 ? S / 8 : 1;
        rd += tbl ? (double)E * 16 * 2 * 64 : (double)E * 16 * 512;      // table: 64-B line per element and codeword
        int T = S;
        for (int l = 0; l < 4; ++l) { if (T >= SLMAX) wr += (double)E * (8 >> l) * 512; T >>= 1; }
    };
    const int phi0 = 256;                                                  // (the all-frozen prefix is the prefix kernel's)
    for (int phi = phi0; phi < N; phi += SLMAX) {
        int S = phi & -phi;
        if (S >= N / 4) {
            if (phi == N / 4) { rd += 2 * (N / 2) * 8.0; wr += 2 * (N / 4) * 2 * 8.0; }
            else { if (phi == N / 2) { rd += 2 * N * 8.0; wr += 2 * N * 8.0; } rd += 2 * N * 8.0; wr += 2 * (N / 4) * (phi == N / 2 ? 4 : 8) * 8.0; }
            S = TOPS; pass(S, true);
        } else pass(S, false);
        for (int T = S / 16; T >= SLMAX / 2; T /= 16) {
            if (T < SLMAX) { rd += 2.0 * T * 512; break; }
            pass(T, false);
        }
    }
    printf("model: %.3f MB read + %.3f MB written per wave-decode (two codewords) = %.3f MB per codeword; read : write = %.2f : %.2f\n",
           rd / 1e6, wr / 1e6, (rd + wr) / 2e6, rd / (rd + wr), wr / (rd + wr));
    printf("touched footprint: scratch layers 16..256 %.1f MiB + tables %.1f MiB (of %.1f MiB allocated scratch)\n",
           (double)waves * (2 * TOPS - SLMAX) * 512 / 1048576.0, tab_b / 1048576.0, scr_b / 1048576.0);
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(a);
        hipLaunchKernelGGL(replay, dim3(waves / 4), dim3(256), 0, 0, scr, tabs, chan, pre, rounds, phi0);
        (void)hipEventRecord(b);
        (void)hipEventSynchronize(b);
        float ms;
        (void)hipEventElapsedTime(&ms, a, b);
        const double bytes = (rd + wr) * waves * rounds;
        printf("replay of %d rounds (%zu codewords): %.2f ms, %.1f GB moved, %.2f TB/s; per round %.2f ms (kernel: 7.8 ms per round at 1.29 TB per 32 rounds = 5.15 TB/s)\n",
               rounds, cw, ms, bytes / 1e9, bytes / ms / 1e9, ms / rounds);
    }
    return 0;
}
