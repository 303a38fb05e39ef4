// tools/mall_microbench.hip — what do the L2 / Infinity Cache give the decoder's access pattern? 4096 persistent waves (16 per CU),
// each walking its OWN region in 512-B rows (8 B per lane, as the decoder's per-wave scratch): a pass writes the region, the next
// pass reads it back and rewrites it. Total footprint = 4096 x region: 16 MiB (inside the 32 MiB of L2) ... 4 GiB (the decoder's).
// FETCH_SIZE / WRITE_SIZE count requests on the fabric side of the L2, Infinity-Cache hits included, so the split between the
// Infinity Cache and DRAM cannot be read from counters (TCC_EA0_RDREQ_DRAM_sum == TCC_EA0_RDREQ_sum on this part: "DRAM" there
// means "not GMI / IO"); this curve shows what bandwidth a working set of a given reuse footprint gets.
// build+run on the GPU box: hipcc --offload-arch=gfx950 -O3 tools/mall_microbench.hip -o /tmp/mb && /tmp/mb
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ __launch_bounds__(256, 4) void walk(double *buf, size_t region_doubles, int passes, int nt) {
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    double *r = buf + (size_t)wave * region_doubles;
    const size_t rows = region_doubles / 64;
    double acc = 0.0;
    for (int p = 0; p < passes; ++p) {
        for (size_t row = 0; row < rows; row += 16) {
            double v[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) v[k] = nt ? __builtin_nontemporal_load(r + (row + k) * 64 + lane) : r[(row + k) * 64 + lane];
#pragma unroll
            for (int k = 0; k < 16; ++k) acc += v[k];
#pragma unroll
            for (int k = 0; k < 16; ++k) { if (nt) __builtin_nontemporal_store(acc + k, r + (row + k) * 64 + lane); else r[(row + k) * 64 + lane] = acc + k; }
        }
    }
    if (acc == 12345.678) buf[0] = acc;
}

int main() {
    const int waves = 4096;
    const size_t max_region = 1 << 17;                 // doubles per wave (1 MiB)
    double *buf;
    if (hipMalloc(&buf, (size_t)waves * max_region * 8) != hipSuccess) return 1;
    (void)hipMemset(buf, 0, (size_t)waves * max_region * 8);
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (int nt = 0; nt <= 1; ++nt)
        for (size_t region = 512; region <= max_region; region *= 2) {          // 4 KiB ... 1 MiB per wave
            const int passes = (int)((size_t)(1 << 24) / region);               // same bytes moved per launch (128 Mi doubles per wave-pass sum)
            for (int rep = 0; rep < 2; ++rep) {
                (void)hipEventRecord(a);
                hipLaunchKernelGGL(walk, dim3(waves / 4), dim3(256), 0, 0, buf, region, passes, nt);
                (void)hipEventRecord(b);
                (void)hipEventSynchronize(b);
                float ms;
                (void)hipEventElapsedTime(&ms, a, b);
                const double bytes = (double)waves * passes * region * 8 * 2;    // read + write
                if (rep) printf("%s region %7zu KiB/wave  footprint %8.1f MiB: %8.2f ms  %6.2f TB/s (read + write at the L1 level)\n", nt ? "nt   " : "plain",
                                region * 8 / 1024, (double)waves * region * 8 / 1048576.0, ms, bytes / ms / 1e9);
            }
        }
    return 0;
}
