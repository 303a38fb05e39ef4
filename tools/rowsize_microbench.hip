// tools/rowsize_microbench.hip — does the DRAM side prefer 1-KiB rows (dwordx4 per lane) over the
// decoder's 512-B rows (dwordx2 per lane)? 4096 persistent waves, each streaming through its own 1-MiB
// region (like the decoder's per-wave scratch): read 16 rows / write 8 rows per step.
// build+run on the GPU box: hipcc --offload-arch=gfx950 -O3 tools/rowsize_microbench.hip -o /tmp/rb && /tmp/rb
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int W>   // W = doubles per lane per access (1: 512-B rows, 2: 1-KiB rows)
__global__ __launch_bounds__(256, 4) void stream(double *buf, size_t region_doubles, int iters) {
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    double *r = buf + (size_t)wave * region_doubles;
    const size_t rows = region_doubles / (64 * W);          // rows of 64*W doubles
    double acc = 0.0;
    for (int it = 0; it < iters; ++it) {
        const size_t base = ((size_t)it * 24 / W) % (rows - 32);
        double v[16];
#pragma unroll
        for (int k = 0; k < 16 / W; ++k) {
            const double *p = r + (base + k) * 64 * W + lane * W;
            if (W == 1) v[k] = p[0];
            else { double2 t = *reinterpret_cast<const double2 *>(p); v[2 * k] = t.x; v[2 * k + 1] = t.y; }
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) acc += v[k];
#pragma unroll
        for (int k = 0; k < 8 / W; ++k) {
            double *q = r + (base + 16 / W + k) * 64 * W + lane * W;
            if (W == 1) q[0] = acc + k;
            else *reinterpret_cast<double2 *>(q) = make_double2(acc + k, acc - k);
        }
    }
    if (acc == 12345.678) buf[0] = acc;
}

int main() {
    const size_t region = 1 << 17;                 // doubles per wave (1 MiB)
    const int waves = 4096, iters = 4000;
    double *buf;
    hipMalloc(&buf, (size_t)waves * region * 8);
    hipMemset(buf, 0, (size_t)waves * region * 8);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int rep = 0; rep < 2; ++rep)
        for (int W = 1; W <= 2; ++W) {
            hipEventRecord(a);
            if (W == 1) hipLaunchKernelGGL(stream<1>, dim3(waves / 4), dim3(256), 0, 0, buf, region, iters);
            else hipLaunchKernelGGL(stream<2>, dim3(waves / 4), dim3(256), 0, 0, buf, region, iters);
            hipEventRecord(b);
            hipEventSynchronize(b);
            float ms;
            hipEventElapsedTime(&ms, a, b);
            const double bytes = (double)waves * iters * 24 * 512;
            printf("row %4d B: %.2f ms, %.2f TB/s (L1-level bytes: 2/3 read, 1/3 write)\n", 512 * W, ms, bytes / ms / 1e9);
        }
    return 0;
}
