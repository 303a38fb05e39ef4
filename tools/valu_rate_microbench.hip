// tools/valu_rate_microbench.hip — issue cost of the fp64 VALU instructions the node arithmetic is built
// from, relative to v_fma_f64, and the raw accuracy of v_rcp_f64 (how many Newton steps a division needs).
// hipcc --offload-arch=gfx950 -O3 -o valu_rate tools/valu_rate_microbench.hip && ./valu_rate
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>

#define ITERS 2048
template <int OP>
__global__ __launch_bounds__(256) void k(double *out, double seed) {
    double x[8];
    for (int i = 0; i < 8; ++i) x[i] = seed + 0.001 * (threadIdx.x + i);
    float f[8];
    for (int i = 0; i < 8; ++i) f[i] = (float)x[i];
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (OP == 0) x[i] = __builtin_fma(x[i], 1.0000001, 1e-9);
            if (OP == 1) x[i] = __builtin_amdgcn_rcp(x[i]);
            if (OP == 2) f[i] = __builtin_amdgcn_rcpf(f[i]);
            if (OP == 3) { f[i] = (float)x[i]; x[i] = x[i] + 1.0; }          // cvt_f32_f64 + add
            if (OP == 4) x[i] = x[i] + 1.0;                                    // add only
            if (OP == 5) x[i] = x[i] * 1.0000001;
            if (OP == 6) x[i] = __builtin_fmax(x[i], 1.5) ;
            if (OP == 7) x[i] = __builtin_ldexp(x[i], 1) ;
            if (OP == 8) x[i] = __builtin_rint(x[i] * 1.0000001);
            if (OP == 9) x[i] = (x[i] > 1.25) ? x[i] - 0.5 : x[i] + 0.25;      // cmp + 2 cndmask + 2 add
            if (OP == 10) x[i] = (double)f[i] + x[i];                           // cvt_f64_f32 + add
            if (OP == 11) x[i] = __builtin_amdgcn_rsq(x[i]);
            if (OP == 12) x[i] = __builtin_sqrt(x[i]);
            if (OP == 13) { int e; x[i] = __builtin_frexp(x[i], &e) + 0.75; }
            if (OP == 14) x[i] = __hiloint2double(__double2hiint(x[i]) ^ 0x100, __double2loint(x[i]));   // 1 int op
        }
    }
    double s = 0;
    for (int i = 0; i < 8; ++i) s += x[i] + f[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ void acc(double *o) {
    // raw accuracy of v_rcp_f64 on [1,2) and after one / two Newton steps
    double worst0 = 0, worst1 = 0, worst2 = 0;
    for (int i = 0; i < 100000; ++i) {
        double d = 1.0 + (double)(i * 64 + threadIdx.x) / 6400000.0;
        double r = __builtin_amdgcn_rcp(d);
        double e = __builtin_fma(-d, r, 1.0);
        worst0 = fmax(worst0, fabs(e));
        r = __builtin_fma(r, e, r);
        e = __builtin_fma(-d, r, 1.0);
        worst1 = fmax(worst1, fabs(e));
        r = __builtin_fma(r, e, r);
        e = __builtin_fma(-d, r, 1.0);
        worst2 = fmax(worst2, fabs(e));
    }
    o[threadIdx.x * 3] = worst0; o[threadIdx.x * 3 + 1] = worst1; o[threadIdx.x * 3 + 2] = worst2;
}
template <int OP> double run(double *d, const char *name, double base) {
    const int blocks = 256 * 4;   // 4 blocks of 4 waves per CU: 4 waves per SIMD
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 1.5);
    hipEventRecord(a);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 1.5);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    // per SIMD: 4 waves * ITERS * 8 instructions (of the op under test)
    double per = ms * 1e-3 / (4.0 * ITERS * 8);
    printf("%-28s %8.3f ms  %7.2f ns per wave-instruction-group  rel %.2f\n", name, ms, per * 1e9, base > 0 ? per / base : 1.0);
    return per;
}
int main() {
    double *d; hipMalloc(&d, 256 * 4 * 256 * 8);
    double b = run<0>(d, "v_fma_f64", 0);
    run<4>(d, "v_add_f64", b); run<5>(d, "v_mul_f64", b); run<1>(d, "v_rcp_f64", b); run<2>(d, "v_rcp_f32", b);
    run<3>(d, "cvt_f32_f64 + add", b); run<10>(d, "cvt_f64_f32 + add", b); run<6>(d, "v_max_f64", b); run<7>(d, "v_ldexp_f64", b);
    run<8>(d, "mul + v_rndne_f64", b); run<9>(d, "cmp+2cndmask+2add", b); run<11>(d, "v_rsq_f64", b); run<12>(d, "sqrt (library)", b);
    run<13>(d, "frexp_mant + add", b); run<14>(d, "v_xor_b32", b);
    hipLaunchKernelGGL(acc, dim3(1), dim3(64), 0, 0, d);
    std::vector<double> h(192); hipMemcpy(h.data(), d, 192 * 8, hipMemcpyDeviceToHost);
    double w0 = 0, w1 = 0, w2 = 0; for (int i = 0; i < 64; ++i) { w0 = fmax(w0, h[3*i]); w1 = fmax(w1, h[3*i+1]); w2 = fmax(w2, h[3*i+2]); }
    printf("v_rcp_f64 relative error: raw %.3g (2^%.1f), after 1 Newton step %.3g, after 2 %.3g\n", w0, log2(w0), w1, w2);
    return 0;
}
