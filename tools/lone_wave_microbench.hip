// tools/lone_wave_microbench.hip — what ONE wave pays per instruction when nothing else runs on its SIMD (the one-codeword-per-wave
// latency kernels): cycles (s_memtime, 100 MHz-independent: the shader clock counter s_memrealtime is NOT used) per
// taken branch, untaken branch, dependent / independent fp64 FMA, scalar ALU op, v_readlane, DPP move, dependent LDS read,
// LDS write -> read of another lane's value, ds_bpermute. Build and run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 tools/lone_wave_microbench.hip -o /tmp/lw && /tmp/lw
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)
#define REP256(x) REP64(x) REP64(x) REP64(x) REP64(x)

__device__ __forceinline__ uint64_t now() { return __builtin_readcyclecounter(); }   // s_memtime: shader-clock cycles

__global__ void bench(uint64_t *out, double *sink) {
    __shared__ double lds[1024];
    __shared__ int chain[1024];
    const int lane = threadIdx.x;
    for (int i = lane; i < 1024; i += 64) { lds[i] = 1.0 + i; chain[i] = ((i + 64) & 1023) * 4; }
    __syncthreads();
    uint64_t t0, t1;
    int k = 0;
    // 0: empty timing pair
    t0 = now(); t1 = now(); out[k++] = t1 - t0;
    // 1: 256 s_nop 0
    t0 = now(); asm volatile(REP256("s_nop 0\n")); t1 = now(); out[k++] = t1 - t0;
    // 2: 256 taken branches (to the next instruction)
    t0 = now(); asm volatile(REP256("s_branch 0\n")); t1 = now(); out[k++] = t1 - t0;
    // 3: 256 untaken conditional branches (scc = 0)
    t0 = now(); asm volatile("s_cmp_eq_u32 0, 1\n" REP256("s_cbranch_scc1 0\n")); t1 = now(); out[k++] = t1 - t0;
    // 4: 256 dependent fp64 FMAs
    double a = sink[0], b = sink[1];
    t0 = now(); asm volatile(REP256("v_fma_f64 %0, %0, %1, %1\n") : "+v"(a) : "v"(b)); t1 = now(); out[k++] = t1 - t0;
    // 5: 256 fp64 FMAs, four independent chains
    double c0 = a, c1 = a + 1, c2 = a + 2, c3 = a + 3;
    t0 = now();
    asm volatile(REP64("v_fma_f64 %0, %0, %4, %4\nv_fma_f64 %1, %1, %4, %4\nv_fma_f64 %2, %2, %4, %4\nv_fma_f64 %3, %3, %4, %4\n")
                 : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(b));
    t1 = now(); out[k++] = t1 - t0;
    // 6: 256 dependent scalar adds
    int s = 1;
    t0 = now(); asm volatile(REP256("s_add_u32 %0, %0, 1\n") : "+s"(s)); t1 = now(); out[k++] = t1 - t0;
    // 7: 256 v_readlane (independent)
    int r = lane, sr;
    t0 = now(); asm volatile(REP256("v_readlane_b32 %0, %1, 3\n") : "=s"(sr) : "v"(r)); t1 = now(); out[k++] = t1 - t0;
    // 8: 256 dependent DPP moves (quad_perm)
    int d = lane;
    t0 = now(); asm volatile(REP256("s_nop 1\nv_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n") : "+v"(d)); t1 = now(); out[k++] = t1 - t0;
    // 9: 64 dependent LDS reads (pointer chase, b32)
    int pch = lane * 4;
    t0 = now(); asm volatile(REP64("ds_read_b32 %0, %0\ns_waitcnt lgkmcnt(0)\n") : "+v"(pch)); t1 = now(); out[k++] = t1 - t0;
    // 10: 64 x (LDS write b64, read b64 of the neighbour lane's slot, wait)
    double w = a;
    int addr = lane * 8, addr2 = ((lane + 1) & 63) * 8;
    t0 = now();
    asm volatile(REP64("ds_write_b64 %1, %0 offset:4096\nds_read_b64 %0, %2 offset:4096\ns_waitcnt lgkmcnt(0)\n") : "+v"(w) : "v"(addr), "v"(addr2));
    t1 = now(); out[k++] = t1 - t0;
    // 11: 64 dependent ds_bpermute
    int bp = lane, bidx = ((lane + 1) & 63) * 4;
    t0 = now(); asm volatile(REP64("ds_bpermute_b32 %0, %1, %0\ns_waitcnt lgkmcnt(0)\n") : "+v"(bp) : "v"(bidx)); t1 = now(); out[k++] = t1 - t0;
    // 12: 64 x v_rcp_f64 dependent
    double rc = a;
    t0 = now(); asm volatile(REP64("v_rcp_f64 %0, %0\n") : "+v"(rc)); t1 = now(); out[k++] = t1 - t0;
    // 13: 256 x s_mov_b32 literal pairs + v_fma using them (constant materialisation next to the use)
    double lc = a;
    t0 = now(); asm volatile(REP64("s_mov_b32 s20, 0x55555555\ns_mov_b32 s21, 0x3fd55555\nv_fma_f64 %0, %0, s[20:21], %0\n") : "+v"(lc) : : "s20", "s21"); t1 = now(); out[k++] = t1 - t0;
    // 14: 64 taken BACKWARD-free far branches: branch over 64 dwords of nops (a new cache line every time)
    t0 = now(); asm volatile(REP64("s_branch 32\n" REP16("s_nop 0\ns_nop 0\n"))); t1 = now(); out[k++] = t1 - t0;
    // 15: 64 x ds_read_b64 x2 issued together + wait (one round trip of a layer visit)
    double q0 = 0, q1 = 0;
    t0 = now(); asm volatile(REP64("ds_read_b64 %0, %2\nds_read_b64 %1, %2 offset:512\ns_waitcnt lgkmcnt(0)\n") : "=v"(q0), "=v"(q1) : "v"(addr)); t1 = now(); out[k++] = t1 - t0;
    if (lane == 0) sink[2] = a + c0 + c1 + c2 + c3 + s + sr + d + pch + w + bp + rc + lc + q0 + q1;
}

int main() {
    uint64_t *d_out; double *d_sink;
    hipMalloc(&d_out, 32 * 8); hipMalloc(&d_sink, 64);
    double h[8] = {1.0000001, 0.9999999, 0, 0, 0, 0, 0, 0};
    hipMemcpy(d_sink, h, 64, hipMemcpyHostToDevice);
    const char *names[] = {"empty timing pair", "256 s_nop 0", "256 taken s_branch (next instruction)", "256 untaken s_cbranch", "256 dependent v_fma_f64",
                           "256 v_fma_f64 in 4 chains", "256 dependent s_add_u32", "256 v_readlane_b32", "256 dependent v_mov_dpp (+ s_nop 1)",
                           "64 dependent ds_read_b32", "64 x (ds_write_b64, ds_read_b64, wait)", "64 dependent ds_bpermute_b32", "64 dependent v_rcp_f64",
                           "64 x (2 s_mov literal + v_fma_f64)", "64 taken s_branch over 128 B", "64 x (2 ds_read_b64 + wait)"};
    const int cnt[] = {1, 256, 256, 256, 256, 256, 256, 256, 256, 64, 64, 64, 64, 64, 64, 64};
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(bench, dim3(1), dim3(64), 0, 0, d_out, d_sink);
        hipDeviceSynchronize();
    }
    uint64_t o[32];
    hipMemcpy(o, d_out, sizeof o, hipMemcpyDeviceToHost);
    for (int i = 0; i < 16; ++i) printf("%-44s %8llu cycles  %7.1f per op\n", names[i], (unsigned long long)o[i], (double)(o[i] - o[0]) / cnt[i]);
    return 0;
}
