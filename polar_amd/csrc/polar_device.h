// polar_device.h — device-side helpers shared by the decode kernels (gfx950).
#pragma once
// Instrumented builds (per-phase cycle counters, margin / slot histograms, the list-size-1 latency profile) write their counters
// where a product build writes decoded bits or scratch: they measure, they do not decode. polar_amd/build.py gives them a library
// name of their own (POLAR_BUILD_TAG or --profile) and defines POLAR_DEV_BUILD; anything else that sets one of these macros — a
// stray POLAR_DEFS — stops here instead of shipping a decoder that is silently not bit-exact.
#if (defined(POLAR_PROFILE) || defined(POLAR_MARGIN) || defined(POLAR_SLOTHIST) || defined(SCLAT_PROF)) && !defined(POLAR_DEV_BUILD)
#error "measurement macro without POLAR_DEV_BUILD: build through polar_amd/build.py with POLAR_BUILD_TAG (or --profile)"
#endif
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {

typedef unsigned long long u64;

// 16 one-byte slot pointers packed in two 64-bit registers, indexed by a wave-uniform i
struct P16 {
    u64 lo, hi;
    __device__ __forceinline__ int get(int i) const {
        return i < 8 ? (int)((lo >> (8 * i)) & 0xFFull) : (int)((hi >> (8 * (i - 8))) & 0xFFull);
    }
    __device__ __forceinline__ void set(int i, int v) {
        if (i < 8) lo = (lo & ~(0xFFull << (8 * i))) | ((u64)(unsigned)v << (8 * i));
        else hi = (hi & ~(0xFFull << (8 * (i - 8)))) | ((u64)(unsigned)v << (8 * (i - 8)));
    }
};

__device__ __forceinline__ double shfl_d(double v, int src) { return __shfl(v, src, 64); }
__device__ __forceinline__ double readlane_d(double v, int src) {      // src wave-uniform: result lives in SGPRs
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src), hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}
// Lane index recomputed on the spot (2 VALU instructions). Everything derived from the lane index is
// loop-invariant, and with ~190 live values the register allocator parks such invariants in SCRATCH
// and reloads them in the middle of every leaf step / layer visit (a memory round trip each, with
// an s_waitcnt vmcnt on the critical path). A volatile asm cannot be hoisted or merged, so values
// derived from this are short-lived and never spilled.
__device__ __forceinline__ int lane_id_opaque() {
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}
__device__ __forceinline__ u64 shfl_u64(u64 v, int src) {
    unsigned lo = (unsigned)v, hi = (unsigned)(v >> 32);
    lo = __shfl(lo, src, 64);
    hi = __shfl(hi, src, 64);
    return ((u64)hi << 32) | lo;
}
// cross-lane max/min over the GS lanes of a group: DPP inside a row of 16 (quad xor1, xor2,
// half-mirror, mirror), ds_bpermute across rows
template <int CTRL>
__device__ __forceinline__ double dpp_d(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xF, 0xF, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
template <int GS, bool MAX>
__device__ __forceinline__ double group_reduce(double v, int lane) {
    auto op = [](double a, double b) { return MAX ? ((b > a) ? b : a) : ((b < a) ? b : a); };
    if (GS >= 2) v = op(v, dpp_d<0xB1>(v));      // quad_perm [1,0,3,2]
    if (GS >= 4) v = op(v, dpp_d<0x4E>(v));      // quad_perm [2,3,0,1]
    if (GS >= 8) v = op(v, dpp_d<0x141>(v));     // row_half_mirror
    if (GS >= 16) v = op(v, dpp_d<0x140>(v));    // row_mirror
    if (GS >= 32) v = op(v, __shfl(v, lane ^ 16, 64));
    if (GS >= 64) v = op(v, __shfl(v, lane ^ 32, 64));
    return v;
}
// 32-bit unsigned max/min over each group of GS lanes with DPP only (no LDS): the butterfly stages
// inside a row of 16 leave the row's result in every lane; for GS = 32 / 64 the row_bcast15 / row_bcast31
// controls (gfx9) carry it into the following rows, so the GROUP result is valid in the last row of
// each group only: lanes selected by group_result_rows<GS>().
template <int CTRL, int ROWMASK>
__device__ __forceinline__ unsigned dpp_u32(unsigned v) {
    return (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, ROWMASK, 0xF, false);
}
template <int GS, bool MAX>
__device__ __forceinline__ unsigned group_reduce_u32(unsigned v) {
    auto op = [](unsigned a, unsigned b) { return MAX ? ((b > a) ? b : a) : ((b < a) ? b : a); };
    if (GS >= 2) v = op(v, dpp_u32<0xB1, 0xF>(v));
    if (GS >= 4) v = op(v, dpp_u32<0x4E, 0xF>(v));
    if (GS >= 8) v = op(v, dpp_u32<0x141, 0xF>(v));
    if (GS >= 16) v = op(v, dpp_u32<0x140, 0xF>(v));
    if (GS >= 32) v = op(v, dpp_u32<0x142, 0xA>(v));     // row_bcast15: rows 1, 3 <- lane 15 of rows 0, 2
    if (GS >= 64) v = op(v, dpp_u32<0x143, 0xC>(v));     // row_bcast31: rows 2, 3 <- lane 31
    return v;
}
// max of `a` and min of `b` over each group at once, as DPP-FUSED VOP2 instructions (the compiler expands the
// update_dpp builtin above into mov + mov_dpp + max: three instructions per stage). The two chains are
// interleaved so that each covers one of the other's two DPP wait states (VALU write -> DPP read); same result
// placement as group_reduce_u32 (group_result_rows).
template <int GS>
__device__ __forceinline__ void group_max_min_u32(unsigned &a, unsigned &b) {
// (every stage carries BOTH wait states of its first DPP read inside its own asm statement: the compiler's hazard
// recognizer does not look into inline asm and may move the VALU producer of `a` right in front of it; the second
// instruction of a stage is covered by the first one)
#define POLAR_DPP_STAGE(CTRL)                                           \
    asm volatile("s_nop 1\n\t"                                         \
                 "v_max_u32_dpp %0, %0, %0 " CTRL "\n\t"                 \
                 "v_min_u32_dpp %1, %1, %1 " CTRL : "+v"(a), "+v"(b));
    if (GS >= 2) POLAR_DPP_STAGE("quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
    if (GS >= 4) POLAR_DPP_STAGE("quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf")
    if (GS >= 8) POLAR_DPP_STAGE("row_half_mirror row_mask:0xf bank_mask:0xf")
    if (GS >= 16) POLAR_DPP_STAGE("row_mirror row_mask:0xf bank_mask:0xf")
    if (GS >= 32) POLAR_DPP_STAGE("row_bcast:15 row_mask:0xa bank_mask:0xf")
    if (GS >= 64) POLAR_DPP_STAGE("row_bcast:31 row_mask:0xc bank_mask:0xf")
#undef POLAR_DPP_STAGE
}
// max alone (one chain: two wait states between a VALU write and the DPP read of it)
template <int GS>
__device__ __forceinline__ void group_max_u32(unsigned &a) {
#define POLAR_DPP_STAGE(CTRL) asm volatile("s_nop 1\n\tv_max_u32_dpp %0, %0, %0 " CTRL : "+v"(a));
    if (GS >= 2) POLAR_DPP_STAGE("quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
    if (GS >= 4) POLAR_DPP_STAGE("quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf")
    if (GS >= 8) POLAR_DPP_STAGE("row_half_mirror row_mask:0xf bank_mask:0xf")
    if (GS >= 16) POLAR_DPP_STAGE("row_mirror row_mask:0xf bank_mask:0xf")
    if (GS >= 32) POLAR_DPP_STAGE("row_bcast:15 row_mask:0xa bank_mask:0xf")
    if (GS >= 64) POLAR_DPP_STAGE("row_bcast:31 row_mask:0xc bank_mask:0xf")
#undef POLAR_DPP_STAGE
    asm volatile("s_nop 0");
}
template <int GS>
__device__ __forceinline__ constexpr u64 group_result_rows() {
    return GS <= 16 ? ~0ull : (GS == 32 ? 0xFFFF0000FFFF0000ull : 0xFFFF000000000000ull);
}
// wave votes as ONE scalar compare of the ballot mask (HIP's __any/__all materialise a 0/1 VGPR first)
__device__ __forceinline__ bool wave_any(bool c) { return __builtin_amdgcn_ballot_w64(c) != 0; }
__device__ __forceinline__ bool wave_all(bool c) { return __builtin_amdgcn_ballot_w64(!c) == 0; }   // over the active lanes
__device__ __forceinline__ void wave_mem_fence() {
    // lanes of one wave exchange data through LDS/global: keep the compiler from caching or
    // reordering across this point (hardware executes a wave's memory ops in order)
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

}  // namespace
