// polar_kernels.hip — gfx950 (CDNA4) kernels for the polar SC/SCL hot path.
//
// Design: ONE LANE PER LIST PATH.  A wavefront (64 lanes) decodes
// G = 64/GS codewords at once, GS = pow2ceil(L) lanes per codeword; lane `lig` of a group
// IS path index `lig` of the reference (PolarCode.cpp's `l`).  All codewords and all paths
// follow the same successive-cancellation schedule (it depends only on phi and the frozen
// mask), so the 64 lanes run the N-step recursion in lockstep with no divergence; the only
// cross-lane work is the fork/prune step (rank of 2L fork metrics, LIFO path-index stack,
// clone = register shuffle).
//
// Memory: every per-path array is laid out [layer][element][lane] so that the 64 lanes of a
// wave touch 64 consecutive doubles (512 B) — coalesced in HBM/L2, conflict-free in LDS.
// The Tal-Vardy lazy copy (getArrayPointer_*, PolarCode.cpp:305-373) becomes a per-lane,
// per-layer "slot pointer": a path WRITES its own slot (= its lane) and READS the slot its
// pointer names; cloning copies the pointers (registers) only. Because a (layer) array is
// always completely rewritten by all active paths in the same step, no copy-on-write and no
// reference counting is needed, and a permutation of slots inside a group keeps the access
// inside the same 512-byte row.
//
// Element order inside a layer is bit-reversed w.r.t. the reference (position j holds the
// reference's beta = bitrev(j)), so that a node combines elements (j, j+S) and partial sums
// are combined by word-wise XOR/concatenation instead of a bit interleave.
//
// Arithmetic is IEEE double with the reference's formulas and operation order
// (PolarCode.cpp:437-451, 483, 505-506); build with -ffp-contract=off.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "polar_kernels.h"
#include "polar_device.h"
#include "polar_edom.h"

#ifdef POLAR_PROFILE
#define PROF_DECL u64 prof_acc[24] = {0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0}; u64 prof_t = __builtin_readcyclecounter();
#define PROF(i) { u64 t_ = __builtin_readcyclecounter(); prof_acc[i] += t_ - prof_t; prof_t = t_; }
#define PROF_OUT if (lane == 0 && p.pm_out) { for (int i_ = 0; i_ < 24; ++i_) atomicAdd((u64 *)p.pm_out + i_, prof_acc[i_]); }
#define PROF_CNT(i, v) { prof_acc[i] += (u64)(v); }
#else
#define PROF_DECL
#define PROF(i)
#define PROF_CNT(i, v)
#define PROF_OUT
#endif

#include "polar_llr_nodes.h"

// Channel LLRs at the boundary are doubles (the reference's type) or floats (polar_decode_scl_llr_batch_dev_f32):
// a float is widened — exactly — in the load itself, there is no staging copy. ch_row() = row `cw` of p.llr.
template <bool ED>
__device__ __forceinline__ const double *ch_row(const PolarDecodeParams &p, size_t cw, int N) {
    return p.llr_f32 ? reinterpret_cast<const double *>(reinterpret_cast<const float *>(p.llr) + cw * (size_t)N)
                     : p.llr + cw * (size_t)N;
}
// (the exp-domain kernels never see floats — ed_front_kernel has widened and converted the channel values — but folding
// that into the macro measured 1 % SLOWER on the headline kernel: register allocation of the hot loops shifts)
#define CH(row, i) (p.llr_f32 ? (double)reinterpret_cast<const float *>(row)[i] : (row)[i])

// Layer storage helpers --------------------------------------------------------------------
// LDS:    layers with S <= SL; layer of size S starts at element (S-1); element e at [e*64 + lane]
// global: layers with S >  SL; layer of size S starts at element (S-2*SL)
//
// GS  : lanes per codeword (power of two >= L)
// LDS_LOG : log2 of the largest layer size kept in LDS
#ifndef FU
#define FU 4
#endif
#ifndef POLAR_SKIP_L1
#define POLAR_SKIP_L1 1      // the layer of size 1 is never stored (+0.45 %)
#endif
#define POLAR_SADDR 1        // HBM rows of the four-layer visits addressed as SGPR base + 32-bit lane offset (+0.4 %)
#ifndef OCC
#define OCC 4
#endif
// per-phase re-derivation of the lane-dependent invariants (see lane_id_opaque in polar_device.h)
#define LANE_CTX                                   \
    const int lane = LAT ? lane_k : lane_id_opaque();   /* (LAT: 135 of 256 registers in use — nothing to keep short-lived) */ \
    const int lig = lane & (GS - 1);               \
    const int gbase = lane & ~(GS - 1);            \
    (void)lig; (void)gbase;
// NL: 0 = block length taken from p (any code); else log2 of the block length this instantiation is compiled for (the
// headline shapes: every layer size, row offset and loop bound is then a constant)
// LAT = 1: the LATENCY form for small batches (round 4) — ONE codeword per wave. The 64 / GS lane groups that otherwise hold
// different codewords share the ELEMENTS of one codeword: lane = e * GS + l decodes path l and owns the elements j = e (mod 64 / GS) of
// every layer. Everything a path carries (metric, slot pointers, partial sums, history) is replicated in its 64 / GS lanes, which
// execute the leaf steps — fork, prune, clone, CRC, selection: the code below, unchanged — in lockstep with identical operands; only
// the layer visits differ (lat_visit), and the whole state lives in LDS: layers (N - 1) GS doubles (element j of slot s at
// [(j GS + s)]: a wave access is 64 consecutive doubles), the converted channel, the partial-sum and history words. A lone wave of the
// batch kernel pays an HBM round trip per dependent access of its scratch layers and evaluates every element of a layer serially.
template <int GS, int LDS_LOG, int PIPE, bool ED, int NL = 0, int LAT = 0>
__global__ __launch_bounds__(PIPE ? 64 : 256, PIPE ? 2 : OCC) void scl_decode_llr_kernel(PolarDecodeParams p) {
    // ED: exp-domain node arithmetic (see f_node_e); the channel values at p.llr are then in stored form
    // (ed_front_kernel) and every codeword whose decisions are not safely reproduced is reported in p.flags
    // PIPE=1: one wave per block (8 waves/CU, register double-buffering); PIPE=0: four independent
    // waves per block sharing the transcendental tables (16 waves/CU with LDS_LOG = 3)
    constexpr int WPB = PIPE ? 1 : 4;
    constexpr int G = LAT ? 1 : 64 / GS;           // codewords per wave
    constexpr int EL = 64 / GS;                    // (LAT) lanes that share the elements of a path's layers
    // partial-sum and history words: one column per LANE ([word][64]); LAT: per PATH ([word][GS] — the 64 / GS lanes of a path hold
    // the same words and write the same values to the same place)
    constexpr int CST = LAT ? GS : 64;
#define POLAR_CL (LAT ? lig : lane)
#define POLAR_CGB (LAT ? 0 : gbase)
    constexpr int SL = 1 << LDS_LOG;
    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // wave in block (uniform: keeps every per-wave base pointer in SGPRs)
    const int wave_id = blockIdx.x * WPB + wib;             // owns one slice of the global scratch
    const int nwaves = gridDim.x * WPB;
    const int lane_k = lane;           // (the name the LAT code uses where LANE_CTX shadows `lane`)
    (void)lane_k;
    const int lig = lane & (GS - 1);   // path index l of the reference
    const int gbase = lane & ~(GS - 1);
    const int grp = LAT ? 0 : lane / GS;
    const int n = NL ? NL : p.n, N = NL ? (1 << NL) : p.N, K = p.K, L = p.L;
    const u64 gmask = (GS == 64) ? ~0ull : ((1ull << GS) - 1ull);

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double *tabs = reinterpret_cast<double *>(smem);                           // T[64] RC[129] LC[129] (+2 pad), per block
    constexpr size_t WAVE_LDS = (size_t)(2 * SL - 1) * 64 * 8 + 128 * 8 + 128;  // bytes per wave
    unsigned char *wbase = smem + 324 * 8 + (size_t)wib * WAVE_LDS;
    double *lds_llr = reinterpret_cast<double *>(wbase);                       // [(2*SL-1)][64] (LAT: not there — its layers are lat_a)
    double *sortbuf = lds_llr + (LAT ? 0 : (size_t)(2 * SL - 1) * 64);         // [128]
    // (plain pointers, ordered by wave_mem_fence(): a volatile-qualified pointer loses its LDS address
    // space and every access becomes a system-coherent FLAT operation that waits for all memory)
    unsigned char *stackv = reinterpret_cast<unsigned char *>(sortbuf + 128);   // [64]
    unsigned char *srcof = stackv + 64;                                         // [64]
    for (int i = threadIdx.x; i < 322; i += WPB * 64) tabs[i] = p.tabs[i];
    if (WPB > 1) __syncthreads();
    const Tabs tb = {tabs, tabs + 64, tabs + 64 + 129};
    u64 guard = 0;                                  // (ED) wave mask of lanes with an undecidable |x| < 40 test
    double gacc = __builtin_inf();                  // (ED) per lane: smallest distance of a node's smaller E to the |x| < 40 threshold
    auto FN = [&](double a, double b) -> double {
        if constexpr (ED) return f_node_e_acc(a, b, gacc); else return f_node(a, b, tb);
    };
    // g-node of element with partial-sum bit `bi` of the word `cw_` (u = (cw_ >> bi) & 1)
    auto GN = [&](double a, double b, uint32_t cw_, int bi) -> double {
        if constexpr (ED) return g_node_e(a, b, cw_ << (31 - bi), tb); else return g_node(a, b, (cw_ >> bi) & 1u);
    };
    // (fallback pass: codewords come from p.cw_list; Monte-Carlo: only the first *p.n_dev rows are alive)
    const long Bv = p.cw_count ? (long)*p.cw_count : (p.n_dev ? ((long)*p.n_dev < p.B ? (long)*p.n_dev : p.B) : p.B);

    // per-wave global scratch (LAT: the same arrays in LDS, behind the wave's other LDS data)
    const size_t big_elems = (N > 2 * SL) ? (size_t)(N - 2 * SL) : 0;
    double *g_llr = LAT ? nullptr : p.llr_scr + (size_t)wave_id * big_elems * 64;
    const int cwords = (N >= 128) ? (N / 32 - 2) : 0;                          // words of big C layers (S >= 64)
    double *lat_ch = reinterpret_cast<double *>(stackv + 128);                  // (LAT) [N] channel, stored form, kernel element order
    double *lat_a = lat_ch + N;                                                // (LAT) layers: size S at (S - 1) GS, element j of slot s at (j GS + s)
    uint32_t *lat_w = reinterpret_cast<uint32_t *>(lat_a + (size_t)N * GS);
    uint32_t *g_cl = LAT ? lat_w : p.c_scr + (size_t)wave_id * 2 * (size_t)cwords * 64;
    uint32_t *g_cr = g_cl + (size_t)cwords * CST;
    uint32_t *g_hist = LAT ? g_cr + (size_t)cwords * CST : p.hist_scr + (size_t)wave_id * 3 * (size_t)p.W * 64;    // decision words [W][64]
    uint32_t *g_horg = g_hist + (size_t)p.W * CST;                             // link to the previous word's slot
    uint32_t *g_tb = g_horg + (size_t)p.W * CST;                               // winner's words, per-lane copy
    // ---- table mode (list size 17..32, N >= 1024, exp-domain): layers 1 and 2 are never stored per path.
    // Every path's layer-1 value x1[e] = g(ch, ch', u1[e]) is one of TWO numbers, its layer-2 value one of 2 / 4 / 8
    // (phi = N/4: g of the shared first-half layer 1 with u2[j]; phi = N/2: f(x1[j], x1[j+N/4]) -> u1[j], u1[j+N/4];
    // phi = 3N/4: g(...) -> additionally u2[j]) — the bits being the path's partial sums. The 32 lanes of a codeword
    // build those values ONCE per codeword (T2[j][variant], 64 B per element: one cache line serves all paths) and each
    // path keeps 3 bits per element (V words, reached through a slot pointer like every per-path array). The visits of
    // layer 3 gather their inputs from the table. Two of the seven HBM-resident layers disappear.
    const int S1 = N / 2, S2 = N / 4;
    const bool tbl = ED && !PIPE && GS == 32 && N >= 1024 && p.tab_scr != nullptr && p.prefix_q > 0;
    double *tab_w = tbl ? p.tab_scr + (size_t)wave_id * G * (size_t)(3 * N) : nullptr;       // per codeword: X[N/2][2], T2[N/4][8]
    uint32_t *g_v = tbl ? p.var_scr + (size_t)wave_id * (size_t)(S2 / 8) * 64 : nullptr;     // V words [N/32][64]

    // Work distribution: a wave's first group of codewords is its own index, every further one comes
    // from a device counter. Waves do not take equally long (per-wave time spreads by ~ +-15 %), and with
    // a static stride the launch ends when the unluckiest wave has finished ALL its groups.
    for (long g0 = (long)wave_id * G; g0 < Bv;) {
        const long cwi = g0 + grp;                 // position in the work list
        const bool valid = (cwi < Bv);
        const long cw = (p.cw_list && valid) ? (long)p.cw_list[cwi] : cwi;   // codeword (row of llr / out)
        guard = 0;
        gacc = __builtin_inf();
        auto cw_of_lane = [&](int ln) -> size_t {
            const long i = g0 + (LAT ? 0 : ln / GS);
            return p.cw_list ? (size_t)p.cw_list[i < Bv ? i : Bv - 1] : (size_t)i;     // (lanes past the end of the work list: any valid row)
        };
        auto FN2 = [&](double a0_, double b0_, double a1_, double b1_, double &r0_, double &r1_) {
            if constexpr (ED) { r0_ = f_node_e_acc(a0_, b0_, gacc); r1_ = f_node_e_acc(a1_, b1_, gacc); }
            else f_node2(a0_, b0_, a1_, b1_, tb, r0_, r1_);
        };

        // initializeDataStructures + assignInitialPath (PolarCode.cpp:195-272): the inactive
        // stack holds 0..L-1, the first pop (initial path) is L-1.
        bool active = valid && (lig == L - 1);
        u64 actw = __ballot(active);               // wave mask of `active`, refreshed where it changes (initial path, kill / clone)
        double pm = 0.0;
        int sp = L - 1;                            // group-uniform stack pointer
        if (lig < L - 1) stackv[gbase + lig] = (unsigned char)lig;
        P16 pL = {0, 0};                           // LLR slot pointer per layer (index sh = n - lam)
        P16 pC = {0, 0};                           // column-0 C slot pointer for big layers (index sh)
        u64 clsmall = 0;                           // column-0 partial sums of layers with S <= 32: bits [S, 2S)
        uint32_t hword = 0;                        // decisions of the current 32 unfrozen steps
        int origin = lig;                          // slot that holds this path's flushed history
        unsigned t = 0;                            // unfrozen steps so far (wave-uniform)
        if constexpr (LAT) {
            // channel row -> stored form (or the plain LLRs for the LLR-domain arithmetic), kernel element order: element e is channel
            // position bitrev_n(e), so that the pair (2b, 2b + 1) the top layer combines sits at (j, j + N/2); input guard of ed_front_kernel
            bool any = false, sized = false;
            const size_t row = (size_t)(valid ? cw : 0) * (size_t)N;
            for (int i0 = 0; i0 < N; i0 += 64 * 8) {
                double x[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int i = i0 + 64 * k + lane;
                    x[k] = (i < N) ? (p.llr_f32 ? (double)reinterpret_cast<const float *>(p.llr)[row + i] : p.llr[row + i]) : 1.0;
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int i = i0 + 64 * k + lane;
                    double v = x[k];
                    if constexpr (ED) { bool f; v = ed_from_channel(x[k], tb, f); any |= f; sized |= fabs(x[k]) >= 0.1; }
                    if (i < N) lat_ch[__brev((unsigned)i) >> (32 - n)] = v;
                }
            }
            if constexpr (ED) { if (wave_any(any) || !wave_any(sized)) guard = ~0ull; }
        }
        wave_mem_fence();

        // ================= all-frozen prefix (computed by prefix_kernel) =================
        // Until the first unfrozen position only ONE path exists per codeword and every decision is the
        // frozen 0: prefix_kernel has already produced, per codeword, the f-chain of the layers above
        // the prefix block (node 0, sizes N/2..Q, contiguous in p.pre) and the path metric of the first
        // Pe leaves. The walk resumes at phi = Pe; a layer of size 2S >= Q is read from that buffer
        // (same addresses for every path of the codeword: broadcast) until its first rewrite at phi = 2S.
        int phi_start = 0, forced_top = 0;
        const double *pre_cw = nullptr;
        if (!LAT && p.prefix_q > 0) {               // (the LAT form is launched without a prefix pass: its walk starts at leaf 0)
            const int Q = p.prefix_q, Pe = p.prefix_len;
            pre_cw = p.pre + (size_t)(valid ? cw : 0) * (size_t)(N - Q + 1);
            if (active) {
                pm = pre_cw[0];
                // zero partial sums of every completed (all-frozen) left subtree
                for (int S = 64; S <= Q && S <= N / 2; S <<= 1) {
                    uint32_t *cz = g_cl + (size_t)(S / 32 - 2) * 64 + lane;
                    for (int w = 0; w < S / 32; ++w) cz[(size_t)w * 64] = 0u;
                    pC.set(__builtin_ctz((unsigned)S), lig);
                }
            }
            wave_mem_fence();
            phi_start = Pe;
            if (Pe < Q) forced_top = n - __builtin_ctz((unsigned)Q) + 1;     // recompute from the block downwards
        }

        PROF_DECL
#ifdef POLAR_MARGIN
        // development aid (tools/margin_profile.py): the smallest gap, over all pruning fork steps of this codeword, between
        // the worst surviving and the best discarded fork metric — what a lower-precision state would have to resolve
        double mingap = __builtin_inf();
#endif
        // per-leaf control word (frozen flag, rate-0 block size): a SCALAR load through the constant
        // address space, issued one leaf ahead — as a plain global load it is a vector memory round
        // trip on the critical path of every leaf
        typedef const uint32_t __attribute__((address_space(4))) *kconst_u32;
        const kconst_u32 ctlp = (kconst_u32)(uintptr_t)p.ctl;
        // LAT (round 6): a lone wave cannot hide the scalar load — it shares lgkmcnt with the LDS operations, so the first LDS wait of
        // every leaf step also waited for the control word of the NEXT one (a constant-cache miss every 16 leaves: ~1 us). The control
        // words of 64 leaves ride in one VGPR (lane i: leaf window + i), loaded one window ahead by a vector load (vmcnt: nothing
        // in the leaf loop waits on it); a leaf's word is one v_readlane.
        uint32_t ctlv = 0, ctlv_n = 0;
        // (LAT, round 6) the two lowest layers are not stored: the four values of the layer of size 4 a path's current quad of leaves hangs
        // off (lat_x, read once per quad) and the two of the layer of size 2 (lat_y) ride in registers, replicated in the lanes of the
        // path like the rest of its state and copied with it by a clone (polar_scl_visits.inc, polar_scl_leaf.inc)
        // lat_x is read at the first leaf of a quad and used again at its third, lat_y is set at the left leaf of a pair and used
        // again at the right one: the walk of this form starts at leaf 0 and all-frozen blocks are aligned, so both are always there.
        double lat_x[4] = {0.0, 0.0, 0.0, 0.0}, lat_y0 = 0.0, lat_y1 = 0.0;
        bool lat_yv = false;                        // (block lengths below 8 only: lat_y holds the pair's operands)
        (void)lat_x; (void)lat_y0; (void)lat_y1; (void)lat_yv;
        auto ctl_window = [&](int w0) -> uint32_t { const int i_ = w0 + lane_id_opaque(); return p.ctl[i_ < N ? i_ : 0]; };
        uint32_t ctl_next;
        if constexpr (LAT) {
            ctlv = ctl_window(phi_start & ~63);
            ctlv_n = ctl_window((phi_start & ~63) + 64);
            ctl_next = (uint32_t)__builtin_amdgcn_readlane((int)ctlv, phi_start & 63);
        } else ctl_next = ctlp[phi_start];
        for (int phi = phi_start; phi < N; ++phi) {
            PROF(0)
            const uint32_t ctl = ctl_next;
            // recursivelyUpdateC (PolarCode.cpp:457-473) from the layer of size S upwards: X = column 1 of
            // that layer (the S bits just completed by a RIGHT child with node index ph)
            auto update_c = [&](int S, uint32_t X, int ph) {
                LANE_CTX
                if constexpr (LAT) {
                    // (round 6) after a right LEAF: the levels S = 1 ... 16 live in the register word and are walked with constant
                    // shifts on its 32-bit halves — the generic loop below shifts a 64-bit word by a variable amount at every
                    // level (a dozen instructions of a lone wave's issue slots each); from the layer of 32 on it takes over
                    if (S == 1 && N >= 128) {
                        uint32_t lo = (uint32_t)clsmall, hi = (uint32_t)(clsmall >> 32);
                        bool done = false;
                        auto level = [&](auto SC_) {
                            constexpr int Sc = decltype(SC_)::value;
                            if (done) return;
                            const uint32_t cl = (Sc == 16) ? (lo >> 16) : ((lo >> Sc) & ((1u << Sc) - 1u));
                            const uint32_t nw = (cl ^ X) | (X << Sc);
                            if (!((ph >> 1) & 1)) {
                                if constexpr (Sc == 16) hi = nw;
                                else if constexpr (Sc == 8) lo = (lo & 0x0000FFFFu) | (nw << 16);
                                else lo = (lo & ~(((1u << (2 * Sc)) - 1u) << (2 * Sc))) | (nw << (2 * Sc));
                                done = true;
                            } else { X = nw; ph >>= 1; }
                        };
                        level(std::integral_constant<int, 1>{}); level(std::integral_constant<int, 2>{}); level(std::integral_constant<int, 4>{});
                        level(std::integral_constant<int, 8>{}); level(std::integral_constant<int, 16>{});
                        if (done) {
                            if (active) clsmall = ((u64)hi << 32) | lo;
                            return;
                        }
                        S = 32;               // (nothing was written: every level so far was a right child)
                    }
                }
                for (;;) {
                    if (4 * S > N) break;                   // C_0 is never read (PolarCode.cpp writes it, nobody uses it)
                    const int psi = ph >> 1;
                    const bool to_right = (psi & 1);        // result becomes column 1 of C_{lam-1}
                    const int sh = __builtin_ctz((unsigned)S);
                    if (S <= 16) {
                        uint32_t cl = (uint32_t)(clsmall >> S) & ((1u << S) - 1u);
                        uint32_t nw = (cl ^ X) | (X << S);   // 2S bits
                        if (!to_right) {
                            const int S2 = 2 * S;
                            const u64 m = ((S2 == 32) ? 0xFFFFFFFFull : ((1ull << S2) - 1ull)) << S2;
                            if (active) clsmall = (clsmall & ~m) | ((u64)nw << S2);
                        }
                        X = nw;
                    } else if (S == 32) {
                        uint32_t cl = (uint32_t)(clsmall >> 32);
                        uint32_t *dst = (to_right ? g_cr : g_cl) + (size_t)0 * CST + POLAR_CL;   // layer size 64 -> word offset 0
                        if (active) { dst[0] = cl ^ X; dst[CST] = X; }
                        if (!to_right && active) pC.set(sh + 1, lig);
                    } else {
                        const int nwd = S / 32;
                        const uint32_t *cl = g_cl + (size_t)(nwd - 2) * CST + POLAR_CGB + pC.get(sh);
                        const uint32_t *cr = g_cr + (size_t)(nwd - 2) * CST + POLAR_CL;
                        uint32_t *dst = (to_right ? g_cr : g_cl) + (size_t)(2 * nwd - 2) * CST + POLAR_CL;
                        if (active) {
                            // all loads of a chunk first, then the stores: one memory round trip per chunk instead of
                            // one per word (a load behind a store waits for the store's acknowledgement as well)
                            auto chunk = [&](auto CH_) {
                                constexpr int CH = decltype(CH_)::value;
                                for (int w = 0; w < nwd; w += CH) {
                                    uint32_t r[CH], l[CH];
#pragma unroll
                                    for (int i = 0; i < CH; ++i) { r[i] = cr[(size_t)(w + i) * CST]; l[i] = cl[(size_t)(w + i) * CST]; }
#pragma unroll
                                    for (int i = 0; i < CH; ++i) { dst[(size_t)(w + i) * CST] = l[i] ^ r[i]; dst[(size_t)(w + i + nwd) * CST] = r[i]; }
                                }
                            };
                            if (nwd >= 8) chunk(std::integral_constant<int, 8>{});
                            else if (nwd == 4) chunk(std::integral_constant<int, 4>{});
                            else chunk(std::integral_constant<int, 2>{});
                            if (!to_right) pC.set(sh + 1, lig);
                        }
                    }
                    wave_mem_fence();
                    if (!to_right) break;
                    S *= 2;
                    ph = psi;
                }
            };
            // all-frozen aligned block of 2^zb leaves starting here (host schedule), 0 = ordinary leaf
            const int zb = (int)(ctl >> 1) & 0x7F;
            {
                const int nphi = phi + (1 << zb);
                if constexpr (LAT) {
                    if (POLAR_UNLIKELY2(((nphi ^ phi) & ~63) != 0)) { ctlv = ctlv_n; ctlv_n = ctl_window((nphi & ~63) + 64); }     // (a block is at most 8 leaves: one window at a time)
                    ctl_next = (uint32_t)__builtin_amdgcn_readlane((int)ctlv, nphi & 63);
                } else ctl_next = ctlp[nphi < N ? nphi : 0];
            }
            const int lam_stop = n - zb;
            // recursivelyCalcLLR for this leaf, then the leaf itself (fragments of this function body: the kernel is one function, its
            // text is kept in three files)
#include "polar_scl_visits.inc"
#include "polar_scl_leaf.inc"
        }  // phi
        PROF_OUT

#include "polar_scl_finish.inc"
        // next group
        if (p.work) {
            unsigned nxt = 0;
            if (lane == 0) nxt = atomicAdd(p.work, 1u);
            g0 = ((long)nwaves + (long)__builtin_amdgcn_readfirstlane((int)nxt)) * G;
        } else {
            g0 += (long)nwaves * G;
        }
    }  // codeword groups
}

// ------------------------------------------------------------------------------------------
// prefix_kernel — the all-frozen prefix [0, Pe) of every codeword, 32 lanes per codeword.
// Until the first unfrozen position one path exists and every decision is the frozen 0, so the leaf
// LLRs of the prefix are a fixed f/g dataflow of the channel LLRs (g with u = 0). The 32 lanes share
// the ELEMENTS: (1) f-chain of the layers above the prefix block (node 0 of the layers of size
// N/2 .. Q), written contiguously per codeword (pre[cw][1 + N - 2S + j]) for the decode kernel to
// read; (2) the Q block values expanded in registers by a log2(Q)-stage butterfly into the Q leaf
// LLRs; (3) the path metric accumulated over the first Pe leaves in order, same operations and order
// as continuePaths_FrozenBit (PolarCode.cpp:475-487) -> pre[cw][0].
template <bool ED>
// Round 4: (a) with ech_out (exp-domain, staged) the first pass reads the caller's RAW channel pairs, converts them (the input
// guard of ed_front_kernel included), writes the stored form for the decode kernel and feeds the f-node from registers: the separate
// conversion pass (a read and a write of the whole batch) is gone; (b) the layers below the first are computed from the staged copy
// in LDS, in place (element j and j + S belong to the same lane), instead of from what was just written to global memory: the
// write -> read round trip through the L2 between the layers of a codeword was the kernel's time.
__global__ __launch_bounds__(256) void prefix_kernel(PolarDecodeParams p, int staged, double *ech_out) {
    __shared__ double tabs[324];
    // staged: the first pass (channel -> layer N/2) reads the channel pairs in their own order — element j of the layer comes
    // from the pair at bitrev(j), a 16-B read from a different line for every lane when read in element order — and
    // turns the results into element order through LDS ([8 codewords][N/2] doubles, dynamic)
    extern __shared__ double pstage[];

    for (int i = threadIdx.x; i < 322; i += 256) tabs[i] = p.tabs[i];
    __syncthreads();
    const Tabs tb = {tabs, tabs + 64, tabs + 64 + 129};
    u64 guard = 0;
    auto FN = [&](double a, double b) -> double {
        if constexpr (ED) return f_node_e(a, b, guard); else return f_node(a, b, tb);
    };
    auto GN0 = [&](double a, double b) -> double {
        if constexpr (ED) return g_node_e(a, b, 0u, tb); else return g_node(a, b, 0u);
    };
    const int lane = threadIdx.x & 63, lig = lane & 31, gbase = lane & 32;
    const int n = p.n, N = p.N, Q = p.prefix_q, Pe = p.prefix_len;
    const int R = Q >> 5;
    const long per_block = 8;
    const long Bv = p.n_dev ? ((long)*p.n_dev < p.B ? (long)*p.n_dev : p.B) : p.B;
    for (long c0 = (long)blockIdx.x * per_block; c0 < Bv; c0 += (long)gridDim.x * per_block) {
        const long cw = c0 + (threadIdx.x >> 5);
        const bool valid = cw < Bv;
        const double *in0 = ch_row<ED>(p, (size_t)(valid ? cw : 0), N);
        double *pre = const_cast<double *>(p.pre) + (size_t)(valid ? cw : 0) * (size_t)(N - Q + 1);
        double x[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        bool bad_in = false;                 // (fused conversion: this codeword's input guard)
        for (int S = N / 2; S >= Q; S >>= 1) {
            if (valid && staged && 2 * S == N) {
                double *stg = pstage + (size_t)(threadIdx.x >> 5) * (size_t)S;
                double *outp = pre + 1;
                if (ED && ech_out) {
                    double *erow = ech_out + (size_t)cw * (size_t)N;
                    bool any = false, sized = false;
                    // (round 6: the loads of four pairs first — the stores to erow may alias the row as far as the compiler knows, so
                    // the plain loop waited for memory once per pair: 32 round trips per lane on two waves per SIMD)
                    int m = lig;
                    for (; m + 96 < S; m += 128) {
                        double x0[4], x1[4];
#pragma unroll
                        for (int k = 0; k < 4; ++k) { x0[k] = CH(in0, 2 * (m + 32 * k)); x1[k] = CH(in0, 2 * (m + 32 * k) + 1); }
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const int mk = m + 32 * k;
                            bool f0, f1;
                            const double a = ed_from_channel(x0[k], tb, f0), b = ed_from_channel(x1[k], tb, f1);
                            any |= f0 | f1;
                            sized |= (fabs(x0[k]) >= 0.1) | (fabs(x1[k]) >= 0.1);
                            erow[2 * mk] = a; erow[2 * mk + 1] = b;
                            stg[__brev((unsigned)mk) >> (33 - n)] = FN(a, b);  // element j = bitrev_{n-1}(m)
                        }
                    }
                    for (; m < S; m += 32) {
                        const double x0 = CH(in0, 2 * m), x1 = CH(in0, 2 * m + 1);
                        bool f0, f1;
                        const double a = ed_from_channel(x0, tb, f0), b = ed_from_channel(x1, tb, f1);
                        any |= f0 | f1;
                        sized |= (fabs(x0) >= 0.1) | (fabs(x1) >= 0.1);
                        erow[2 * m] = a; erow[2 * m + 1] = b;
                        stg[__brev((unsigned)m) >> (33 - n)] = FN(a, b);       // element j = bitrev_{n-1}(m)
                    }
                    // input guard per codeword (ed_front_kernel): a non-finite or < 1e-9 value, or no value >= 0.1 at all
                    const u64 m_any = __builtin_amdgcn_ballot_w64(any), m_sz = __builtin_amdgcn_ballot_w64(sized);
                    const u64 half = gbase ? 0xFFFFFFFF00000000ull : 0x00000000FFFFFFFFull;
                    bad_in = ((m_any & half) != 0) || ((m_sz & half) == 0);
                } else {
                    for (int m = lig; m < S; m += 32) {
                        const double r = FN(CH(in0, 2 * m), CH(in0, 2 * m + 1));
                        stg[__brev((unsigned)m) >> (33 - n)] = r;              // element j = bitrev_{n-1}(m)
                    }
                }
                wave_mem_fence();                                            // (the 32 lanes of a codeword are in one wave)
                for (int j = lig; j < S; j += 32) {
                    const double r = stg[j];
                    outp[j] = r;
                    if (S == Q) {
#pragma unroll
                        for (int rr = 0; rr < 8; ++rr) if (rr == (j >> 5)) x[rr] = r;
                    }
                }
                wave_mem_fence();
            } else if (valid && staged) {
                // the layer above is still in the staging row (elements [0, 2S)): in place, lane-local
                double *stg = pstage + (size_t)(threadIdx.x >> 5) * (size_t)(N / 2);
                double *outp = pre + 1 + (size_t)(N - 2 * S);
                for (int j = lig; j < S; j += 32) {
                    const double r = FN(stg[j], stg[j + S]);
                    stg[j] = r;
                    outp[j] = r;
                    if (S == Q) {
#pragma unroll
                        for (int rr = 0; rr < 8; ++rr) if (rr == (j >> 5)) x[rr] = r;
                    }
                }
                wave_mem_fence();
            } else if (valid) {
                const bool from_ch = (2 * S == N);
                const double *inp = pre + 1 + (size_t)(N - 4 * S);      // layer of size 2S (unused when from_ch)
                double *outp = pre + 1 + (size_t)(N - 2 * S);
                // four elements per lane and pass: their eight loads are in flight together (S is a multiple of 128:
                // S >= Q >= 64 ... the tail loop takes what is left)
                int j = lig;
                for (; j + 96 < S; j += 128) {
                    double a[4], b[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        if (from_ch) {
                            const unsigned idx = __brev((unsigned)(j + 32 * k)) >> (32 - n);
                            a[k] = CH(in0, idx); b[k] = CH(in0, idx + 1);
                        } else {
                            a[k] = inp[j + 32 * k]; b[k] = inp[j + 32 * k + S];
                        }
                    }
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const double r = FN(a[k], b[k]);
                        outp[j + 32 * k] = r;
                        if (S == Q) {
#pragma unroll
                            for (int rr = 0; rr < 8; ++rr) if (rr == ((j + 32 * k) >> 5)) x[rr] = r;
                        }
                    }
                }
                for (; j < S; j += 32) {
                    double a, b;
                    if (from_ch) {
                        unsigned idx = __brev((unsigned)j) >> (32 - n);
                        a = CH(in0, idx); b = CH(in0, idx + 1);
                    } else {
                        a = inp[j]; b = inp[j + S];
                    }
                    const double r = FN(a, b);
                    outp[j] = r;
                    if (S == Q) {
#pragma unroll
                        for (int rr = 0; rr < 8; ++rr) if (rr == (j >> 5)) x[rr] = r;
                    }
                }
            }
            wave_mem_fence();
        }
        // butterfly: a stage with half-size h turns every node of size 2h into its f-child (lower half)
        // and its g-child (upper half, u = 0); value index i = r*32 + lig
        for (int h = Q / 2; h >= 1; h >>= 1) {
            if (h >= 32) {
                const int hr = h >> 5;
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    if (r < R && (r & hr) == 0) {
#pragma unroll
                        for (int r2 = 0; r2 < 8; ++r2) {
                            if (r2 == r + hr) {
                                const double lo = x[r], hi = x[r2];
                                x[r] = FN(lo, hi);
                                x[r2] = GN0(lo, hi);
                            }
                        }
                    }
                }
            } else {
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    if (r < R) {
                        const double mine = x[r];
                        const double other = shfl_d(mine, lane ^ h);
                        x[r] = (lig & h) ? GN0(other, mine) : FN(mine, other);
                    }
                }
            }
        }
        // path metric over the leaves 0..Pe-1 in order (leaf phi sits in x[phi>>5] of lane phi&31)
        double acc = 0.0;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            if (r < R && r * 32 < Pe) {
                bool ng; double al, sneg, spos;
                leaf_terms<ED>(x[r], true, __builtin_amdgcn_ballot_w64(true), tb, ng, al, sneg, spos);
                const double spv = ng ? spos : sneg;
                const int cnt = (Pe - r * 32 < 32) ? (Pe - r * 32) : 32;
                // (the sum must run in leaf order; unrolled, the 32 cross-lane reads are in flight together and only the
                // additions are serial)
#pragma unroll
                for (int c = 0; c < 32; ++c) {
                    const double t = shfl_d(spv, gbase + c);
                    acc = (c < cnt) ? acc + t : acc;
                }
            }
        }
        if (valid && lig == 0) pre[0] = acc;
        if constexpr (ED) {
            const bool fl = ((guard >> gbase) & 0xFFFFFFFFull) != 0;
            if (ech_out) { if (valid && lig == 0) p.flags[cw] = (fl || bad_in) ? 1 : 0; }       // (no conversion pass has set it)
            else if (valid && lig == 0 && fl) p.flags[cw] = 1;
            guard = 0;
        }
        wave_mem_fence();
    }
}

// This file is compiled four times (polar_amd/build.py): POLAR_ED_TU = 0 instantiates the LLR-domain kernels and
// the small helper kernels, POLAR_ED_TU = 1 the exp-domain kernels of the groups of 4, 8, 16 and 64 lanes, POLAR_ED_TU = 2 the
// exp-domain list of 32 — translation units that build in parallel, that one with its own scheduler options
// (max-memory-clause strategy + the AMDGPU register-pressure trackers: +2.2 ... 3.8 % on the headline kernel, -11 % on the
// groups of 8: build.py) —, POLAR_ED_TU = 3 the exp-domain one-codeword-per-wave (LAT) kernels.
#ifndef POLAR_ED_TU
#define POLAR_ED_TU 0
#endif
#if POLAR_ED_TU < 2
#if POLAR_ED_TU
hipError_t polar_launch_prefix_ed1(const PolarDecodeParams &p, double *ech_out, hipStream_t st) {
#else
hipError_t polar_launch_prefix_ed0(const PolarDecodeParams &p, double *ech_out, hipStream_t st) {
#endif
    long blocks = (p.B + 7) / 8;
    if (blocks > 8192) blocks = 8192;
    const size_t stage = (size_t)8 * (size_t)(p.N / 2) * sizeof(double);        // 64 KiB at N = 2048: two blocks per CU
    const int staged = polar_prefix_is_staged(p.N);
    if (ech_out && !staged) return hipErrorInvalidValue;                        // (the fused conversion exists for the staged first pass)
    hipLaunchKernelGGL(prefix_kernel<POLAR_ED_TU != 0>, dim3((unsigned)blocks), dim3(256), staged ? stage : 0, st, p, staged, ech_out);
    return hipGetLastError();
}
#endif  // POLAR_ED_TU < 2

#if !POLAR_ED_TU
int polar_prefix_is_staged(int N) { return (N >= 64 && (size_t)8 * (size_t)(N / 2) * sizeof(double) <= 64 * 1024) ? 1 : 0; }
// ech_out (exp-domain only, staged block lengths only): p.llr are the caller's RAW rows; the kernel converts them, writes the stored
// form there and sets p.flags (0 / 1) for every codeword — no ed_front_kernel before it
hipError_t polar_launch_prefix(const PolarDecodeParams &p, bool ed, double *ech_out, hipStream_t st) {
    return ed ? polar_launch_prefix_ed1(p, ech_out, st) : polar_launch_prefix_ed0(p, nullptr, st);
}
// ------------------------------------------------------------------------------------------
// ed_front_kernel — channel LLRs -> stored form of the exp-domain kernel (p.llr -> p.ech), plus the
// input guard: flags[cw] = 1 when the codeword holds a non-finite LLR or one below 1e-9 (the
// reference's f-node results are then its own rounding noise), else 0.
template <typename TIN>
__global__ __launch_bounds__(256) void ed_front_kernel(const TIN *llr, double *ech, uint8_t *flags, const double *tabs_g, int N, long B, const unsigned *n_dev) {
    if (n_dev && (long)*n_dev < B) B = (long)*n_dev;
    __shared__ double tabs[324];
    for (int i = threadIdx.x; i < 322; i += 256) tabs[i] = tabs_g[i];
    __syncthreads();
    const Tabs tb = {tabs, tabs + 64, tabs + 64 + 129};
    const int lane = threadIdx.x & 63;
    for (long cw = (long)blockIdx.x * 4 + (threadIdx.x >> 6); cw < B; cw += (long)gridDim.x * 4) {
        const TIN *src = llr + (size_t)cw * N;
        double *dst = ech + (size_t)cw * N;
        bool any = false, sized = false;
        for (int i = lane; i < N; i += 64) {
            bool f;
            const double x = (double)src[i];
            dst[i] = ed_from_channel(x, tb, f);
            any |= f;
            sized |= fabs(x) >= 0.1;
        }
        // A row whose WHOLE channel is tiny (no |llr| >= 0.1: e.g. LLRs scaled by 1e-3, whose largest values are 0.02 ... 0.03;
        // an ordinary row has hundreds of values above 1) is decided by the reference at the rounding noise of its own f-chains from the second layer on —
        // the same class as the per-value guard above, which looks at one value at a time and lets such rows pass. They go to
        // the LLR-domain kernel, which follows the reference's arithmetic further down (fuzz: 4 of 3 600 such rows differed).
        const bool bad = wave_any(any) || !wave_any(sized);
        if (lane == 0) flags[cw] = bad ? 1 : 0;
    }
}
hipError_t polar_launch_ed_front(const void *llr, int llr_f32, double *ech, uint8_t *flags, const double *tabs, int N, long B, const unsigned *n_dev, hipStream_t st) {
    long blocks = (B + 3) / 4;
    if (blocks > 16384) blocks = 16384;
    if (llr_f32) hipLaunchKernelGGL(ed_front_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, st, (const float *)llr, ech, flags, tabs, N, B, n_dev);
    else hipLaunchKernelGGL(ed_front_kernel<double>, dim3((unsigned)blocks), dim3(256), 0, st, (const double *)llr, ech, flags, tabs, N, B, n_dev);
    return hipGetLastError();
}
// flagged codewords -> work list of the fallback pass (order irrelevant: every codeword is independent)
__global__ __launch_bounds__(256) void ed_collect_kernel(const uint8_t *flags, long B, const unsigned *n_dev, uint32_t *list, unsigned *count) {
    if (n_dev && (long)*n_dev < B) B = (long)*n_dev;
    for (long cw = (long)blockIdx.x * 256 + threadIdx.x; cw < B; cw += (long)gridDim.x * 256)
        if (flags[cw]) list[atomicAdd(count, 1u)] = (uint32_t)cw;
}
hipError_t polar_launch_ed_collect(const uint8_t *flags, long B, const unsigned *n_dev, uint32_t *list, unsigned *count, hipStream_t st) {
    long blocks = (B + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(ed_collect_kernel, dim3((unsigned)blocks), dim3(256), 0, st, flags, B, n_dev, list, count);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
int polar_decode_waves_per_block(int pipe) { return pipe ? 1 : 4; }
size_t polar_decode_lds_bytes(int lds_log, int pipe) {
    return 324 * 8 + (size_t)polar_decode_waves_per_block(pipe) * ((size_t)((2u << lds_log) - 1) * 64 * 8 + 128 * 8 + 128);
}

#endif  // !POLAR_ED_TU

template <int GS, bool ED>
static hipError_t launch_gs(const PolarDecodeParams &p, int lds_log, int pipe, int grid, hipStream_t st) {
    // `grid` counts WAVES; blocks = grid / waves-per-block (the host rounds grid to a multiple)
    size_t lds = polar_decode_lds_bytes(lds_log, pipe);
    const int wpb = polar_decode_waves_per_block(pipe);
#define POLAR_LAUNCH(LL, PP) hipLaunchKernelGGL((scl_decode_llr_kernel<GS, LL, PP, ED>), dim3(grid / wpb), dim3(64 * wpb), lds, st, p)
#ifndef POLAR_NO_FIXED_N
    if constexpr (GS == 32 && ED) {
        // the headline shapes (N = 2048 / 1024, list of 17..32, default tuning) have instantiations of their own
        if (lds_log == 3 && !pipe && (p.n == 11 || p.n == 10)) {
            if (p.n == 11) hipLaunchKernelGGL((scl_decode_llr_kernel<GS, 3, 0, ED, 11>), dim3(grid / wpb), dim3(64 * wpb), lds, st, p);
            else hipLaunchKernelGGL((scl_decode_llr_kernel<GS, 3, 0, ED, 10>), dim3(grid / wpb), dim3(64 * wpb), lds, st, p);
            return hipGetLastError();
        }
    }
#endif
    switch (lds_log * 2 + (pipe ? 1 : 0)) {
        case 6: POLAR_LAUNCH(3, 0); break;
#ifndef POLAR_DEV_ONE    // (development builds: the default tuning only)
        case 4: POLAR_LAUNCH(2, 0); break;
        case 7: POLAR_LAUNCH(3, 1); break;
        case 8: POLAR_LAUNCH(4, 0); break;
        case 9: POLAR_LAUNCH(4, 1); break;
        case 10: POLAR_LAUNCH(5, 0); break;
        case 11: POLAR_LAUNCH(5, 1); break;
#endif
        default: return hipErrorInvalidValue;
    }
#undef POLAR_LAUNCH
    return hipGetLastError();
}

// LAT instantiations (one codeword per wave, state in LDS): exp-domain arithmetic for the groups of 2, 4 and 8 lanes in a translation
// unit of their own (POLAR_ED_TU = 3, round 6: a lone wave wants code the throughput kernels do not — fewer branches, see
// polar_amd/build.py FLAGS_LAT), LLR-domain arithmetic for the groups of 2 (POLAR_ED_TU = 0)
template <int GS, bool ED>
static hipError_t launch_lat(const PolarDecodeParams &p, int blocks, hipStream_t st) {
    const size_t lds = polar_decode_lat_lds_bytes(p.N, GS, p.W);
    // (per launch, not once per process: the attribute belongs to the function ON THE CURRENT DEVICE, and the multi-device Monte-Carlo
    // driver launches from one thread per device)
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&scl_decode_llr_kernel<GS, 3, 1, ED, 0, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;        // (the host checks `lds` against the device's limit before it chooses this kernel)
    hipLaunchKernelGGL((scl_decode_llr_kernel<GS, 3, 1, ED, 0, 1>), dim3(blocks), dim3(64), lds, st, p);
    return hipGetLastError();
}
#if POLAR_ED_TU == 3
hipError_t polar_launch_decode_lat_ed1(const PolarDecodeParams &p, int gs, int blocks, hipStream_t st) {
    switch (gs) {
        case 2: return launch_lat<2, true>(p, blocks, st);
        case 4: return launch_lat<4, true>(p, blocks, st);
        case 8: return launch_lat<8, true>(p, blocks, st);
        default: return hipErrorInvalidValue;
    }
}
#elif POLAR_ED_TU == 0
hipError_t polar_launch_decode_lat_ed1(const PolarDecodeParams &p, int gs, int blocks, hipStream_t st);
hipError_t polar_launch_decode_lat(const PolarDecodeParams &p, int gs, bool ed, int blocks, hipStream_t st) {
    if (ed) return polar_launch_decode_lat_ed1(p, gs, blocks, st);
    if (gs == 2) return launch_lat<2, false>(p, blocks, st);
    return hipErrorInvalidValue;
}
size_t polar_decode_lat_lds_bytes(int N, int gs, int W) {
    const int cwords = (N >= 128) ? (N / 32 - 2) : 0;
    // tables + exchange buffers + converted channel + layers + partial-sum and history words (one column per path)
    return 324 * 8 + (128 * 8 + 128) + (size_t)N * 8 + (size_t)N * gs * 8 + ((size_t)2 * cwords + (size_t)3 * W) * gs * 4 + 64;
}
#endif

#if POLAR_ED_TU == 2
hipError_t polar_launch_decode_llr_ed1_gs32(const PolarDecodeParams &p, int lds_log, int pipe, int grid, hipStream_t st) {
    return launch_gs<32, true>(p, lds_log, pipe, grid, st);
}
#elif POLAR_ED_TU == 1
hipError_t polar_launch_decode_llr_ed1_gs32(const PolarDecodeParams &p, int lds_log, int pipe, int grid, hipStream_t st);
hipError_t polar_launch_decode_llr_ed1(const PolarDecodeParams &p, int gs, int lds_log, int pipe, int grid, hipStream_t st) {
    switch (gs) {
#ifndef POLAR_DEV_GS32
        case 4: return launch_gs<4, true>(p, lds_log, pipe, grid, st);
        case 8: return launch_gs<8, true>(p, lds_log, pipe, grid, st);
        case 16: return launch_gs<16, true>(p, lds_log, pipe, grid, st);
        case 64: return launch_gs<64, true>(p, lds_log, pipe, grid, st);
#endif
        case 32: return polar_launch_decode_llr_ed1_gs32(p, lds_log, pipe, grid, st);
        default: return hipErrorInvalidValue;
    }
}
#elif POLAR_ED_TU == 0
hipError_t polar_launch_decode_llr_ed0(const PolarDecodeParams &p, int gs, int lds_log, int pipe, int grid, hipStream_t st) {
    switch (gs) {
        case 1: return launch_gs<1, false>(p, lds_log, pipe, grid, st);
#ifndef POLAR_DEV_GS32
        case 2: return launch_gs<2, false>(p, lds_log, pipe, grid, st);
        case 4: return launch_gs<4, false>(p, lds_log, pipe, grid, st);
        case 8: return launch_gs<8, false>(p, lds_log, pipe, grid, st);
        case 16: return launch_gs<16, false>(p, lds_log, pipe, grid, st);
        case 64: return launch_gs<64, false>(p, lds_log, pipe, grid, st);
#endif
        case 32: return launch_gs<32, false>(p, lds_log, pipe, grid, st);
        default: return hipErrorInvalidValue;
    }
}
hipError_t polar_launch_decode_llr(const PolarDecodeParams &p, int gs, int lds_log, int pipe, int grid, bool ed, hipStream_t st) {
    return ed ? polar_launch_decode_llr_ed1(p, gs, lds_log, pipe, grid, st) : polar_launch_decode_llr_ed0(p, gs, lds_log, pipe, grid, st);
}
#endif
