// polar_decode.cpp — decode_scl_llr on the device: the dispatch over the kernel families, the
// device-resident entry points, the probability-domain members of the class surface, encoder / workload generation.
// Reference: PolarCode::decode_scl_llr (PolarCode.cpp:130-148), decode_scl_p1 (:110-128), encode (:60-91).
#include "polar_host.h"

namespace {

// What the per-wave state scratch of a launch may take: at most 24 GiB of the 288 GB, and at most half of what the device has
// free right now plus what the handle already holds for this purpose (a smaller or busy GPU runs fewer persistent waves instead
// of failing with POLAR_E_NOMEM: round-5 advisor).
size_t scratch_budget(const polar_code *h) {
    size_t budget = (size_t)24 << 30, free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) budget = std::min(budget, (free_b + h->d_llr_scr.cap * sizeof(double)) / 2);
    return std::max(budget, (size_t)64 << 20);
}
// the persistent grid of the P1 kernels: `want` waves, each with `per_wave` bytes of scratch — halved until the scratch is there
template <typename Ensure>
int grid_that_fits(const polar_code *h, long want, size_t per_wave, Ensure ensure, int *grid_out) {
    // (the budget is a driver query — hipMemGetInfo takes milliseconds: asked only when the scratch the handle holds is too small)
    long grid = want;
    if ((size_t)want * per_wave > h->d_llr_scr.cap * sizeof(double)) grid = std::max<long>(1, std::min<long>(want, (long)(scratch_budget(h) / per_wave)));
    for (;;) {
        const int rc = ensure((int)grid);
        if (rc != POLAR_E_NOMEM || grid <= 64) { *grid_out = (int)grid; return rc; }
        (void)hipGetLastError();
        grid /= 2;
    }
}

}  // namespace

extern "C" {

// ------------------------------------------------------------------------------------------
int polar_decode_scl_llr_batch_dev(polar_code_t *h, const double *d_llr, long B, int L, uint8_t *d_out,
                                   double *d_pm, void *stream) {
    return polar_decode_scl_llr_batch_dev_ev(h, d_llr, B, L, d_out, d_pm, stream, nullptr, nullptr);
}

// list size 1, small batches: one codeword per wave, whole state in LDS (sc_lat_kernel)
}  // extern "C"

bool polar_host::use_sc_lat(const polar_code *h, long B) {
    return h->n <= polar_sc_lat_max_log() && polar_sc_lat_lds_bytes(h->N, (int)h->sc_lat_ops.size()) <= h->lds_per_block &&
           h->knobs.lat_max_b >= 0 && B <= (h->knobs.lat_max_b ? h->knobs.lat_max_b : 2048);
}

extern "C" {

int polar_decode_scl_llr_batch_dev_ev(polar_code_t *h, const double *d_llr, long B, int L, uint8_t *d_out,
                                      double *d_pm, void *stream, void *ev_start, void *ev_stop) {
    return decode_impl(h, d_llr, 0, B, nullptr, L, d_out, d_pm, stream, ev_start, ev_stop);
}

// B rows are allocated; when n_dev != nullptr only the first min(B, *n_dev) exist (count read on the device)
}  // extern "C"

int polar_host::decode_impl(polar_code *h, const void *d_llr, int llr_f32, long B, const unsigned int *n_dev, int L, uint8_t *d_out,
                            double *d_pm, void *stream, void *ev_start, void *ev_stop, int phase, int *deferred) {
    if (!h || !d_llr || !d_out) return fail(POLAR_E_ARG, "NULL argument");
    if (deferred) *deferred = 0;
    if (L < 1 || L > POLAR_MAX_LIST) return fail(POLAR_E_ARG, "list size %d out of range [1, %d]", L, POLAR_MAX_LIST);
    if (B < 0) return fail(POLAR_E_ARG, "negative batch");
    if (B == 0) return POLAR_OK;
    DevGuard dg_;
    int rc = ensure_device(h, dg_);
    if (rc) return rc;
    const int gs = pow2ceil(L);
    const int G = 64 / gs;
    // two tuned variants: "pipe" (8 waves/CU, S<=16 in LDS, register double-buffering) and the
    // default high-occupancy one (4-wave blocks, S<=8 in LDS, 16 waves/CU)
    int wpc = h->waves_per_cu ? h->waves_per_cu : 16;
    int lds_log_auto = 0;
    // Small batches of the large lists (no one-codeword-per-wave form: their state does not fit the LDS), round 6: when every group
    // of the call is resident at ONE wave per SIMD anyway, fewer and fatter waves — layers up to 32 in LDS, the register-double-
    // buffered form — answer sooner: a lone wave pays a memory round trip per dependent access of an HBM-resident layer
    // (profiles/r06/small_batch_geometry.txt: L = 32 B = 256 5.10 -> 4.48 ms, L = 16 B = 1024 5.07 -> 4.64; nothing at 4096).
    if (!h->waves_per_cu && !h->lds_log && gs >= 16 && (B + G - 1) / G <= (long)h->num_cu * 4 &&
        polar_decode_lds_bytes(5, 1) <= h->lds_per_block) { wpc = 4; lds_log_auto = 5; }
    const int pipe = (wpc > 8) ? 0 : 1;
    int lds_log = h->lds_log ? h->lds_log : (lds_log_auto ? lds_log_auto : (pipe ? 4 : 3));
    const int wpb = polar_decode_waves_per_block(pipe);
    const size_t lds = polar_decode_lds_bytes(lds_log, pipe);
    const int max_blocks_by_lds = (int)(h->lds_per_block / lds);
    if (max_blocks_by_lds < 1) return fail(POLAR_E_ARG, "lds_log %d does not fit the LDS", lds_log);
    if (wpc > max_blocks_by_lds * wpb) wpc = max_blocks_by_lds * wpb;
    long groups = (B + G - 1) / G;
    long maxgrid = (long)h->num_cu * wpc;
    int grid = (int)std::min(groups, maxgrid);
    grid = ((grid + wpb - 1) / wpb) * wpb;          // whole blocks
    const int SL = 1 << lds_log;
    const size_t big = (h->N > 2 * SL) ? (size_t)(h->N - 2 * SL) : 0;
    const size_t cwords = (h->N >= 128) ? (size_t)(h->N / 32 - 2) : 0;
    if ((size_t)grid * big * 64 + 64 > h->d_llr_scr.cap) {
        // the per-wave state scratch is big*512 B per resident wave (1 MiB at N=2048, 16 MiB at N=32768): when it has to grow it is
        // capped (scratch_budget: 24 GiB, half of the free memory) by running fewer persistent waves
        const size_t per_wave = big * 64 * sizeof(double) + 1;
        const long cap = (long)(scratch_budget(h) / per_wave);
        if (grid > cap) grid = (int)std::max<long>(wpb, (cap / wpb) * wpb);
    }
    if ((rc = h->d_llr_scr.ensure((size_t)grid * big * 64 + 64))) return rc;
    if ((rc = h->d_c_scr.ensure((size_t)grid * 2 * cwords * 64 + 64))) return rc;
    if ((rc = h->d_hist_scr.ensure((size_t)grid * 3 * h->W * 64 + 64))) return rc;
    PolarDecodeParams p;
    p.n = h->n; p.N = h->N; p.K = h->K; p.crc = h->crc; p.L = L; p.W = h->W; p.B = B;
    {   // all-frozen prefix [0, P): handled cooperatively by the kernel when one codeword owns 32 lanes
        int P = 0;
        while (P < h->N && h->frozen[P]) ++P;
        int Q = 0;
        if (gs >= 4 && h->prefix_on) {
            if (P >= 256) Q = 256;
            else { Q = 64; while (Q <= P) Q <<= 1; if (P < 33) Q = 0; }
            if (Q > h->N / 2) Q = 0;
        }
        p.prefix_q = Q;
        p.prefix_len = Q ? std::min(P, Q) : 0;
    }
    p.llr = (const double *)d_llr; p.llr_f32 = llr_f32; p.p0 = nullptr; p.out = d_out; p.pm_out = d_pm;
    p.frozen = h->d_frozen.p; p.info_rank = h->d_info_rank.p; p.crc_mask = h->d_crc_mask.p; p.tabs = h->d_tabs.p;
    p.ctl = h->d_ctl.p;
    p.pre = nullptr;
    p.flags = nullptr; p.cw_list = nullptr; p.cw_count = nullptr; p.n_dev = n_dev;
    p.tab_scr = nullptr; p.var_scr = nullptr;
    if (p.prefix_q) {
        if ((rc = h->d_pre.ensure((size_t)B * (size_t)(h->N - p.prefix_q + 1)))) return rc;
        p.pre = h->d_pre.p;
    }
    p.llr_scr = h->d_llr_scr.p; p.c_scr = h->d_c_scr.p; p.hist_scr = h->d_hist_scr.p;
    if ((rc = h->d_work.ensure(1))) return rc;
    p.work = h->d_work.p;
    hipStream_t st = (hipStream_t)stream;
    // Node arithmetic: exp-domain kernel (one division per f-node instead of four transcendentals) for the
    // list sizes where the f-node dominates; codewords it flags (decisions within 1e-10 of the |x| < 40
    // test, degenerate inputs) are decoded again by the LLR-domain kernel in a fallback pass over a
    // device-side work list: no host synchronisation, normally zero entries.
    const int mode = h->knobs.mode_override >= 0 ? h->knobs.mode_override : h->mode;
    if (L == 1 && mode != 1 && !d_pm) {          // (a requested path metric needs the general kernel: this one has none)
        // ---- list size 1: pruned successive cancellation, eight lanes per codeword (polar_kernels_sc.hip); flagged
        // codewords (degenerate inputs, |x| < 40 decisions too close to call) go through the general kernel below
        const long groups8 = (B + 7) / 8;
        // (measured and dropped: as many waves as make the rounds of eight-codeword groups whole — 4 096 instead of 5 120 for
        // 65 536 codewords — is 2.5 % SLOWER: the kernel wants the latency hiding of 20 waves per CU more than a full last round)
        const int sgrid = (int)std::min<long>(groups8, (long)h->num_cu * polar_sc8_waves_per_cu(h->N));
        // (the in-place reads are 16-byte vector loads: a caller's pointer that is not 16-byte aligned takes the front pass; the knob:
        // A/B measurements and the parity tests of both paths)
        const bool fold = h->sc_fold && !h->knobs.sc_no_fold && ((uintptr_t)d_llr & 15u) == 0;
        // small batches: one codeword per wave, whole state in LDS (sc_lat_kernel: a lone wave of the eight-codeword kernel pays
        // a memory round trip per dependent access of its HBM-resident layers — B = 1: 0.85 ms against 0.33 ms on a host core)
        const bool lat = use_sc_lat(h, B);
        if (!fold && !lat && (rc = h->d_ech.ensure((size_t)B * h->N))) return rc;
        if ((rc = h->d_list.ensure((size_t)B))) return rc;
        // control words and flag words in ONE buffer, zeroed by ONE memset: [0] work counter of the decode kernel, [1] length of
        // the fallback work list, [2] work counter of the fallback pass, [4 ...] one flag bit per codeword (round 3: four
        // memsets and two kernels — bits -> bytes -> list — around the decode kernel; a step at batch 65536 is 3.3 ms)
        const size_t nfw = (size_t)(B + 31) / 32 + 1;
        if ((rc = h->d_flag_words.ensure(4 + nfw))) return rc;
        unsigned int *ctrl = h->d_flag_words.p, *fwords = h->d_flag_words.p + 4;
        // (the alpha scratch is shared with the general kernel's, which the fallback pass uses)
        if ((rc = h->d_llr_scr.ensure(std::max((size_t)sgrid * polar_sc8_scratch_doubles_per_wave(h->N) + 64, (size_t)grid * big * 64 + 64)))) return rc;
        p.llr_scr = h->d_llr_scr.p;
        if (phase != 2) {
            HIP_TRY(hipMemsetAsync(ctrl, 0, (4 + nfw) * sizeof(unsigned int), st));
            if (!fold && !lat) HIP_TRY(polar_launch_sc8_front(d_llr, llr_f32, h->d_ech.p, fwords, h->d_tabs.p, h->n, B, n_dev, st));
            PolarScParams sp;
            sp.n = h->n; sp.N = h->N; sp.K = h->K; sp.B = B;
            sp.llr = (fold || lat) ? d_llr : nullptr; sp.llr_f32 = llr_f32;
            sp.ech_t = (fold || lat) ? nullptr : h->d_ech.p; sp.out = d_out; sp.ops = h->d_sc_ops.p; sp.n_ops = (int)h->sc_ops.size();
            sp.order = h->d_order.p; sp.tabs = h->d_tabs.p; sp.a_scr = h->d_llr_scr.p;
            sp.flag_words = fwords; sp.work = ctrl; sp.n_dev = n_dev;
            sp.flag_bytes = lat ? h->lat_flag_bytes : nullptr;
            if (ev_start) HIP_TRY(hipEventRecord((hipEvent_t)ev_start, st));
            if (lat) { sp.ops = h->d_sc_lat_ops.p; sp.n_ops = (int)h->sc_lat_ops.size(); }
            if (lat) HIP_TRY(polar_launch_sc_lat(sp, (int)std::min<long>(B, (long)h->num_cu * 4), st));
            else HIP_TRY(polar_launch_sc8_decode(sp, sgrid, st));
            if (ev_stop) HIP_TRY(hipEventRecord((hipEvent_t)ev_stop, st));
            if (phase == 1 && lat && deferred) { *deferred = 1; return POLAR_OK; }
        }
        HIP_TRY(polar_launch_sc_collect(fwords, B, n_dev, h->d_list.p, ctrl + 1, st));
        PolarDecodeParams pf = p;
        pf.prefix_q = 0; pf.prefix_len = 0; pf.pre = nullptr;
        pf.work = ctrl + 2;
        pf.cw_list = h->d_list.p; pf.cw_count = ctrl + 1; pf.n_dev = nullptr;
        HIP_TRY(polar_launch_decode_llr(pf, gs, lds_log, pipe, std::min(grid, 16 * wpb), false, st));
        return POLAR_OK;
    }
    // (the exp-domain kernels exist for groups of 4 lanes and more: smaller lists take the LLR-domain kernel in every mode)
    // (round 3: automatic mode takes the exp-domain kernel from lists of 3 on — it was 5: with the block-placement hints the
    // 4-lane groups run 16 % faster on it, config 3: 4.4 -> 5.1 M cw/s)
    const bool ed = ((mode == 2) || (mode == 0 && gs >= 4)) && gs >= 4;
    // Small batches of the small lists: ONE codeword per wave, its elements spread over the 64 / gs lanes of each path, the state in
    // LDS (scl_decode_llr_kernel<.., LAT = 1>; exp-domain arithmetic for groups of 4 and 8 lanes, LLR-domain for groups of 2). The
    // kernel converts the channel itself (no conversion pass, no prefix kernel).
    // (groups of 2 lanes: the batch path is the LLR-domain kernel, but ONE codeword per wave is faster with the exp-domain nodes — 2.2
    // against 2.9 ms — so the latency form takes them unless mode 1 forces the LLR-domain arithmetic)
    const bool lat_ed = (gs == 2) ? (mode != 1) : ed;
    const size_t lat_lds = polar_decode_lat_lds_bytes(h->N, gs, h->W);
    const long lat_resident = lat_lds <= h->lds_per_block ? (long)h->num_cu * std::min<long>(4, (long)(h->lds_per_block / lat_lds)) : 0;   // waves the LDS lets a device hold
    const bool lat_list = (gs == 2 || (ed && (gs == 4 || gs == 8))) && h->knobs.lat_max_b >= 0 && lat_resident > 0 &&
                          B <= (h->knobs.lat_max_b ? h->knobs.lat_max_b : 2 * lat_resident);
    // (measured, N = 2048, round 6 — profiles/r06/latency_table_248.json: L = 4 B = 1 ... 256 1.80 ... 1.94 ms against 3.80 ... 4.36 ms
    // for the batch kernel, L = 2 1.73 ... 2.09 against 5.9 ... 7.0 ms, L = 8 2.02 ... 2.17 against 3.83 ... 4.40. The LDS lets the
    // device hold one wave per CU for lists of 4 and 8 at N = 2048, three for lists of 2; TWO rounds of that are still faster than
    // the batch kernel — B = 512: 3.80 against 4.46 ms at L = 4, 4.25 against 4.59 at L = 8; B = 1024: 4.10 against 7.24 at L = 2 —,
    // three are not: that is the default threshold)
    if (lat_list) {
        PolarDecodeParams pl = p;
        pl.prefix_q = 0; pl.prefix_len = 0; pl.pre = nullptr;
        const int blocks = (int)std::min<long>(B, lat_resident);
        if (lat_ed) {
            // (sized for the largest batch this path ever takes — a few hundred entries — so that the first call reserves it)
            const size_t cap = (size_t)std::max<long>(B, 2 * lat_resident);
            if ((rc = h->d_flags.ensure(cap))) return rc;
            if ((rc = h->d_list.ensure(cap))) return rc;
            if ((rc = h->d_count.ensure(1))) return rc;
            pl.flags = h->d_flags.p;
        }
        if (phase != 2) {
            HIP_TRY(hipMemsetAsync(p.work, 0, sizeof(unsigned int), st));
            if (ev_start) HIP_TRY(hipEventRecord((hipEvent_t)ev_start, st));
            HIP_TRY(polar_launch_decode_lat(pl, gs, lat_ed, blocks, st));
            if (ev_stop) HIP_TRY(hipEventRecord((hipEvent_t)ev_stop, st));
            if (!lat_ed) return POLAR_OK;
            if (phase == 1 && deferred) { *deferred = 2; return POLAR_OK; }        // (flag BYTES in d_flags: the caller looks)
        } else if (!lat_ed) return POLAR_OK;
        HIP_TRY(hipMemsetAsync(h->d_count.p, 0, sizeof(unsigned int), st));
        HIP_TRY(polar_launch_ed_collect(h->d_flags.p, B, n_dev, h->d_list.p, h->d_count.p, st));
        HIP_TRY(hipMemsetAsync(p.work, 0, sizeof(unsigned int), st));
        PolarDecodeParams pf = p;
        pf.prefix_q = 0; pf.prefix_len = 0; pf.pre = nullptr;
        pf.cw_list = h->d_list.p; pf.cw_count = h->d_count.p; pf.n_dev = nullptr;
        HIP_TRY(polar_launch_decode_llr(pf, gs, lds_log, pipe, std::min(grid, 64 * wpb), false, st));
        return POLAR_OK;
    }
    HIP_TRY(hipMemsetAsync(p.work, 0, sizeof(unsigned int), st));
    if (!ed) {
        if (p.prefix_q) HIP_TRY(polar_launch_prefix(p, false, nullptr, st));
        if (ev_start) HIP_TRY(hipEventRecord((hipEvent_t)ev_start, st));
        HIP_TRY(polar_launch_decode_llr(p, gs, lds_log, pipe, grid, false, st));
        if (ev_stop) HIP_TRY(hipEventRecord((hipEvent_t)ev_stop, st));
        return POLAR_OK;
    }
    if ((rc = h->d_ech.ensure((size_t)B * h->N))) return rc;
    if ((rc = h->d_flags.ensure((size_t)B))) return rc;
    if ((rc = h->d_list.ensure((size_t)B))) return rc;
    if ((rc = h->d_count.ensure(1))) return rc;
    HIP_TRY(hipMemsetAsync(h->d_count.p, 0, sizeof(unsigned int), st));
    // (round 4: where the prefix kernel's first pass is staged through LDS it converts the raw rows itself — no conversion pass)
    const bool fuse_front = p.prefix_q > 0 && polar_prefix_is_staged(h->N) && !h->knobs.no_fuse_front;
    if (!fuse_front) HIP_TRY(polar_launch_ed_front(d_llr, llr_f32, h->d_ech.p, h->d_flags.p, h->d_tabs.p, h->N, B, n_dev, st));
    PolarDecodeParams pe = p;
    pe.llr = h->d_ech.p; pe.llr_f32 = 0; pe.flags = h->d_flags.p;
    if (gs == 32 && !pipe && h->N >= 1024 && p.prefix_q > 0 && !h->knobs.no_tables) {
        // table mode: layers 1 and 2 as per-codeword value tables (polar_kernels.hip)
        if ((rc = h->d_tab_scr.ensure((size_t)grid * G * 3 * h->N + 64))) return rc;
        if ((rc = h->d_var_scr.ensure((size_t)grid * (h->N / 32) * 64 + 64))) return rc;
        pe.tab_scr = h->d_tab_scr.p; pe.var_scr = h->d_var_scr.p;
    }
    if (pe.prefix_q && fuse_front) {
        PolarDecodeParams pp = pe;
        pp.llr = (const double *)d_llr; pp.llr_f32 = llr_f32;
        HIP_TRY(polar_launch_prefix(pp, true, h->d_ech.p, st));
    } else if (pe.prefix_q) HIP_TRY(polar_launch_prefix(pe, true, nullptr, st));
    if (ev_start) HIP_TRY(hipEventRecord((hipEvent_t)ev_start, st));
    HIP_TRY(polar_launch_decode_llr(pe, gs, lds_log, pipe, grid, true, st));
    if (ev_stop) HIP_TRY(hipEventRecord((hipEvent_t)ev_stop, st));
    // fallback pass (LLR-domain kernel, no prefix kernel) over the flagged codewords
    HIP_TRY(polar_launch_ed_collect(h->d_flags.p, B, n_dev, h->d_list.p, h->d_count.p, st));
    HIP_TRY(hipMemsetAsync(p.work, 0, sizeof(unsigned int), st));
    PolarDecodeParams pf = p;
    pf.prefix_q = 0; pf.prefix_len = 0; pf.pre = nullptr;
    pf.cw_list = h->d_list.p; pf.cw_count = h->d_count.p; pf.n_dev = nullptr;
    // (normally empty: a few blocks; a code with weak unfrozen leaves may send most of its codewords here)
    const int fgrid = h->weak_leaves ? grid : std::min(grid, 64 * wpb);
    HIP_TRY(polar_launch_decode_llr(pf, gs, lds_log, pipe, fgrid, false, st));
    return POLAR_OK;
}

extern "C" {

// single-precision LLRs at the boundary: every float is widened (exactly) in the load stage of the first kernel that
// touches the channel values (ed_front_kernel / prefix_kernel / the layer-1 visits) — no staging copy
int polar_decode_scl_llr_batch_dev_f32(polar_code_t *h, const float *d_llr, long B, int L, uint8_t *d_out,
                                       double *d_pm, void *stream) {
    return decode_impl(h, d_llr, 1, B, nullptr, L, d_out, d_pm, stream, nullptr, nullptr);
}

// PolarCode::decode_scl_p1 (PolarCode.cpp:110-128): probability-domain SCL
int polar_decode_scl_p1_batch(polar_code_t *h, const double *p1, const double *p0, long B, int L, uint8_t *out) {
    if (!h || !p1 || !p0 || !out) return fail(POLAR_E_ARG, "NULL argument");
    if (L < 1 || L > POLAR_MAX_LIST) return fail(POLAR_E_ARG, "list size %d out of range [1, %d]", L, POLAR_MAX_LIST);
    if (B < 0) return fail(POLAR_E_ARG, "negative batch");
    if (B == 0) return POLAR_OK;
    DevGuard dg_;
    int rc = ensure_device(h, dg_);
    if (rc) return rc;
    const int N = h->N;
    const int gs = pow2ceil(L), G = 64 / gs;
    long groups = (B + G - 1) / G;
    // (one wave per block, the whole state in a per-wave scratch of 2 N rows: 16 waves per CU hide its latency — round 4 launched 4 —
    // as long as the scratch of all of them stays below 24 GiB)
    const size_t cwords = (N >= 128) ? (size_t)(N / 32 - 2) : 0;
    if ((rc = h->d_in.ensure((size_t)B * N * 2))) return rc;
    if ((rc = h->d_out.ensure((size_t)B * h->K))) return rc;
    int grid = 1;
    rc = grid_that_fits(h, std::min<long>(groups, (long)h->num_cu * 16), (size_t)N * 64 * 2 * sizeof(double), [&](int g) {
        int r = h->d_llr_scr.ensure((size_t)g * N * 64 * 2 + 64);
        if (!r) r = h->d_c_scr.ensure((size_t)g * 2 * cwords * 64 + 64);
        if (!r) r = h->d_hist_scr.ensure((size_t)g * h->W * 64 + 64);
        return r;
    }, &grid);
    if (rc) return rc;
    HIP_TRY(hipMemcpy(h->d_in.p, p1, (size_t)B * N * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(h->d_in.p + (size_t)B * N, p0, (size_t)B * N * sizeof(double), hipMemcpyHostToDevice));
    PolarDecodeParams p;
    p.n = h->n; p.N = N; p.K = h->K; p.crc = h->crc; p.L = L; p.W = h->W; p.B = B;
    p.prefix_q = 0; p.prefix_len = 0; p.ctl = nullptr; p.pre = nullptr; p.work = nullptr;
    p.llr = h->d_in.p; p.llr_f32 = 0; p.p0 = h->d_in.p + (size_t)B * N; p.out = h->d_out.p; p.pm_out = nullptr;
    p.frozen = h->d_frozen.p; p.info_rank = h->d_info_rank.p; p.crc_mask = h->d_crc_mask.p; p.tabs = h->d_tabs.p;
    p.llr_scr = h->d_llr_scr.p; p.c_scr = h->d_c_scr.p; p.hist_scr = h->d_hist_scr.p;
    p.flags = nullptr; p.cw_list = nullptr; p.cw_count = nullptr; p.n_dev = nullptr; p.tab_scr = nullptr; p.var_scr = nullptr;
    HIP_TRY(polar_launch_decode_p1(p, gs, grid, nullptr));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out, h->d_out.p, (size_t)B * h->K, hipMemcpyDeviceToHost));
    return POLAR_OK;
}
int polar_decode_scl_p1(polar_code_t *h, const double *p1, const double *p0, int L, uint8_t *out) {
    return polar_decode_scl_p1_batch(h, p1, p0, 1, L, out);
}

// PolarM decode_sc_p1 (PolarCode.m:290-295): out are doubles like MATLAB's (0.5 when a leaf is exactly 0.5)
int polar_decode_sc_p1_batch(polar_code_t *h, const double *p1, long B, double *out) {
    if (!h || !p1 || !out) return fail(POLAR_E_ARG, "NULL argument");
    if (B < 0) return fail(POLAR_E_ARG, "negative batch");
    if (B == 0) return POLAR_OK;
    DevGuard dg_;
    int rc = ensure_device(h, dg_);
    if (rc) return rc;
    const int N = h->N;
    if ((rc = h->d_in.ensure((size_t)B * N + (size_t)B * h->K))) return rc;
    PolarScP1Params p;
    p.n = h->n; p.N = N; p.K = h->K; p.B = B;
    p.p1 = h->d_in.p; p.out = h->d_in.p + (size_t)B * N;
    p.frozen = h->d_frozen.p; p.order = h->d_order.p; p.scr = nullptr;
    HIP_TRY(hipMemcpy(h->d_in.p, p1, (size_t)B * N * sizeof(double), hipMemcpyHostToDevice));
    // Small batches (PolarM calls this once per codeword, main_MC_CC_Comparison.m:96): one codeword per WAVE, state in LDS; from
    // about one codeword per lane of the waves the device holds the lane-per-codeword kernel wins (its 64 codewords per wave
    // share every instruction). Same doubles either way.
    const long lat_waves = polar_sc_p1_lat_lds_bytes(N) <= h->lds_per_block ? (long)h->num_cu * std::max<long>(1, (long)(h->lds_per_block / polar_sc_p1_lat_lds_bytes(N))) : 0;
    const long lat_max = h->knobs.lat_max_b < 0 ? 0 : (h->knobs.lat_max_b ? h->knobs.lat_max_b : lat_waves * 4);
    if (lat_waves > 0 && B <= lat_max) {
        HIP_TRY(polar_launch_sc_p1_lat(p, (int)std::min<long>(B, lat_waves), nullptr));
    } else {
        int grid = 1;
        rc = grid_that_fits(h, std::min<long>((B + 63) / 64, (long)h->num_cu * 16), (size_t)N * 64 * 4 * sizeof(double),
                            [&](int g) { return h->d_llr_scr.ensure((size_t)g * 4 * N * 64 + 64); }, &grid);
        if (rc) return rc;
        p.scr = h->d_llr_scr.p;
        HIP_TRY(polar_launch_sc_p1(p, grid, nullptr));
    }
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out, p.out, (size_t)B * h->K * sizeof(double), hipMemcpyDeviceToHost));
    return POLAR_OK;
}
int polar_decode_sc_p1(polar_code_t *h, const double *p1, double *out) { return polar_decode_sc_p1_batch(h, p1, 1, out); }

// ------------------------------------------------------------------------------------------
}  // extern "C"

void polar_host::fill_enc(const polar_code *h, PolarEncodeParams &p) {
    memset(&p, 0, sizeof p);
    p.n = h->n; p.N = h->N; p.K = h->K; p.crc = h->crc;
    p.order = h->d_order.p; p.crcm = h->d_crcm.p;
    p.stride = 1;
    p.info_block_div = 100;
}

extern "C" {

int polar_encode_batch_dev(polar_code_t *h, const uint8_t *d_info, long B, uint8_t *d_coded, void *stream) {
    if (!h || !d_info || !d_coded) return fail(POLAR_E_ARG, "NULL argument");
    if (B <= 0) return B == 0 ? POLAR_OK : fail(POLAR_E_ARG, "negative batch");
    DevGuard dg_;
    int rc = ensure_device(h, dg_);
    if (rc) return rc;
    PolarEncodeParams p;
    fill_enc(h, p);
    p.B = B; p.info = d_info; p.coded = d_coded;
    HIP_TRY(polar_launch_encode(p, (hipStream_t)stream));
    return POLAR_OK;
}

int polar_encode_batch(polar_code_t *h, const uint8_t *info, long B, uint8_t *coded) {
    if (!h || !info || !coded) return fail(POLAR_E_ARG, "NULL argument");
    if (B <= 0) return B == 0 ? POLAR_OK : fail(POLAR_E_ARG, "negative batch");
    DevGuard dg_;
    int rc = ensure_device(h, dg_);
    if (rc) return rc;
    if ((rc = h->d_bytes_a.ensure((size_t)B * h->K))) return rc;
    if ((rc = h->d_bytes_b.ensure((size_t)B * h->N))) return rc;
    HIP_TRY(hipMemcpy(h->d_bytes_a.p, info, (size_t)B * h->K, hipMemcpyHostToDevice));
    if ((rc = polar_encode_batch_dev(h, h->d_bytes_a.p, B, h->d_bytes_b.p, nullptr))) return rc;
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(coded, h->d_bytes_b.p, (size_t)B * h->N, hipMemcpyDeviceToHost));
    return POLAR_OK;
}
int polar_encode(polar_code_t *h, const uint8_t *info, uint8_t *coded) { return polar_encode_batch(h, info, 1, coded); }

int polar_synth_llr_dev(polar_code_t *h, uint64_t seed, uint64_t trial0, long B, double s,
                        double *d_llr, uint8_t *d_info, void *stream) {
    if (!h || !d_llr) return fail(POLAR_E_ARG, "NULL argument");
    if (B <= 0) return B == 0 ? POLAR_OK : fail(POLAR_E_ARG, "negative batch");
    DevGuard dg_;
    int rc = ensure_device(h, dg_);
    if (rc) return rc;
    PolarEncodeParams p;
    fill_enc(h, p);
    p.B = B; p.seed = seed; p.trial0 = trial0; p.s = s; p.llr = d_llr; p.info_out = d_info;
    HIP_TRY(polar_launch_synth(p, (hipStream_t)stream));
    return POLAR_OK;
}

// Pre-size every device scratch buffer decodes of up to B codewords at list sizes 1 .. L need, by running one decode per
// kernel family on generated inputs (list size 1: the pruned SC kernel and its flag words; 2: the LLR-domain kernel's 2-lane
// groups; every power-of-two lane group up to pow2ceil(L), with and without the path-metric output): afterwards
// polar_decode_scl_llr_batch_dev* calls within (B, L) allocate nothing (no hipFree / hipMalloc, i.e. no implicit device
// synchronisation, inside the nominally asynchronous calls; polar_debug_get "allocs" counts them).
int polar_reserve(polar_code_t *h, long B, int L) {
    if (!h) return fail(POLAR_E_ARG, "NULL handle");
    if (L < 1 || L > POLAR_MAX_LIST) return fail(POLAR_E_ARG, "list size %d out of range [1, %d]", L, POLAR_MAX_LIST);
    if (B <= 0) return B == 0 ? POLAR_OK : fail(POLAR_E_ARG, "negative batch");
    DevGuard dg_;
    int rc = ensure_device(h, dg_);
    if (rc) return rc;
    DevBuf<double> llr, pm;
    DevBuf<uint8_t> out;
    if ((rc = llr.ensure((size_t)B * h->N + 1)) || (rc = out.ensure((size_t)B * h->K)) || (rc = pm.ensure((size_t)B))) { llr.release(); out.release(); pm.release(); return rc; }
    rc = polar_synth_llr_dev(h, 1, 0, B, polar_snr_sqrt_linear(h, 2.0), llr.p, nullptr, nullptr);
    const int top = std::min(pow2ceil(L), POLAR_MAX_LIST);
    // every kernel family a call within (B, L) can reach: per list size the batch kernel at B and the one-codeword-per-wave
    // kernel at one codeword (its flag / work-list buffers are its own: polar_reserve(B, 2) above the latency threshold used
    // to leave them to the first small call), list size 1 also with a requested metric (the general kernel) and from rows
    // that are NOT 16-byte aligned (the converted copy the in-place reads cannot serve)
    for (int l = 1; l <= top && !rc; l <<= 1) {
        rc = polar_decode_scl_llr_batch_dev(h, llr.p, B, l, out.p, nullptr, nullptr);
        if (!rc && l <= 8) rc = polar_decode_scl_llr_batch_dev(h, llr.p, 1, l, out.p, nullptr, nullptr);
        if (!rc && l == 1) rc = polar_decode_scl_llr_batch_dev(h, llr.p, B, l, out.p, pm.p, nullptr);
        if (!rc && l == 1) rc = polar_decode_scl_llr_batch_dev(h, llr.p + 1, B, l, out.p, nullptr, nullptr);
    }
    hipError_t e = hipDeviceSynchronize();
    llr.release(); out.release(); pm.release();
    if (!rc && e != hipSuccess) return fail(POLAR_E_DEVICE, "polar_reserve: %s", hipGetErrorString(e));
    return rc;
}

int polar_count_errors_dev(polar_code_t *h, const uint8_t *d_a, const uint8_t *d_b, long B,
                           unsigned long long *d_err_count, void *stream) {
    if (!h || !d_a || !d_b || !d_err_count) return fail(POLAR_E_ARG, "NULL argument");
    if (B <= 0) return B == 0 ? POLAR_OK : fail(POLAR_E_ARG, "negative batch");
    DevGuard dg_;
    int rc = ensure_device(h, dg_);
    if (rc) return rc;
    HIP_TRY(polar_launch_count_errors(d_a, d_b, B, h->K, d_err_count, nullptr, (hipStream_t)stream));
    return POLAR_OK;
}

}  // extern "C"
