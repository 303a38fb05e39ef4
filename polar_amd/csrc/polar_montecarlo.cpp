// polar_montecarlo.cpp — PolarCode::get_bler_quick (PolarCode.cpp:658-785; PolarM/PolarCode.m:781-850,
// main_MC_CC_Comparison.m:44-119) on 1..n GPUs: device-side rounds (generation, encoder, channel, decode, counting), the
// pipelined-round scheduler, and the Monte-Carlo code construction of PolarM (PolarCode.m:143-196). DESIGN.md §6.
#include "polar_multi.h"

extern "C" {

// ---- Monte-Carlo (PolarCode::get_bler_quick, PolarCode.cpp:658-785) -----------------------
static void fill_channel(const polar_code *h, PolarEncodeParams &p, int constellation, double snr_point) {
    p.constellation = constellation;
    if (constellation == 0) {
        p.s = polar_snr_sqrt_linear(h, snr_point);           // Eb/N0 in dB, PolarCode.cpp:744-745
        p.info_block_div = 100;
    } else {
        // main_MC_CC_Comparison.m:88-92: sigma = sqrt(1/2) * 10^(-snr_db/20), n0 = sigma^2
        p.sigma = std::sqrt(1.0 / 2) * std::pow(10.0, -snr_point / 20);
        p.n0 = p.sigma * p.sigma;
        p.cnorm = polar_const_norm(constellation);
        p.info_block_div = 1;                                // fresh info every run (:50)
    }
}

// One Monte-Carlo round of T trials {t0 + i*stride} for every enabled (L, Eb/N0) point, entirely stream-ordered on
// the device: per list size the alive list starts with all T trials; per point: synth(alive) -> decode -> count
// block/bit errors and append the failing trials to the next alive list (PolarCode.cpp:728-742: a trial decoded at a
// lower Eb/N0 is counted as run, not simulated). No host round trip between the points; the counters of the round
// ([P][2] block errors, bit errors) stay in h->d_mc_ctr until mc_round_collect().
static int mc_round_launch(polar_code_t *h, int constellation, uint64_t seed, uint64_t t0, long T, long stride,
                           const double *ebno, int n_e, const uint8_t *Ls, int n_L, const uint8_t *enabled, hipStream_t st) {
    const int N = h->N, K = h->K, P = n_e * n_L;
    int rc;
    if ((rc = h->d_in.ensure((size_t)T * N))) return rc;
    if ((rc = h->d_out.ensure((size_t)T * K))) return rc;
    if ((rc = h->d_bytes_a.ensure((size_t)T * K))) return rc;      // sent info
    if ((rc = h->d_alive[0].ensure((size_t)T))) return rc;
    if ((rc = h->d_alive[1].ensure((size_t)T))) return rc;
    if ((rc = h->d_nalive.ensure(2))) return rc;
    if ((rc = h->d_mc_ctr.ensure((size_t)2 * P))) return rc;
    HIP_TRY(hipMemsetAsync(h->d_mc_ctr.p, 0, (size_t)2 * P * sizeof(unsigned long long), st));
    for (int li = 0; li < n_L; ++li) {
        int cur = 0;
        bool first = true;
        for (int ie = 0; ie < n_e; ++ie) {
            if (!enabled[li * n_e + ie]) continue;                     // :725
            if (first) {
                HIP_TRY(polar_launch_mc_init_alive(h->d_alive[0].p, h->d_nalive.p, t0, stride, T, st));
                cur = 0; first = false;
            }
            const int nxt = cur ^ 1;
            HIP_TRY(hipMemsetAsync(h->d_nalive.p + nxt, 0, sizeof(unsigned int), st));
            PolarEncodeParams p;
            fill_enc(h, p);
            p.B = T; p.seed = seed; p.sel = h->d_alive[cur].p; p.n_dev = h->d_nalive.p + cur;
            fill_channel(h, p, constellation, ebno[ie]);
            p.llr = h->d_in.p; p.info_out = h->d_bytes_a.p;
            HIP_TRY(polar_launch_synth(p, st));
            if ((rc = decode_impl(h, h->d_in.p, 0, T, h->d_nalive.p + cur, Ls[li], h->d_out.p, nullptr, st, nullptr, nullptr))) return rc;
            HIP_TRY(polar_launch_mc_count_compact(h->d_out.p, h->d_bytes_a.p, T, K, h->d_alive[cur].p, h->d_nalive.p + cur,
                                                  h->d_alive[nxt].p, h->d_nalive.p + nxt, h->d_mc_ctr.p + 2 * (size_t)(li * n_e + ie), st));
            cur = nxt;
        }
    }
    return POLAR_OK;
}
// the counters of the last round -> host accumulators (err, bit_err may be NULL); run += T for every enabled point (:728)
static int mc_round_collect(polar_code_t *h, long T, int P, const uint8_t *enabled, uint64_t *err, uint64_t *bit_err, uint64_t *run, hipStream_t st) {
    std::vector<unsigned long long> c((size_t)2 * P);
    HIP_TRY(hipMemcpyAsync(c.data(), h->d_mc_ctr.p, c.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    for (int i = 0; i < P; ++i) {
        if (!enabled[i]) continue;
        if (err) err[i] += (uint64_t)c[2 * i];
        if (bit_err) bit_err[i] += (uint64_t)c[2 * i + 1];
        if (run) run[i] += (uint64_t)T;
    }
    return POLAR_OK;
}

// ---- pipelined rounds (round 5) -------------------------------------------------------------------------------------
// Within a round the Eb/N0 points depend on each other (a point simulates the trials that FAILED at the point before:
// PolarCode.cpp:728-742), and beyond the first they are small — 41 000 / 9 600 / 1 400 / 150 of 262 144 trials on BASELINE
// configuration 4's grid — while a launch of the list kernels takes a wave-decode (8 ms at L = 32) however little it carries:
// four under-filled launches with a tail each per round. ACROSS rounds nothing depends on anything, so a step decodes, per
// list size, ONE merged batch: point 1 of the newest round, point 2 of the round before, point 3 of the one before that, ...
// (each stage generated at its own Eb/N0 into its rows of the batch, counted and compacted from them afterwards). The
// host-side schedule (bler_impl) keeps the reference's per-round semantics exactly: whether round r simulates point i is
// decided from point i's errors in the rounds before r, which have all passed point i by then.
struct McStage { int li, ie, slot; long T; uint64_t base; bool fresh; };

static int mc_step_launch(polar_code_t *h, int constellation, uint64_t seed, const std::vector<McStage> &stages, int part, int parts,
                          const double *axis, int n_e, const uint8_t *Ls, int n_L, int n_slots, hipStream_t st) {
    const int N = h->N, K = h->K, P = n_e * n_L;
    int rc;
    if ((rc = h->d_mc_ctr.ensure((size_t)2 * P))) return rc;
    HIP_TRY(hipMemsetAsync(h->d_mc_ctr.p, 0, (size_t)2 * P * sizeof(unsigned long long), st));
    if ((int)h->mc_slots.size() < n_L * n_slots) h->mc_slots.resize((size_t)n_L * n_slots);
    if ((rc = h->d_slot_n.ensure((size_t)n_L * n_slots))) return rc;
    h->h_slot_n.resize((size_t)n_L * n_slots);
    // rows of every list size's merged batch (the buffers are sized once, for the largest: a reallocation between two list sizes
    // would lose the rows they share, below)
    std::vector<long> rows_of(n_L, 0);
    long rows_max = 0;
    for (const McStage &s : stages) {
        polar_code::McSlot &sl = h->mc_slots[(size_t)s.li * n_slots + s.slot];
        if (s.fresh) sl.cnt = (s.T - part + parts - 1) / parts;          // this device's trials of the round: base + part, + parts, ...
        rows_of[s.li] += sl.cnt;
        rows_max = std::max(rows_max, rows_of[s.li]);
    }
    if (rows_max > 0) {
        // (a quarter of headroom when the buffers grow: the first steps of a call carry one round, the later ones the survivors of
        // the rounds before as well — a 4-GiB reallocation in the middle of a sweep is a second lost)
        const size_t cap_rows = (size_t)rows_max * N <= h->d_in.cap ? (size_t)rows_max : (size_t)rows_max + (size_t)rows_max / 4;
        if ((rc = h->d_in.ensure(cap_rows * N))) return rc;
        if ((rc = h->d_out.ensure(cap_rows * K))) return rc;
        if ((rc = h->d_bytes_a.ensure(cap_rows * K))) return rc;      // sent info
    }
    // A trial's LLRs at a point do not depend on the list size (one noise vector per run, shared by every (L, Eb/N0):
    // PolarCode.cpp:708-710), and the first stage of a round simulates ALL its trials: when the list sizes of a sweep start the
    // same round at the same point (the reference's main.cpp: five list sizes), its rows are generated ONCE — they come first in
    // the merged batch and stay where they are for the next list size, whose later stages are generated behind them.
    const McStage *shared = nullptr;          // the fresh stage whose rows are in d_in / d_bytes_a [0, shared_cnt)
    long shared_cnt = 0;
    std::vector<const McStage *> ord;
    for (int li = 0; li < n_L; ++li) {
        const long rows = rows_of[li];
        if (rows == 0) continue;
        ord.clear();
        for (const McStage &s : stages) if (s.li == li && s.fresh) ord.push_back(&s);
        for (const McStage &s : stages) if (s.li == li && !s.fresh) ord.push_back(&s);
        long off = 0;
        for (const McStage *sp : ord) {
            const McStage &s = *sp;
            const size_t id = (size_t)li * n_slots + s.slot;
            polar_code::McSlot &sl = h->mc_slots[id];
            if (sl.cnt == 0) continue;
            if (s.fresh) {
                if ((rc = sl.list[0].ensure((size_t)sl.cnt)) || (rc = sl.list[1].ensure((size_t)sl.cnt))) return rc;
                sl.cur = 0;
                HIP_TRY(polar_launch_mc_init_alive(sl.list[0].p, h->d_slot_n.p + id, s.base + (uint64_t)part, parts, sl.cnt, st));
            }
            const bool reuse = s.fresh && off == 0 && shared && shared->ie == s.ie && shared->base == s.base && shared->T == s.T && shared_cnt == sl.cnt;
            if (!reuse) {
                PolarEncodeParams p;
                fill_enc(h, p);
                p.B = sl.cnt; p.seed = seed; p.sel = sl.list[sl.cur].p; p.n_dev = nullptr;
                fill_channel(h, p, constellation, axis[s.ie]);
                p.llr = h->d_in.p + (size_t)off * N; p.info_out = h->d_bytes_a.p + (size_t)off * K;
                HIP_TRY(polar_launch_synth(p, st));
                if (off == 0) { shared = s.fresh ? &s : nullptr; shared_cnt = sl.cnt; }     // (whatever is at row 0 now)
            }
            off += sl.cnt;
        }
        if ((rc = decode_impl(h, h->d_in.p, 0, rows, nullptr, Ls[li], h->d_out.p, nullptr, st, nullptr, nullptr))) return rc;
        off = 0;
        for (const McStage *sp : ord) {
            const McStage &s = *sp;
            const size_t id = (size_t)li * n_slots + s.slot;
            polar_code::McSlot &sl = h->mc_slots[id];
            if (sl.cnt == 0) continue;
            HIP_TRY(hipMemsetAsync(h->d_slot_n.p + id, 0, sizeof(unsigned int), st));
            HIP_TRY(polar_launch_mc_count_compact(h->d_out.p + (size_t)off * K, h->d_bytes_a.p + (size_t)off * K, sl.cnt, K, sl.list[sl.cur].p, nullptr,
                                                  sl.list[sl.cur ^ 1].p, h->d_slot_n.p + id, h->d_mc_ctr.p + 2 * (size_t)(li * n_e + s.ie), st));
            off += sl.cnt;
        }
    }
    // the new list lengths come back with the counters (bler_impl: mc_step_finish after the stream is done)
    HIP_TRY(hipMemcpyAsync(h->h_slot_n.data(), h->d_slot_n.p, h->h_slot_n.size() * sizeof(unsigned int), hipMemcpyDeviceToHost, st));
    return POLAR_OK;
}
// after the step's stream work is done: the stages' lists are the failures now
static void mc_step_finish(polar_code_t *h, const std::vector<McStage> &stages, int n_slots) {
    for (const McStage &s : stages) {
        const size_t id = (size_t)s.li * n_slots + s.slot;
        polar_code::McSlot &sl = h->mc_slots[id];
        if (sl.cnt == 0) continue;
        sl.cnt = (long)h->h_slot_n[id];
        sl.cur ^= 1;
    }
}

static int mc_batch_impl(polar_code_t *h, int constellation, uint64_t seed, uint64_t t0, long T, long stride,
                         const double *ebno, int n_e, const uint8_t *Ls, int n_L,
                         const uint8_t *enabled, uint64_t *err, uint64_t *bit_err, uint64_t *run) {
    if (!h || !ebno || !Ls || !enabled || !err || !run) return fail(POLAR_E_ARG, "NULL argument");
    if (T <= 0 || stride <= 0 || n_e <= 0 || n_L <= 0) return fail(POLAR_E_ARG, "bad sizes");
    for (int i = 0; i < n_L; ++i)
        if (Ls[i] < 1 || Ls[i] > POLAR_MAX_LIST) return fail(POLAR_E_ARG, "list size %d out of range", (int)Ls[i]);
    DevGuard dg_;
    int rc = ensure_device(h, dg_);
    if (rc) return rc;
    if ((rc = mc_round_launch(h, constellation, seed, t0, T, stride, ebno, n_e, Ls, n_L, enabled, nullptr))) return rc;
    return mc_round_collect(h, T, n_e * n_L, enabled, err, bit_err, run, nullptr);
}

int polar_mc_batch(polar_code_t *h, uint64_t seed, uint64_t t0, long T, long stride,
                   const double *ebno, int n_e, const uint8_t *Ls, int n_L,
                   const uint8_t *enabled, uint64_t *err, uint64_t *run) {
    return mc_batch_impl(h, 0, seed, t0, T, stride, ebno, n_e, Ls, n_L, enabled, err, nullptr, run);
}
int polar_mc_batch_ber(polar_code_t *h, uint64_t seed, uint64_t t0, long T, long stride,
                       const double *ebno, int n_e, const uint8_t *Ls, int n_L,
                       const uint8_t *enabled, uint64_t *err, uint64_t *bit_err, uint64_t *run) {
    return mc_batch_impl(h, 0, seed, t0, T, stride, ebno, n_e, Ls, n_L, enabled, err, bit_err, run);
}
int polar_mc_batch_bicm(polar_code_t *h, int constellation, uint64_t seed, uint64_t t0, long T, long stride,
                        const double *snr_db, int n_s, const uint8_t *Ls, int n_L,
                        const uint8_t *enabled, uint64_t *err, uint64_t *run) {
    if (constellation < POLAR_CONST_ASK4_GRAY || constellation > POLAR_CONST_ASK16_GRAY)
        return fail(POLAR_E_ARG, "unknown constellation %d", constellation);
    return mc_batch_impl(h, constellation, seed, t0, T, stride, snr_db, n_s, Ls, n_L, enabled, err, nullptr, run);
}
int polar_synth_bicm_llr_dev(polar_code_t *h, int constellation, uint64_t seed, uint64_t trial0, long B, double snr_db,
                             double *d_llr, uint8_t *d_info, void *stream) {
    if (!h || !d_llr) return fail(POLAR_E_ARG, "NULL argument");
    if (constellation < POLAR_CONST_ASK4_GRAY || constellation > POLAR_CONST_ASK16_GRAY)
        return fail(POLAR_E_ARG, "unknown constellation %d", constellation);
    if (B <= 0) return B == 0 ? POLAR_OK : fail(POLAR_E_ARG, "negative batch");
    DevGuard dg_;
    int rc = ensure_device(h, dg_);
    if (rc) return rc;
    PolarEncodeParams p;
    fill_enc(h, p);
    p.B = B; p.seed = seed; p.trial0 = trial0; p.llr = d_llr; p.info_out = d_info;
    fill_channel(h, p, constellation, snr_db);
    HIP_TRY(polar_launch_synth(p, (hipStream_t)stream));
    return POLAR_OK;
}

int polar_mc_construction(int n, int constellation, double design_snr_db, uint64_t seed, uint64_t trial0,
                          long num_runs, long batch, uint64_t *num_err) {
    if (!num_err) return fail(POLAR_E_ARG, "NULL argument");
    if (n < 1 || n > POLAR_MAX_N_LOG2) return fail(POLAR_E_ARG, "n = %d out of range [1, %d]", n, POLAR_MAX_N_LOG2);
    if (constellation < POLAR_CONST_ASK4_GRAY || constellation > POLAR_CONST_BPSK)
        return fail(POLAR_E_ARG, "unknown constellation %d", constellation);
    if (num_runs < 0 || batch < 0) return fail(POLAR_E_ARG, "negative run count");
    if (num_runs == 0) return POLAR_OK;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(POLAR_E_DEVICE, "no HIP device: the Monte-Carlo construction has no CPU path");
    const int N = 1 << n, words = (N + 31) / 32;
    if (batch == 0) batch = std::max<long>(64, std::min<long>(32768, (256L << 20) / ((long)N * 8)));   // <= 256 MiB of p1
    batch = std::min(batch, num_runs);
    const int grid = (int)std::min<long>((batch + 63) / 64, 8192);
    DevBuf<double> d_p1, d_y;
    DevBuf<uint32_t> d_info;
    DevBuf<uint8_t> d_x;
    DevBuf<unsigned long long> d_cnt;
    int rc;
    struct Guard {
        DevBuf<double> &a, &b; DevBuf<uint32_t> &c; DevBuf<uint8_t> &d; DevBuf<unsigned long long> &e;
        ~Guard() { a.release(); b.release(); c.release(); d.release(); e.release(); }
    } guard{d_p1, d_y, d_info, d_x, d_cnt};
    if ((rc = d_p1.ensure((size_t)batch * N))) return rc;
    if ((rc = d_info.ensure((size_t)batch * words))) return rc;
    if ((rc = d_y.ensure((size_t)grid * N * 64))) return rc;
    if ((rc = d_x.ensure((size_t)grid * 2 * N * 64))) return rc;
    if ((rc = d_cnt.ensure((size_t)N))) return rc;
    HIP_TRY(hipMemset(d_cnt.p, 0, (size_t)N * sizeof(unsigned long long)));
    PolarConstructParams p;
    p.n = n; p.N = N; p.seed = seed; p.constellation = constellation;
    p.sigma = std::sqrt(1.0 / 2) * std::pow(10.0, -design_snr_db / 20);          // PolarCode.m:170
    p.n0 = p.sigma * p.sigma;
    p.cnorm = polar_const_norm(constellation);
    p.p1 = d_p1.p; p.info = d_info.p; p.y_scr = d_y.p; p.x_scr = d_x.p; p.num_err = d_cnt.p;
    for (long t = 0; t < num_runs; t += batch) {
        p.B = std::min(batch, num_runs - t);
        p.trial0 = trial0 + (uint64_t)t;
        HIP_TRY(polar_launch_mc_front(p, (int)std::min<long>(p.B, 8192), nullptr));
        HIP_TRY(polar_launch_mc_genie(p, (int)std::min<long>((p.B + 63) / 64, grid), nullptr));
    }
    std::vector<unsigned long long> cnt(N);
    HIP_TRY(hipMemcpy(cnt.data(), d_cnt.p, (size_t)N * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    for (int i = 0; i < N; ++i) num_err[i] += (uint64_t)cnt[i];
    return POLAR_OK;
}

}  // extern "C"

namespace {

#ifdef POLAR_TEST_HOOKS
constexpr bool kTestHooks = true;       // libpolar_amd_test.so: the fault-injection knobs of include/polar_amd_debug.h act
#else
constexpr bool kTestHooks = false;      // the product: they do not exist (polar_debug_set rejects their keys)
#endif

// Round sizes (trials of one round over ALL devices): `batch` fixed, or (batch == 0) geometric — the first round is
// max(256, 2 max_err) trials (rounded up to a multiple of the device count), every later one as many as all rounds before
// it together, at most 262144 PER DEVICE: the early stop `num_err > max_err` (:725) keeps its meaning (a point overshoots
// its stopping time by less than 2x) and long sweeps reach full-size launches on every device. (Round 3 capped the round
// over all devices: at 8 GPUs each got 32768 trials per round — four resident rounds of the list-of-32 kernel, less than
// one of the list-size-1 kernel.)
long next_round(long batch, long max_err, long done, long max_runs, int n_dev) {
    long T;
    if (batch > 0) T = batch;
    else if (done == 0) { T = std::max<long>(256, 2 * max_err); T = ((T + n_dev - 1) / n_dev) * n_dev; }
    else T = std::min<long>(done, 262144L * n_dev);
    return std::min(T, max_runs - done);
}

// rank / world / reduce: this process is one of `world` that share the sweep (polar_get_bler_quick_rank): its devices take the
// partitions rank * n_dev + d of world * n_dev, and after every step `reduce` sums the step's counters over the processes
int bler_impl(polar_code_t *h, int constellation, const int *devices, int n_dev, const double *ebno, int n_e, const uint8_t *Ls, int n_L,
              long max_runs, long max_err, uint64_t seed, long batch, double *bler_out, double *ber_out,
              uint64_t *err_out, uint64_t *run_out, int *used_rccl, int rank = 0, int world = 1, polar_reduce_fn reduce = nullptr, void *reduce_user = nullptr) {
    if (!h || !ebno || !Ls || !bler_out) return fail(POLAR_E_ARG, "NULL argument");
    if (n_e <= 0 || n_L <= 0 || max_runs <= 0 || batch < 0 || n_dev < 1) return fail(POLAR_E_ARG, "bad sizes");
    if (world < 1 || rank < 0 || rank >= world || (world > 1 && !reduce)) return fail(POLAR_E_ARG, "bad rank / world / reduce");
    if (constellation == POLAR_CONST_BPSK) constellation = 0;
    if (constellation != 0 && (constellation < POLAR_CONST_ASK4_GRAY || constellation > POLAR_CONST_ASK16_GRAY))
        return fail(POLAR_E_ARG, "unknown constellation %d", constellation);
    for (int i = 0; i < n_L; ++i)
        if (Ls[i] < 1 || Ls[i] > POLAR_MAX_LIST) return fail(POLAR_E_ARG, "list size %d out of range", (int)Ls[i]);
    const int P = n_e * n_L;
    std::vector<uint64_t> err(P, 0), bit(P, 0), run(P, 0);
    DevGuard dg_;
    (void)hipGetDevice(&dg_.prev);
    // one context (clone of the tables + scratch) per device; streams, communicators and worker threads live on the
    // handle and are reused by the next call with the same device list
    std::vector<polar_code *> ctx(n_dev);
    std::vector<int> devs(n_dev);
    int ndev_visible = 0;
    bool dup = false;
    if (hipGetDeviceCount(&ndev_visible) != hipSuccess || ndev_visible <= 0)
        return fail(POLAR_E_DEVICE, "no HIP device available; this library has no CPU decode path");
    for (int d = 0; d < n_dev; ++d) {
        const int dev = devices ? devices[d] : d;
        if (dev < 0 || dev >= ndev_visible) return fail(POLAR_E_ARG, "device %d not visible (%d devices)", dev, ndev_visible);
        devs[d] = dev;
        for (int e = 0; e < d; ++e)
            if (devs[e] == dev) {
                // (test hook share_device — polar_debug_set, no environment form — lets one GPU stand in for several, so
                // that the per-device contexts, worker threads, strided trial partition and counter sum are exercised on a
                // single-GPU box; RCCL cannot have two ranks on one device, the counters are then summed on the host)
                if (!(kTestHooks && h->knobs.share_device)) return fail(POLAR_E_ARG, "device %d listed twice", dev);
                dup = true;
            }
    }
    if (h->multi_poisoned) return fail(POLAR_E_DEVICE, "an earlier multi-device round of this handle never returned: the handle accepts no further get_bler_quick calls");
    const bool want_rccl = (n_dev > 1 || h->knobs.force_rccl) && !h->knobs.no_rccl && !dup;
    if (h->multi && (h->multi->devs != devs || (want_rccl && !h->multi->rccl && g_rccl.load()))) multi_release(h, false);
    for (int d = 0; d < n_dev; ++d) {
        bool again = false;
        for (int e = 0; e < d; ++e) again |= (devs[e] == devs[d]);
        if (h->device < 0 && d == 0) h->device = devs[d];
        // (a repeated device gets a context of its own; an earlier call's are reused)
        if (again) {
            ctx[d] = nullptr;
            for (polar_code *c : h->clones) {
                bool used = false;
                for (int e = 0; e < d; ++e) used |= (ctx[e] == c);
                if (c->device == devs[d] && !used) { ctx[d] = c; break; }
            }
            if (!ctx[d]) ctx[d] = clone_on_device(h, devs[d], true);
        } else ctx[d] = clone_on_device(h, devs[d], false);
        DevGuard g2;
        int rc = ensure_device(ctx[d], g2);
        g2.prev = -1;
        if (rc) return rc;
    }
    if (!h->multi) {
        MultiCtx *m = new MultiCtx;
        m->devs = devs;
        m->streams.assign(n_dev, nullptr);
        h->multi = m;                                // owned from here on: an early return below leaks nothing
        for (int d = 0; d < n_dev; ++d) {
            hipError_t e = hipSetDevice(devs[d]);
            if (e == hipSuccess) e = hipStreamCreateWithFlags(&m->streams[d], hipStreamNonBlocking);
            if (e != hipSuccess) { multi_release(h, false); return fail(POLAR_E_DEVICE, "stream on device %d: %s", devs[d], hipGetErrorString(e)); }
        }
        // RCCL communicators (single process, one rank per device); without RCCL the counters are summed on the host
        if (want_rccl && g_rccl.load()) {
            m->comms.assign(n_dev, nullptr);
            ++g_comm_inits;
            m->rccl = (g_rccl.CommInitAll(m->comms.data(), n_dev, devs.data()) == 0);
            if (!m->rccl) m->comms.clear();
        }
        m->start_workers(n_dev, kTestHooks && h->knobs.force_workers);
        h->worker_threads_started += (long)m->threads.size();
    }
    MultiCtx *mc = h->multi;
    const bool rccl = mc->rccl && want_rccl;
    if (used_rccl) *used_rccl = rccl ? 1 : 0;
    int rc_all = POLAR_OK;
    std::string err_msg;
    h->last_rounds = 0; h->last_round_max_per_device = 0;
    h->round_us.clear();
    // Everything a worker touches during a step lives in ONE shared object that the job holds by value: a worker the
    // watchdog had to give up on (MultiCtx::run_all step 3) may wake up after this function has returned.
    struct Job {
        int n_dev, P, n_e, n_L, n_slots, constellation, fail_dev, fail_coll, stall_dev, part0, parts;
        long stall_ms;
        uint64_t seed;
        bool rccl;
        MultiCtx *mc;
        std::vector<polar_code *> ctx;
        std::vector<double> axis;
        std::vector<uint8_t> Ls;
        std::vector<McStage> stages;
        std::vector<int> rcs;
        std::vector<std::string> msgs;
        std::vector<std::vector<unsigned long long>> host_ctr;
        int step_no = 0;
        std::atomic<int> n_failed{0}, n_failed_coll{0};
    };
    const int n_slots = n_e + 1, parts = world * n_dev;
    auto job = std::make_shared<Job>();
    job->n_dev = n_dev; job->P = P; job->n_e = n_e; job->n_L = n_L; job->n_slots = n_slots; job->constellation = constellation;
    // (fault injection: the test build only — in the product these stay off and the branches on them are compiled out)
    job->fail_dev = kTestHooks ? h->knobs.fail_device : -1; job->fail_coll = kTestHooks ? h->knobs.fail_collective : -1;
    job->stall_dev = kTestHooks ? h->knobs.stall_device : -1; job->stall_ms = kTestHooks ? h->knobs.stall_ms : 0;
    job->part0 = rank * n_dev; job->parts = parts;
    job->seed = seed; job->rccl = rccl; job->mc = mc; job->ctx = ctx;
    job->axis.assign(ebno, ebno + n_e); job->Ls.assign(Ls, Ls + n_L);
    job->rcs.assign(n_dev, POLAR_OK); job->msgs.assign(n_dev, std::string());
    job->host_ctr.assign(n_dev, std::vector<unsigned long long>((size_t)2 * P, 0));
    auto worker = [job](int d) {
        Job &J = *job;
        MultiCtx *mc = J.mc;
        polar_code *c = J.ctx[d];
        const int n_dev = J.n_dev, P = J.P;
        hipStream_t st = mc->streams[d];
        int rc = POLAR_OK;
        std::string msg;
        if (hipSetDevice(c->device) != hipSuccess) { rc = POLAR_E_DEVICE; msg = "hipSetDevice failed"; }
        else if (kTestHooks && d == J.fail_dev && J.step_no == 1) { rc = POLAR_E_DEVICE; msg = "injected failure (fail_device)"; }
        else {
            // (test hook: this worker does not answer for stall_ms in its second step — a hang outside every collective)
            if (kTestHooks && d == J.stall_dev && J.step_no == 1) std::this_thread::sleep_for(std::chrono::milliseconds(J.stall_ms));
            rc = mc_step_launch(c, J.constellation, J.seed, J.stages, J.part0 + d, J.parts, J.axis.data(), J.n_e, J.Ls.data(), J.n_L, J.n_slots, st);
            if (rc) msg = polar_last_error();
        }
        // (1) every worker learns whether ALL of them got this far: either every one enters the collective or none does
        // (a lone rank skipping it would leave the others blocked in it for good)
        if (rc) ++J.n_failed;
        const bool met = n_dev > 1 ? mc->bar->wait() : true;        // false: the watchdog aborted the barrier
        const bool step_ok = met && J.n_failed.load() == 0 && !mc->abort_req.load();
        if (step_ok) {
            // sum of the step's counters over the devices (xGMI), in place on every device
            bool coll_failed = false;
            if (kTestHooks && d == J.fail_coll && J.step_no == 1) coll_failed = true;       // (test hook: the enqueue "fails" on this rank only)
            else if (J.rccl) {
                void *comm = mc->get_comm(d);
                if (!comm || g_rccl.AllReduce(c->d_mc_ctr.p, c->d_mc_ctr.p, (size_t)2 * P, kNcclUint64, kNcclSum, comm, st) != 0) coll_failed = true;
            }
            if (coll_failed) { rc = POLAR_E_DEVICE; msg = (kTestHooks && d == J.fail_coll && J.step_no == 1) ? "injected failure (fail_collective)" : "ncclAllReduce failed"; ++J.n_failed_coll; }
            // (2) a rank whose enqueue failed AFTER the first barrier would leave its peers blocked behind a collective that
            // never completes: everybody meets again, and when any enqueue failed (or the watchdog fired) every rank aborts
            // its OWN communicator BEFORE it waits for its stream
            const bool met2 = n_dev > 1 ? mc->bar->wait() : true;
            if (!met2 || J.n_failed_coll.load() != 0) {
                if (J.rccl) mc->abort_own(d);
                if (!rc) { rc = POLAR_E_DEVICE; msg = met2 ? "round aborted: the counter reduction failed on another device" : "round aborted: watchdog"; }
            } else if (!J.rccl || d == 0) {
                if (hipMemcpyAsync(J.host_ctr[d].data(), c->d_mc_ctr.p, (size_t)2 * P * 8, hipMemcpyDeviceToHost, st) != hipSuccess) { rc = POLAR_E_DEVICE; msg = "counter copy failed"; }
            }
        } else if (!rc) { rc = POLAR_E_DEVICE; msg = (met && !mc->abort_req.load()) ? "round aborted: another device failed" : "round aborted: watchdog"; }
        if (mc->wait_stream(d) != hipSuccess && !rc) { rc = POLAR_E_DEVICE; msg = "stream synchronize failed"; }
        if (!rc) mc_step_finish(c, J.stages, J.n_slots);
        J.rcs[d] = rc; J.msgs[d] = msg;
    };
    // The schedule (see mc_step_launch): rounds in flight, oldest first; per list size each round has a next point `pend`. In a
    // step every round simulates, per list size, its first ENABLED point in [pend, pend of the round before it at the start
    // of the step) — never overtaking the round before it, so that when round r decides on point i (enabled iff point i's
    // errors so far are <= max_err, PolarCode.cpp:725) every round before r has passed point i and no later round has touched
    // it: the decision, the trials simulated and the run counts are exactly those of the reference's round-after-round loop.
    struct PipeRound { long T; uint64_t base; int slot; std::vector<int> pend; std::vector<uint8_t> fresh; };
    std::vector<PipeRound> inflight;
    long done = 0, round_index = 0;
    std::vector<unsigned long long> tot((size_t)2 * P + 1);           // (+ the failure flag of the cross-process reduction)
    for (;;) {
        bool any = false;
        for (int i = 0; i < P; ++i) any |= (err[i] <= (uint64_t)max_err);                // :725
        if (done < max_runs && any && (int)inflight.size() < n_slots) {
            PipeRound R;
            R.T = next_round(batch, max_err, done, max_runs, parts);                    // trials of this round, all devices of all ranks together
            R.base = (uint64_t)done; R.slot = (int)(round_index % n_slots);
            R.pend.assign(n_L, 0); R.fresh.assign(n_L, 1);
            inflight.push_back(R);
            done += R.T; ++round_index;
            ++h->last_rounds;
            h->last_round_max_per_device = std::max(h->last_round_max_per_device, (R.T - job->part0 + parts - 1) / parts);
        }
        if (inflight.empty()) break;
        job->stages.clear();
        for (int li = 0; li < n_L; ++li) {
            int limit = n_e;
            for (PipeRound &R : inflight) {
                const int start = R.pend[li];
                int found = -1;
                for (int ie = start; ie < limit; ++ie)
                    if (err[li * n_e + ie] <= (uint64_t)max_err) { found = ie; break; }
                if (found >= 0) {
                    job->stages.push_back(McStage{li, found, R.slot, R.T, R.base, R.fresh[li] != 0});
                    R.fresh[li] = 0;
                    run[li * n_e + found] += (uint64_t)R.T;                            // :728
                    R.pend[li] = found + 1;
                } else R.pend[li] = limit;
                limit = start;
            }
        }
        while (!inflight.empty()) {
            bool fin = true;
            for (int li = 0; li < n_L; ++li) fin &= (inflight.front().pend[li] >= n_e);
            if (!fin) break;
            inflight.erase(inflight.begin());
        }
        if (job->stages.empty()) continue;
        std::fill(job->rcs.begin(), job->rcs.end(), POLAR_OK);
        for (auto &s_ : job->msgs) s_.clear();
        job->n_failed = 0; job->n_failed_coll = 0;
        const auto t_step = std::chrono::steady_clock::now();
        mc->run_all(worker, h->knobs.multi_timeout_s, h->knobs.multi_grace_s);
        if (mc->timed_out) {
            rc_all = POLAR_E_DEVICE;
            err_msg = "a multi-device round exceeded the watchdog (" + std::to_string(h->knobs.multi_timeout_s) + " s): communicators aborted" +
                      (mc->stuck ? "; a worker never returned, the handle accepts no further get_bler_quick calls" : "");
        }
        if (mc->stuck)                               // which contexts the workers that never came back are working on
            for (int d = 0; d < n_dev; ++d) if (mc->busy && mc->busy[d]) ctx[d]->ctx_stuck = true;
        // (a stuck worker may still write the job's vectors: they are not read then)
        // report the device that failed first-hand, not a peer that was merely told to stop
        for (int pass = 0; pass < 2 && !rc_all && !mc->stuck; ++pass)
            for (int d = 0; d < n_dev; ++d)
                if (job->rcs[d] && (pass == 1 || job->msgs[d].compare(0, 13, "round aborted") != 0)) { rc_all = job->rcs[d]; err_msg = "device " + std::to_string(devs[d]) + ": " + job->msgs[d]; break; }
        std::fill(tot.begin(), tot.end(), 0ull);
        if (!rc_all)
            for (int d = 0; d < (rccl ? 1 : n_dev); ++d)
                for (int i = 0; i < 2 * P; ++i) tot[i] += job->host_ctr[d][i];
        if (reduce) {
            // One process of several: the reduction is collective — every rank calls it once per step, whatever happened to it
            // locally (a rank that left the loop without it would leave its peers waiting in their all-reduce for good). The
            // last element carries the failure flag: after the sum every rank knows whether ANY rank failed, and all stop.
            tot[2 * P] = rc_all ? 1ull : 0ull;
            const int rr = reduce(reduce_user, (uint64_t *)tot.data(), 2 * P + 1);
            if (rr != 0 && !rc_all) { rc_all = POLAR_E_DEVICE; err_msg = "the counter reduction over the processes failed"; }
            else if (!rc_all && tot[2 * P] != 0) { rc_all = POLAR_E_DEVICE; err_msg = "another process of the sweep reported a failure in this step (" + std::to_string((unsigned long long)tot[2 * P]) + " of " + std::to_string(world) + ")"; }
        }
        if (rc_all) break;
        for (int i = 0; i < P; ++i) { err[i] += tot[2 * i]; bit[i] += tot[2 * i + 1]; }
        h->round_us.push_back((long)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t_step).count());
        ++job->step_no;
    }
    // a failed round leaves the communicators in an unknown state: abort and rebuild them next time
    if (rc_all) multi_release(h, true);
    if (rc_all) return fail(rc_all, "%s", err_msg.c_str());
    for (int i = 0; i < P; ++i) {
        bler_out[i] = run[i] ? (double)err[i] / (double)run[i] : 0.0;                 // :777-781
        if (ber_out) ber_out[i] = run[i] ? (double)bit[i] / (double)run[i] : 0.0;     // PolarM/PolarCode.m:848 (per run, as the reference)
        if (err_out) err_out[i] = err[i];
        if (run_out) run_out[i] = run[i];
    }
    return POLAR_OK;
}

}  // namespace

extern "C" {

int polar_get_bler_quick(polar_code_t *h, const double *ebno, int n_e, const uint8_t *Ls, int n_L,
                         long max_runs, long max_err, uint64_t seed, long batch, double *bler_out) {
    if (!h) return fail(POLAR_E_ARG, "NULL argument");
    int dev = h->device;
    if (dev < 0 && hipGetDevice(&dev) != hipSuccess) return fail(POLAR_E_DEVICE, "no HIP device available; this library has no CPU decode path");
    return bler_impl(h, 0, &dev, 1, ebno, n_e, Ls, n_L, max_runs, max_err, seed, batch, bler_out, nullptr, nullptr, nullptr, nullptr);
}
int polar_get_bler_quick_ber(polar_code_t *h, const double *ebno, int n_e, const uint8_t *Ls, int n_L,
                             long max_runs, long max_err, uint64_t seed, long batch, double *bler_out, double *ber_out) {
    if (!h) return fail(POLAR_E_ARG, "NULL argument");
    int dev = h->device;
    if (dev < 0 && hipGetDevice(&dev) != hipSuccess) return fail(POLAR_E_DEVICE, "no HIP device available; this library has no CPU decode path");
    return bler_impl(h, 0, &dev, 1, ebno, n_e, Ls, n_L, max_runs, max_err, seed, batch, bler_out, ber_out, nullptr, nullptr, nullptr);
}
int polar_get_bler_quick_rank(polar_code_t *h, int constellation, int rank, int world, polar_reduce_fn reduce, void *user,
                              const double *axis, int n_e, const uint8_t *Ls, int n_L, long max_runs, long max_err, uint64_t seed,
                              long batch, double *bler_out, double *ber_out, uint64_t *err_out, uint64_t *run_out, long *rounds_out) {
    if (!h) return fail(POLAR_E_ARG, "NULL argument");
    int dev = h->device;
    if (dev < 0 && hipGetDevice(&dev) != hipSuccess) return fail(POLAR_E_DEVICE, "no HIP device available; this library has no CPU decode path");
    const int rc = bler_impl(h, constellation, &dev, 1, axis, n_e, Ls, n_L, max_runs, max_err, seed, batch, bler_out, ber_out, err_out, run_out, nullptr,
                             rank, world, reduce, user);
    if (!rc && rounds_out) *rounds_out = h->last_rounds;
    return rc;
}
int polar_get_bler_quick_multi(polar_code_t *h, const int *devices, int n_dev, const double *ebno, int n_e,
                               const uint8_t *Ls, int n_L, long max_runs, long max_err, uint64_t seed, long batch,
                               double *bler_out, double *ber_out, int *used_rccl) {
    return bler_impl(h, 0, devices, n_dev, ebno, n_e, Ls, n_L, max_runs, max_err, seed, batch, bler_out, ber_out, nullptr, nullptr, used_rccl);
}
int polar_get_bler_quick_multi_ex(polar_code_t *h, int constellation, const int *devices, int n_dev, const double *axis, int n_e,
                                  const uint8_t *Ls, int n_L, long max_runs, long max_err, uint64_t seed, long batch,
                                  double *bler_out, double *ber_out, uint64_t *err_out, uint64_t *run_out, long *rounds_out,
                                  int *used_rccl) {
    if (!h) return fail(POLAR_E_ARG, "NULL argument");
    int dev0 = h->device;
    if (!devices && n_dev == 1 && dev0 < 0 && hipGetDevice(&dev0) != hipSuccess)
        return fail(POLAR_E_DEVICE, "no HIP device available; this library has no CPU decode path");
    // (devices == NULL with one device: the handle's own, like polar_get_bler_quick)
    const int rc = bler_impl(h, constellation, (!devices && n_dev == 1) ? &dev0 : devices, n_dev, axis, n_e, Ls, n_L, max_runs, max_err, seed, batch,
                             bler_out, ber_out, err_out, run_out, used_rccl);
    if (!rc && rounds_out) *rounds_out = h->last_rounds;
    return rc;
}

}  // extern "C"
