// polar_handle.cpp — the handle behind include/polar_amd.h: host-side counterpart of the reference's PolarCode object
// (PolarC/PolarCode.h:17-90). The constructor work (bit-reversal table, Bhattacharyya construction, random-parity matrix)
// runs once on the host and is uploaded as small device tables; everything per codeword (encode, channel, SC / SCL decode,
// error counting) runs in the HIP kernels. There is no CPU decode path: without a HIP device every compute entry point
// fails with POLAR_E_DEVICE.
#include "polar_host.h"

namespace polar_host {

thread_local std::string g_err;

int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

std::atomic<unsigned long> g_allocs{0};

// Host cores this process may use: the affinity mask, capped by the cgroup CPU quota (the GPU boxes report 256 logical
// CPUs and run the container on a 16-CPU quota).
int usable_cpus() {
    int n = (int)std::thread::hardware_concurrency();
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0) n = CPU_COUNT(&set);
    if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char q[32]; long per = 0;
        if (fscanf(f, "%31s %ld", q, &per) == 2 && strcmp(q, "max") != 0 && per > 0) n = std::min<long>(n, std::max<long>(1, atol(q) / per));
        fclose(f);
    }
    return std::max(1, n);
}

void make_bitrev(polar_code *h) {   // create_bit_rev_order, PolarCode.cpp:647-656
    h->bitrev.resize(h->N);
    for (int i = 0; i < h->N; ++i) {
        unsigned r = 0;
        for (int b = 0; b < h->n; ++b) r |= ((unsigned(i) >> b) & 1u) << (h->n - 1 - b);
        h->bitrev[i] = (uint16_t)r;
    }
}

int derive_tables(polar_code *h) {
    const int N = h->N, K = h->K, crc = h->crc, E = K + crc;
    // the first K+crc entries of `order` must be exactly the unfrozen positions
    std::vector<int> rank(N, -1);
    int t = 0;
    for (int i = 0; i < N; ++i) if (!h->frozen[i]) rank[i] = t++;
    if (t != E) return fail(POLAR_E_ARG, "frozen mask has %d unfrozen positions, expected K+crc = %d", t, E);
    std::vector<uint8_t> seen(N, 0);
    h->info_rank.assign(E, 0);
    for (int b = 0; b < E; ++b) {
        int pos = h->order[b];
        if (pos >= N || rank[pos] < 0 || seen[pos])
            return fail(POLAR_E_ARG, "order[%d] = %d is frozen, duplicate or out of range", b, pos);
        seen[pos] = 1;
        h->info_rank[b] = (uint16_t)rank[pos];
    }
    h->W = (E + 31) / 32;
    if (h->W == 0) h->W = 1;
    // rate-0 schedule: aligned all-frozen blocks of 8 / 4 leaves are evaluated in one batch step
    h->sched.assign(N, 0);
    if (h->n >= 4) {
        for (int phi = 0; phi < N;) {
            auto allf = [&](int z) { if (phi % z || phi + z > N) return false; for (int i = 0; i < z; ++i) if (!h->frozen[phi + i]) return false; return true; };
            if (allf(8)) { h->sched[phi] = 3; phi += 8; }
            else if (allf(4)) { h->sched[phi] = 2; phi += 4; }
            else ++phi;
        }
    }
    // schedule of the pruned SC kernel (list size 1): depth-first over the code tree; all-frozen subtrees decide
    // zeros (only their |x| bound is checked), all-unfrozen ones decide at their root
    h->sc_ops.clear();
    {
        auto emit = [&](int type, int S, int base) {
            int sh = 0; while ((1 << sh) < S) ++sh;
            h->sc_ops.push_back((uint32_t)type | ((uint32_t)sh << 3) | ((uint32_t)base << 8));
        };
        struct Rec {
            polar_code *h; decltype(emit) &em; int N;
            void go(int lo, int S) {
                bool allf = true, nonef = true;
                for (int i = lo; i < lo + S; ++i) { if (h->frozen[i]) nonef = false; else allf = false; }
                if (allf) { if (S < N) em(6, S, lo); }
                else if (nonef) em(3, S, lo);
                else {
                    const int hS = S / 2;
                    em(0, hS, lo); go(lo, hS);
                    em(1, hS, lo); go(lo + hS, hS);
                    em(4, hS, lo);
                }
            }
        } rec{h, emit, N};
        rec.go(0, N);
        // The one-codeword-per-wave kernel (sc_lat_kernel) decodes a MIXED node of size 8 (neither all frozen nor all
        // unfrozen) in registers, as one op: type 7, the frozen pattern of its eight leaves in bits 24..31. The ops below
        // such a node are two thirds of the plain schedule (N = 2048, K = 1024: 1693 ops -> 585), and every op of a lone
        // wave is a dependent LDS round trip.
        h->sc_lat_ops.clear();
        {
            auto emit2 = [&](int type, int S, int base, uint32_t hi) {
                int sh = 0; while ((1 << sh) < S) ++sh;
                h->sc_lat_ops.push_back((uint32_t)type | ((uint32_t)sh << 3) | ((uint32_t)base << 8) | (hi << 24));
            };
            struct Rec2 {
                polar_code *h; decltype(emit2) &em; int N;
                void go(int lo, int S) {
                    bool allf = true, nonef = true;
                    for (int i = lo; i < lo + S; ++i) { if (h->frozen[i]) nonef = false; else allf = false; }
                    if (allf) { if (S < N) em(6, S, lo, 0u); }
                    else if (nonef) em(3, S, lo, 0u);
                    else if (S == 8) {
                        uint32_t pat = 0;
                        for (int i = 0; i < 8; ++i) pat |= (uint32_t)(h->frozen[lo + i] ? 1u : 0u) << i;
                        em(7, S, lo, pat);
                    } else {
                        const int hS = S / 2;
                        em(0, hS, lo, 0u); go(lo, hS);
                        em(1, hS, lo, 0u); go(lo + hS, hS);
                        // combine (left half ^= right half): no op of its own — one more step in the combine count of the LAST op
                        // of the right subtree, the node-completing op the chain starts from (types 3 / 6: bits 24..27, type 7:
                        // the size field, its size being fixed)
                        uint32_t &last = h->sc_lat_ops.back();
                        if ((last & 7u) == 7u) last += 1u << 3;
                        else last += 1u << 24;
                    }
                }
            } rec2{h, emit2, N};
            rec2.go(0, N);
            for (uint32_t &w : h->sc_lat_ops) if ((w & 7u) == 7u) w -= 3u << 3;      // (type 7 was emitted with log2(8) in the count field)
        }
        // An F or G step followed by the F step of the child it just produced (depth-first order: always the next
        // entry, one size down) takes that F - and one more - along while its results are in registers, as long
        // as the layers involved are HBM-resident (polar_sc8_min_global_log()): bits 24..25 = number of F steps folded in.
        std::vector<uint32_t> fused;
        for (size_t i = 0; i < h->sc_ops.size(); ++i) {
            uint32_t op = h->sc_ops[i];
            const int type = (int)(op & 7u), sh = (int)((op >> 3) & 15u);
            if (type <= 1 && N <= 65536) {
                int extra = 0;
                while (extra < 2 && i + 1 < h->sc_ops.size()) {
                    const uint32_t nx = h->sc_ops[i + 1];
                    if ((nx & 7u) != 0u || (int)((nx >> 3) & 15u) != sh - 1 - extra || sh - 1 - extra < polar_sc8_min_global_log() - 1 || sh - extra < polar_sc8_min_global_log()) break;
                    ++extra; ++i;
                }
                op |= (uint32_t)extra << 24;
            }
            fused.push_back(op);
        }
        h->sc_ops.swap(fused);
        // The two visits of the top layer read the caller's rows in place (no permuted copy of the batch, no front pass)
        // when both are depth-3 chains into HBM-resident layers and nothing else touches the channel values.
        h->sc_fold = h->n >= polar_sc8_fold_min_log();
        for (uint32_t op : h->sc_ops) {
            const int type = (int)(op & 7u), sh = (int)((op >> 3) & 15u), extra = (int)((op >> 24) & 3u);
            if (sh == h->n) h->sc_fold = false;
            if (type <= 1 && sh == h->n - 1 && extra != 2) h->sc_fold = false;
        }
    }
    h->ctl.resize(N);
    for (int i = 0; i < N; ++i) h->ctl[i] = (uint32_t)(h->frozen[i] ? 1u : 0u) | ((uint32_t)h->sched[i] << 1);
    // Unfrozen leaves in the worst synthetic channels (explicit tables, rates near 1, a design parameter that does not
    // describe the channel): their LLR is an f-chain over hundreds of channel values, 1e-30 and below, and what the
    // reference decides on is the rounding noise of its own arithmetic (HISTORY.md "Where bit-exactness ends"). The
    // LLR-domain kernel follows that arithmetic much further down than the exp-domain one, whose stored form resolves
    // 1e-16 ABSOLUTE near 0. Classified here, once, at no cost per decode: a leaf whose capacity over a BEC(1/2) is
    // below 1e-3 (1 - z, tracked as such: z itself rounds to 1) gets bit 8 of its control word, and the exp-domain
    // kernel hands every codeword in which such a leaf comes out below 1e-8 to the LLR-domain kernel. Codes built for
    // their channel have no such leaf, or never such a value in it (the 16-ASK BICM table: one marked leaf, whose
    // LLR is large on the channel the table was made for).
    h->weak_leaves = 0;
    {
        std::vector<double> z(1, 0.5), om(1, 0.5), z2, om2;         // erasure probability and its complement
        for (int l = 0; l < h->n; ++l) {
            z2.resize(2 * z.size()); om2.resize(2 * z.size());
            for (size_t i = 0; i < z.size(); ++i) {
                z2[2 * i] = 2 * z[i] - z[i] * z[i]; om2[2 * i] = om[i] * om[i];              // f: bit 0 of the leaf index, top layer first
                z2[2 * i + 1] = z[i] * z[i];        om2[2 * i + 1] = om[i] * (1.0 + z[i]);   // g
            }
            z.swap(z2); om.swap(om2);
        }
        for (int i = 0; i < N; ++i)
            if (!h->frozen[i] && om[i] < 1e-3) { h->ctl[i] |= 0x100u; ++h->weak_leaves; }
    }
    // CRC row i as a parity mask over unfrozen ranks, check bit included: crc_check passes iff
    // parity(history & mask_i) == 0 for every row (PolarCode.cpp:93-108)
    h->crc_mask.assign((size_t)crc * h->W, 0u);
    for (int i = 0; i < crc; ++i) {
        uint32_t *m = &h->crc_mask[(size_t)i * h->W];
        for (int j = 0; j < K; ++j)
            if (h->crcm[(size_t)i * K + j] & 1) m[h->info_rank[j] >> 5] ^= 1u << (h->info_rank[j] & 31);
        int r = h->info_rank[K + i];
        m[r >> 5] ^= 1u << (r & 31);
    }
    return POLAR_OK;
}

int ensure_device(polar_code *h, DevGuard &dg) {
    // (every compute entry point passes here: when the worker of a multi-device round that never returned was working on this
    // very context — the handle itself is device 0 of its list — its scratch and tables may still be in that worker's hands)
    if (h->ctx_stuck) return fail(POLAR_E_DEVICE, "an earlier multi-device round of this handle never returned: the handle accepts no further calls (create a new one)");
    int cur = -1;
    if (hipGetDevice(&cur) == hipSuccess && h->device >= 0 && cur != h->device) dg.prev = cur;
    if (h->dev_ready) {
        HIP_TRY(hipSetDevice(h->device));
        return POLAR_OK;
    }
    int cnt = 0;
    hipError_t e = hipGetDeviceCount(&cnt);
    if (e != hipSuccess || cnt <= 0)
        return fail(POLAR_E_DEVICE, "no HIP device available (%s); this library has no CPU decode path",
                    e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    if (h->device < 0) HIP_TRY(hipGetDevice(&h->device));     // (created before any device was visible)
    HIP_TRY(hipSetDevice(h->device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, h->device));
    h->num_cu = prop.multiProcessorCount;
    {
        int v = 0;
        h->lds_per_block = (hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, h->device) == hipSuccess && v > 0) ? (size_t)v : (size_t)64 * 1024;
    }
    int rc;
    if ((rc = upload(h->d_frozen, h->frozen))) return rc;
    if ((rc = upload(h->d_ctl, h->ctl))) return rc;
    if ((rc = upload(h->d_sc_ops, h->sc_ops))) return rc;
    if ((rc = upload(h->d_sc_lat_ops, h->sc_lat_ops))) return rc;
    if ((rc = upload(h->d_order, h->order))) return rc;
    if ((rc = upload(h->d_info_rank, h->info_rank))) return rc;
    if ((rc = upload(h->d_crc_mask, h->crc_mask))) return rc;
    if ((rc = upload(h->d_crcm, h->crcm))) return rc;
    {   // tables of the fp64 exp/log routines (polar_kernels.hip): T[64], RC[129], LC[129]
        std::vector<double> t(322);
        // T[f] = 2^(f/64), written as 2 * 2^(-(64-f)/64) (the value the negative-index form selects)
        for (int j = 0; j < 64; ++j) t[j] = j ? 2.0 * std::exp2(-(double)(64 - j) / 64.0) : 1.0;
        for (int j = 0; j <= 128; ++j) {
            t[64 + j] = 1.0 / (1.0 + (double)j / 128.0);
            t[64 + 129 + j] = std::log1p((double)j / 128.0);
        }
        if ((rc = upload(h->d_tabs, t))) return rc;
    }
    h->dev_ready = true;
    return POLAR_OK;
}

// The measurement knobs of the environment, read ONCE per handle (at creation) and validated; nothing below ever calls
// getenv again (it races with setenv in multi-threaded hosts, and a deployed library must not change its kernel path
// because a variable appeared). polar_debug_set() changes them afterwards (tests, A/B tools).
int read_env_knobs(polar_code *h) {
    auto on = [](const char *name) { const char *e = getenv(name); return e && *e && strcmp(e, "0") != 0; };
    if (const char *e = getenv("POLAR_MODE")) {
        if (!(e[0] >= '0' && e[0] <= '2' && e[1] == 0)) return fail(POLAR_E_ARG, "POLAR_MODE=%s: must be 0, 1 or 2", e);
        h->knobs.mode_override = e[0] - '0';
    }
    h->knobs.sc_no_fold = on("POLAR_SC_NO_FOLD");
    h->knobs.no_tables = on("POLAR_NO_TABLES");
    h->knobs.no_rccl = on("POLAR_NO_RCCL");
    h->knobs.force_rccl = on("POLAR_FORCE_RCCL");
    return POLAR_OK;
}

void drop_clones(polar_code *h) {
    // (a multi-device round of this handle never returned: a worker thread may still be inside the driver with one of the
    // per-device contexts — they are leaked with it, MultiCtx::run_all step 3)
    if (h->multi_poisoned) return;
    for (polar_code *c : h->clones) polar_destroy(c);
    h->clones.clear();
    if (h->hpipe) for (polar_code *&c : h->hpipe->ctx) if (c) { polar_destroy(c); c = nullptr; }   // (the extra decode lanes of host_decode)
}

}  // namespace polar_host

extern "C" {

const char *polar_last_error(void) { return g_err.c_str(); }
int polar_version(void) { return 100; }

int polar_create(int n, int K, double eps, int crc, polar_code_t **out) {
    if (!out) return fail(POLAR_E_ARG, "out is NULL");
    if (n < 1 || n > POLAR_MAX_N_LOG2) return fail(POLAR_E_ARG, "n = %d out of range [1, %d]", n, POLAR_MAX_N_LOG2);
    const int N = 1 << n;
    if (K < 1 || crc < 0 || crc > POLAR_MAX_CRC || K + crc > N)
        return fail(POLAR_E_ARG, "need 1 <= K, 0 <= crc <= %d, K + crc <= N (K=%d crc=%d N=%d)", POLAR_MAX_CRC, K, crc, N);
    polar_code *h = new polar_code;
    h->n = n; h->N = N; h->K = K; h->crc = crc; h->eps = eps;
    make_bitrev(h);
    // initialize_frozen_bits (PolarCode.cpp:17-58): BEC/Bhattacharyya recursion ...
    std::vector<double> z(N, eps);
    for (int it = 0; it < n; ++it) {
        const int inc = 1 << it;
        for (int j = 0; j < inc; ++j)
            for (int i = 0; i < N; i += 2 * inc) {
                double c1 = z[i + j], c2 = z[i + j + inc];
                z[i + j] = c1 + c2 - c1 * c2;
                z[i + j + inc] = c1 * c2;
            }
    }
    // ... then the SAME library call as the reference (std::sort on uint16_t indices with the
    // comparator of PolarCode.cpp:40), so that ties (e.g. channels whose parameter underflowed
    // to 0.0) land in the reference's order under the same libstdc++.
    h->order.resize(N);
    std::iota(h->order.begin(), h->order.end(), (uint16_t)0);
    const std::vector<uint16_t> &br = h->bitrev;
    std::sort(h->order.begin(), h->order.end(), [&](int i1, int i2) { return z[br[i1]] < z[br[i2]]; });
    h->frozen.assign(N, 1);
    for (int i = 0; i < K + crc; ++i) h->frozen[h->order[i]] = 0;
    // random-parity "CRC": crc*K draws of the process-global rand(), as PolarCode.cpp:51-56
    h->crcm.resize((size_t)crc * K);
    for (int b = 0; b < crc; ++b)
        for (int j = 0; j < K; ++j) h->crcm[(size_t)b * K + j] = (uint8_t)(rand() % 2);
    int rc = derive_tables(h);
    if (!rc) rc = read_env_knobs(h);
    if (rc) { delete h; return rc; }
    (void)hipGetDevice(&h->device);          // bound to the current device (stays -1 when none is visible yet)
    *out = h;
    return POLAR_OK;
}

int polar_create_explicit(int n, int K, int crc, const uint8_t *frozen, const uint16_t *order,
                          const uint8_t *crc_matrix, polar_code_t **out) {
    if (!out || !frozen || !order) return fail(POLAR_E_ARG, "NULL argument");
    if (n < 1 || n > POLAR_MAX_N_LOG2) return fail(POLAR_E_ARG, "n = %d out of range", n);
    const int N = 1 << n;
    if (K < 1 || crc < 0 || crc > POLAR_MAX_CRC || K + crc > N) return fail(POLAR_E_ARG, "bad K/crc");
    if (crc > 0 && !crc_matrix) return fail(POLAR_E_ARG, "crc_matrix is NULL with crc = %d", crc);
    polar_code *h = new polar_code;
    h->n = n; h->N = N; h->K = K; h->crc = crc; h->eps = NAN;
    make_bitrev(h);
    h->frozen.assign(frozen, frozen + N);
    h->order.assign(order, order + N);
    h->crcm.assign((size_t)crc * K, 0);
    if (crc) memcpy(h->crcm.data(), crc_matrix, (size_t)crc * K);
    int rc = derive_tables(h);
    if (!rc) rc = read_env_knobs(h);
    if (rc) { delete h; return rc; }
    (void)hipGetDevice(&h->device);
    *out = h;
    // a valid handle, and a status the caller can see: unfrozen leaves in the worst synthetic channels (derive_tables)
    if (h->weak_leaves) { g_err = "explicit table leaves " + std::to_string(h->weak_leaves) + " unfrozen leaves in channels of BEC(1/2) capacity below 1e-3"; return POLAR_W_WEAK_LEAVES; }
    return POLAR_OK;
}

void polar_destroy(polar_code_t *h) {
    if (!h) return;
    // a multi-device round of this handle never returned (MultiCtx::run_all step 3): a worker thread may still be inside the
    // driver with this handle's device contexts — nothing is freed
    if (h->multi_poisoned) return;
    multi_release(h, false);
    for (polar_code *c : h->clones) polar_destroy(c);
    h->clones.clear();
    hostpipe_release(h);
    DevGuard dg_;
    {
        int cur = -1;
        if (h->dev_ready && hipGetDevice(&cur) == hipSuccess && cur != h->device) dg_.prev = cur;
    }
    if (h->dev_ready) (void)hipSetDevice(h->device);
    h->d_frozen.release(); h->d_ctl.release(); h->d_crcm.release(); h->d_order.release(); h->d_info_rank.release();
    h->d_crc_mask.release(); h->d_tabs.release(); h->d_pre.release(); h->d_llr_scr.release(); h->d_c_scr.release(); h->d_hist_scr.release();
    h->d_in.release(); h->d_f32.release(); h->d_out.release(); h->d_bytes_a.release(); h->d_bytes_b.release();
    h->d_counter.release(); h->d_sel.release(); h->d_work.release();
    h->d_ech.release(); h->d_flags.release(); h->d_list.release(); h->d_count.release();
    h->d_alive[0].release(); h->d_alive[1].release(); h->d_nalive.release(); h->d_mc_ctr.release();
    for (auto &sl : h->mc_slots) { sl.list[0].release(); sl.list[1].release(); }
    h->d_slot_n.release();
    if (h->pin_in) (void)hipHostFree(h->pin_in);
    if (h->pin_out) (void)hipHostFree(h->pin_out);
    h->d_sc_ops.release(); h->d_sc_lat_ops.release(); h->d_flag_words.release(); h->d_var_scr.release(); h->d_tab_scr.release();
    delete h;
}

int polar_get_params(const polar_code_t *h, int *n, int *N, int *K, int *crc) {
    if (!h) return fail(POLAR_E_ARG, "NULL handle");
    if (n) *n = h->n;
    if (N) *N = h->N;
    if (K) *K = h->K;
    if (crc) *crc = h->crc;
    return POLAR_OK;
}
int polar_get_frozen(const polar_code_t *h, uint8_t *o) {
    if (!h || !o) return fail(POLAR_E_ARG, "NULL argument");
    memcpy(o, h->frozen.data(), h->N); return POLAR_OK;
}
int polar_get_order(const polar_code_t *h, uint16_t *o) {
    if (!h || !o) return fail(POLAR_E_ARG, "NULL argument");
    memcpy(o, h->order.data(), 2 * (size_t)h->N); return POLAR_OK;
}
int polar_get_bitrev(const polar_code_t *h, uint16_t *o) {
    if (!h || !o) return fail(POLAR_E_ARG, "NULL argument");
    memcpy(o, h->bitrev.data(), 2 * (size_t)h->N); return POLAR_OK;
}
// number of unfrozen leaves derive_tables() marked as weak (see there)
int polar_get_weak_leaves(const polar_code_t *h) { return h ? h->weak_leaves : -1; }
int polar_get_crc_matrix(const polar_code_t *h, uint8_t *m) {
    if (!h || (!m && h->crc)) return fail(POLAR_E_ARG, "NULL argument");
    if (h->crc) memcpy(m, h->crcm.data(), (size_t)h->crc * h->K);
    return POLAR_OK;
}

int polar_set_crc_matrix(polar_code_t *h, const uint8_t *m) {
    if (!h || (!m && h->crc)) return fail(POLAR_E_ARG, "NULL argument");
    drop_clones(h);
    if (h->crc) memcpy(h->crcm.data(), m, (size_t)h->crc * h->K);
    int rc = derive_tables(h);
    if (rc) return rc;
    if (h->dev_ready) {
        if ((rc = upload(h->d_crc_mask, h->crc_mask))) return rc;
        if ((rc = upload(h->d_crcm, h->crcm))) return rc;
    }
    return POLAR_OK;
}

int polar_set_tuning(polar_code_t *h, int waves_per_cu, int lds_log) {
    if (!h) return fail(POLAR_E_ARG, "NULL handle");
    if (waves_per_cu < 0 || waves_per_cu > 32) return fail(POLAR_E_ARG, "waves_per_cu out of range");
    if (lds_log != 0 && (lds_log < 2 || lds_log > 5)) return fail(POLAR_E_ARG, "lds_log must be 0 or 2..5");
    if (lds_log == 2 && waves_per_cu != 0 && waves_per_cu <= 8)
        return fail(POLAR_E_ARG, "lds_log = 2 exists only for the 4-wave-block kernels (waves_per_cu > 8)");
    h->waves_per_cu = waves_per_cu;
    h->lds_log = lds_log;
    drop_clones(h);
    return POLAR_OK;
}

int polar_set_mode(polar_code_t *h, int mode) {
    if (!h) return fail(POLAR_E_ARG, "NULL handle");
    if (mode < 0 || mode > 2) return fail(POLAR_E_ARG, "mode must be 0 (auto), 1 (LLR-domain) or 2 (exp-domain)");
    h->mode = mode;
    drop_clones(h);
    return POLAR_OK;
}

double polar_snr_sqrt_linear(const polar_code_t *h, double ebno_db) {   // PolarCode.cpp:744-745
    if (!h) return NAN;
    return std::pow(10.0f, ebno_db / 20) * std::sqrt(((double)h->K) / ((double)h->N));
}

}  // extern "C"
