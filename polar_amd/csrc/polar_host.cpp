// polar_host.cpp — C-ABI of include/polar_amd.h: handle, code tables, device plumbing.
//
// Host-side counterpart of the reference's PolarCode object (PolarC/PolarCode.h:17-90):
// the constructor work (bit-reversal table, Bhattacharyya construction, random-parity matrix)
// runs once on the host and is uploaded as small device tables; everything per-codeword
// (encode, channel, SC/SCL decode, error counting) runs in the HIP kernels. There is no CPU
// decode path here: without a HIP device every compute entry point fails with POLAR_E_DEVICE.
#include <hip/hip_runtime.h>

#include <dlfcn.h>
#include <sched.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <thread>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <numeric>
#include <string>
#include <vector>

#include "polar_amd.h"
#include "polar_kernels.h"
#include "polar_synth.h"

namespace {

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

#define HIP_TRY(expr)                                                                      \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess)                                                              \
            return fail(POLAR_E_DEVICE, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

std::atomic<unsigned long> g_allocs{0};     // hipMalloc calls of the handles' scratch buffers so far (polar_debug_get "allocs")

template <typename T>
struct DevBuf {
    T *p = nullptr;
    size_t cap = 0;   // elements
    int ensure(size_t n) {
        if (n <= cap) return POLAR_OK;
        if (p) (void)hipFree(p);
        p = nullptr; cap = 0;
        ++g_allocs;
        hipError_t e = hipMalloc((void **)&p, n * sizeof(T));
        if (e != hipSuccess) return fail(POLAR_E_NOMEM, "hipMalloc(%zu bytes) failed: %s", n * sizeof(T), hipGetErrorString(e));
        cap = n;
        return POLAR_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

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

// Copies between the caller's pageable memory and the pinned staging slots of the host-pointer decode path, spread
// over a few parked threads: one core moves ~10 GB/s, the PCIe link ~55 GB/s.
struct CopyPool {
    std::vector<std::thread> threads;
    std::mutex m;
    std::condition_variable cv_job, cv_done;
    unsigned gen = 0;
    bool quit = false;
    char *dst = nullptr; const char *src = nullptr; size_t bytes = 0;
    std::atomic<size_t> next{0};
    int pending = 0;
    static constexpr size_t kSlice = (size_t)1 << 20;
    void work() {
        for (;;) {
            const size_t off = next.fetch_add(kSlice);
            if (off >= bytes) return;
            memcpy(dst + off, src + off, std::min(kSlice, bytes - off));
        }
    }
    void start(int n) {
        for (int i = 0; i < n; ++i)
            threads.emplace_back([this] {
                unsigned seen = 0;
                for (;;) {
                    {
                        std::unique_lock<std::mutex> lk(m);
                        cv_job.wait(lk, [&] { return quit || gen != seen; });
                        if (quit) return;
                        seen = gen;
                    }
                    work();
                    std::lock_guard<std::mutex> lk(m);
                    if (--pending == 0) cv_done.notify_all();
                }
            });
    }
    void copy(void *d, const void *s, size_t n) {          // (the calling thread takes its share)
        if (threads.empty() || n < 2 * kSlice) { memcpy(d, s, n); return; }
        {
            std::lock_guard<std::mutex> lk(m);
            dst = (char *)d; src = (const char *)s; bytes = n; next = 0; pending = (int)threads.size(); ++gen;
        }
        cv_job.notify_all();
        work();
        std::unique_lock<std::mutex> lk(m);
        cv_done.wait(lk, [&] { return pending == 0; });
    }
    ~CopyPool() {
        { std::lock_guard<std::mutex> lk(m); quit = true; }
        cv_job.notify_all();
        for (auto &t : threads) t.join();
    }
};

}  // namespace

// The pipelined host-pointer decode path (host_decode): a ring of pinned staging slots and device slots, one copy stream,
// two decode lanes (the handle itself and a private copy of its tables with its own scratch) on their own streams.
struct HostPipe {
    static constexpr int kMaxLanes = 8, kMaxSlots = kMaxLanes + 2;
    int R = 0;                               // ring slots in use (lanes + 2)
    hipStream_t copy = nullptr, lane[kMaxLanes] = {};
    hipEvent_t h2d[kMaxSlots] = {}, done[kMaxSlots] = {};
    void *pin_in[kMaxSlots] = {}; uint8_t *pin_out[kMaxSlots] = {};
    void *d_in[kMaxSlots] = {}; uint8_t *d_out[kMaxSlots] = {};
    size_t in_cap = 0, out_cap = 0;          // bytes per slot
    polar_code *ctx[kMaxLanes] = {};         // [0] unused (the handle itself); the others are owned, and dropped with the clones by every setter
    std::unique_ptr<CopyPool> pool;
    // what the last pipelined call did (polar_debug_get "host_chunks", "host_chunk_cw", "host_lanes", "host_threads",
    // "host_us_copy_in" / "_wait" / "_copy_out" / "_total": where the calling thread spent its time)
    long last_chunks = 0, last_chunk_cw = 0, last_lanes = 0, last_threads = 0;
    long us_copy_in = 0, us_wait = 0, us_copy_out = 0, us_total = 0;
};

struct polar_code {
    int n = 0, N = 0, K = 0, crc = 0;
    double eps = 0.0;
    std::vector<uint8_t> frozen;     // [N]
    std::vector<uint16_t> order;     // [N]
    std::vector<uint16_t> bitrev;    // [N]
    std::vector<uint8_t> crcm;       // [crc*K]
    // derived
    int W = 0;
    std::vector<uint16_t> info_rank; // [K+crc]
    std::vector<uint32_t> crc_mask;  // [crc*W]
    std::vector<uint8_t> sched;      // [N] rate-0 block schedule for the kernel (0 / 2 / 3)
    std::vector<uint32_t> ctl;       // [N] frozen | sched << 1 | weak-unfrozen-leaf << 8
    int weak_leaves = 0;             // unfrozen leaves no construction for an ordinary channel would leave unfrozen (derive_tables)
    std::vector<uint32_t> sc_ops;    // schedule of the list-size-1 kernel (PolarScParams::ops)
    std::vector<uint32_t> sc_lat_ops;//   the same for its one-codeword-per-wave form: no folded F steps, mixed nodes of size 8 as ONE op (type 7)
    bool sc_fold = false;            //   its top-layer visits read the caller's rows in place (derive_tables)
    // device
    bool dev_ready = false;
    int device = -1, num_cu = 0;
    size_t lds_per_block = 0;        // hipDeviceAttributeMaxSharedMemoryPerBlock (160 KiB on gfx950): what the one-codeword-per-wave kernels are gated on
    DevBuf<uint8_t> d_frozen, d_crcm;
    DevBuf<uint16_t> d_order, d_info_rank;
    DevBuf<uint32_t> d_crc_mask, d_ctl, d_sc_ops, d_sc_lat_ops, d_var_scr;
    DevBuf<double> d_tab_scr;
    DevBuf<unsigned int> d_flag_words;
    DevBuf<double> d_llr_scr, d_tabs, d_pre;
    DevBuf<uint32_t> d_c_scr, d_hist_scr;
    // staging for the host-pointer entry points
    DevBuf<double> d_in;
    DevBuf<float> d_f32;
    DevBuf<uint8_t> d_out, d_bytes_a, d_bytes_b;
    DevBuf<unsigned long long> d_counter;
    DevBuf<unsigned int> d_work;
    DevBuf<uint64_t> d_sel;
    // exp-domain fast path: stored-form channel values, guard flags, fallback work list + its length
    DevBuf<double> d_ech;
    DevBuf<uint8_t> d_flags;
    DevBuf<uint32_t> d_list;
    DevBuf<unsigned int> d_count;
    int mode = 0;                    // 0 auto, 1 LLR-domain kernel only, 2 exp-domain kernel + fallback pass
    // Measurement / test knobs. The environment is read ONCE, when the handle is created (read_env_knobs): a decode never
    // calls getenv. The fault-injection and device-sharing hooks have no environment form at all: polar_debug_set() only.
    struct Knobs {
        int mode_override = -1;      // POLAR_MODE=<0|1|2>: replaces `mode`
        bool sc_no_fold = false;     // POLAR_SC_NO_FOLD: list size 1 decodes a permuted, converted copy (front pass)
        bool no_tables = false;      // POLAR_NO_TABLES: list of 17..32 without the layer-1/2 value tables
        bool no_fuse_front = false;  // (hook) exp-domain lists: separate conversion pass in front of the prefix kernel (the round-3 path)
        bool no_rccl = false;        // POLAR_NO_RCCL: multi-device counters summed on the host
        bool force_rccl = false;     // POLAR_FORCE_RCCL: RCCL even with one device
        bool share_device = false;   // (test hook) one GPU may be listed several times: separate contexts, host-side sum
        int fail_device = -1;        // (test hook) this worker reports a failure in its second round, before the collective
        int fail_collective = -1;    // (test hook) this worker's collective enqueue "fails" in its second round (after the barrier)
        long multi_timeout_s = 1800; // watchdog of a multi-device round: communicators are aborted when a round takes longer
        long multi_grace_s = 10;     //   ... and how long each of its two further steps waits for the workers (MultiCtx::run_all)
        bool force_workers = false;  // (test hook) worker threads (and so the watchdog) even with one device
        int stall_device = -1;       // (test hook) this worker sleeps stall_ms in its second round before it launches anything
        long stall_ms = 0;
        long lat_max_b = 0;          // batches up to this size take the one-codeword-per-wave kernels (0 = default, -1 = never)
        // the pipelined host-pointer path (host_decode): 0 = default everywhere
        long host_pipe_min_bytes = 0;  // input bytes from which a host-pointer batch is pipelined (-1 = never: one copy in, decode, one copy out)
        long host_chunk_bytes = 0;     // input bytes per chunk / staging slot
        long host_lanes = 0;           // decode lanes (1 or 2)
        long host_threads = 0;         // threads that copy between the caller's memory and the pinned slots (calling thread included)
        long host_ramp = 0;            // -1: no small first chunks (all chunks equal)
    } knobs;
    // Monte-Carlo engine (device side): alive lists (double-buffered), their lengths, per-round counters
    DevBuf<uint64_t> d_alive[2];
    DevBuf<unsigned int> d_nalive;           // [2]
    DevBuf<unsigned long long> d_mc_ctr;     // [n_L*n_e][2]: block errors, bit errors of the round
    // pipelined rounds (mc_step_launch): the alive lists of the rounds in flight — slot (list size, round mod slots), double-buffered —,
    // the lengths the device wrote, and their host copy
    struct McSlot { DevBuf<uint64_t> list[2]; int cur = 0; long cnt = 0; };
    std::vector<McSlot> mc_slots;
    DevBuf<unsigned int> d_slot_n;
    std::vector<unsigned int> h_slot_n;
    // per-device clones for polar_get_bler_quick_multi (owned by this handle)
    std::vector<polar_code *> clones;
    // streams + RCCL communicators of the last multi-device call, kept for the next one with the same device list
    // (an 8-rank ncclCommInitAll costs about as long as a short sweep runs)
    struct MultiCtx *multi = nullptr;
    bool multi_poisoned = false;     // a multi-device round never returned (MultiCtx::run_all step 3): no further multi-device calls
    // zero-copy staging of the host-pointer entry points for the smallest batches (host_decode): pinned, device-mapped
    void *pin_in = nullptr, *pin_in_dev = nullptr;     // LLR rows
    uint8_t *pin_out = nullptr, *pin_out_dev = nullptr; // decoded bits [B][K] followed by one flag byte per codeword
    size_t pin_in_cap = 0, pin_out_cap = 0;
    uint8_t *lat_flag_bytes = nullptr;                  // (set around a decode_impl call by host_decode)
    struct HostPipe *hpipe = nullptr;                   // pipelined staging of the large host-pointer batches (host_decode)
    // statistics of the last get_bler_quick* call (polar_debug_get)
    long last_rounds = 0, last_round_max_per_device = 0, worker_threads_started = 0;
    std::vector<long> round_us;      //   wall time of every round of that call (polar_debug_get "round_us_min" / "_median" / "_max" / "_first")
    // tuning
    int waves_per_cu = 0, lds_log = 0, pipe = -1;
    bool prefix_on = true;
};

namespace {

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

template <typename T>
int upload(DevBuf<T> &d, const std::vector<T> &v) {
    size_t n = v.size() ? v.size() : 1;
    int rc = d.ensure(n);
    if (rc) return rc;
    if (v.size()) HIP_TRY(hipMemcpy(d.p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return POLAR_OK;
}

// Every compute entry point runs on the handle's device (the one current at creation) and leaves the
// caller's current device as it found it.
struct DevGuard {
    int prev = -1;
    ~DevGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

int ensure_device(polar_code *h, DevGuard &dg) {
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

int pow2ceil(int v) { int p = 1; while (p < v) p <<= 1; return p; }

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

void multi_release(polar_code *h, bool abort_comms);     // (defined next to bler_impl)
polar_code *copy_ctx(polar_code *h, int dev);            // (defined next to clone_on_device)

}  // namespace

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

static void hostpipe_release(polar_code_t *h);      // (defined with host_decode)

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
int polar_get_crc_matrix(const polar_code_t *h, uint8_t *m) {
    if (!h || (!m && h->crc)) return fail(POLAR_E_ARG, "NULL argument");
    if (h->crc) memcpy(m, h->crcm.data(), (size_t)h->crc * h->K);
    return POLAR_OK;
}
// the per-device contexts of polar_get_bler_quick_multi are copies of the handle's tables and settings: any setter
// drops them (they are rebuilt by the next multi-device call)
static void drop_clones(polar_code_t *h) {
    for (polar_code *c : h->clones) polar_destroy(c);
    h->clones.clear();
    if (h->hpipe) for (polar_code *&c : h->hpipe->ctx) if (c) { polar_destroy(c); c = nullptr; }   // (the extra decode lanes of host_decode)
}

// test hook: number of unfrozen leaves derive_tables() marked as weak (see there)
int polar_debug_weak_leaves(const polar_code_t *h) { return h ? h->weak_leaves : -1; }

// test / measurement hooks (include/polar_amd.h): the knobs the environment sets at creation, and the ones that have no
// environment form
int polar_debug_set(polar_code_t *h, const char *key, long value) {
    if (!h || !key) return fail(POLAR_E_ARG, "NULL argument");
    polar_code::Knobs &k = h->knobs;
    const std::string s(key);
    if (s == "mode_override") { if (value < -1 || value > 2) return fail(POLAR_E_ARG, "mode_override must be -1 (none), 0, 1 or 2"); k.mode_override = (int)value; }
    else if (s == "sc_no_fold") k.sc_no_fold = value != 0;
    else if (s == "no_tables") k.no_tables = value != 0;
    else if (s == "no_fuse_front") k.no_fuse_front = value != 0;
    else if (s == "no_prefix") h->prefix_on = (value == 0);          // (the all-frozen prefix decoded leaf by leaf by the list kernel itself)
    // (the cached streams / communicators / worker threads of the last device list were built under the old setting)
    else if (s == "no_rccl") { k.no_rccl = value != 0; multi_release(h, false); }
    else if (s == "force_rccl") { k.force_rccl = value != 0; multi_release(h, false); }
    else if (s == "share_device") { k.share_device = value != 0; multi_release(h, false); }
    else if (s == "fail_device") k.fail_device = (int)value;
    else if (s == "fail_collective") k.fail_collective = (int)value;
    else if (s == "lat_max_b") k.lat_max_b = value;
    else if (s == "host_pipe_min_bytes") k.host_pipe_min_bytes = value;
    else if (s == "host_chunk_bytes") k.host_chunk_bytes = value;
    else if (s == "host_lanes") { if (value < 0 || value > HostPipe::kMaxLanes) return fail(POLAR_E_ARG, "host_lanes must be 0 (default) or 1 .. %d", HostPipe::kMaxLanes); k.host_lanes = value; }
    else if (s == "host_ramp") k.host_ramp = value;
    else if (s == "host_threads") { if (value < 0) return fail(POLAR_E_ARG, "host_threads must be >= 0"); k.host_threads = value; }
    else if (s == "multi_grace_s") { if (value < 0) return fail(POLAR_E_ARG, "multi_grace_s must be >= 0"); k.multi_grace_s = value; }
    else if (s == "force_workers") { k.force_workers = value != 0; multi_release(h, false); }
    else if (s == "stall_device") k.stall_device = (int)value;
    else if (s == "stall_ms") { if (value < 0) return fail(POLAR_E_ARG, "stall_ms must be >= 0"); k.stall_ms = value; }
    else if (s == "multi_timeout_s") { if (value < 0) return fail(POLAR_E_ARG, "multi_timeout_s must be >= 0 (0 = no watchdog)"); k.multi_timeout_s = value; }
    else return fail(POLAR_E_ARG, "polar_debug_set: unknown key '%s'", key);
    drop_clones(h);          // (the per-device contexts carry a copy of the knobs)
    return POLAR_OK;
}
// (measurement builds: the handle's alpha scratch, where instrumented kernels leave their counters)
void *polar_debug_scratch_ptr(polar_code_t *h) { return h ? (void *)h->d_llr_scr.p : nullptr; }
long polar_debug_get(const polar_code_t *h, const char *key) {
    if (!key) return -1;
    const std::string s(key);
    if (s == "allocs") return (long)g_allocs.load();
    if (s == "comm_inits") return (long)polar_debug_comm_inits();
    if (!h) return -1;
    if (s == "weak_leaves") return h->weak_leaves;
    if (s == "mode_override") return h->knobs.mode_override;
    if (s == "last_rounds") return h->last_rounds;
    if (s == "last_round_max_per_device") return h->last_round_max_per_device;
    if (s == "worker_threads_started") return h->worker_threads_started;
    if (s.compare(0, 9, "round_us_") == 0) {        // wall time of the steps of the last get_bler_quick* call
        if (h->round_us.empty()) return 0;
        std::vector<long> v(h->round_us);
        if (s == "round_us_count") return (long)v.size();
        if (s == "round_us_first") return v.front();
        std::sort(v.begin(), v.end());
        if (s == "round_us_min") return v.front();
        if (s == "round_us_max") return v.back();
        if (s == "round_us_median") return v[v.size() / 2];
        return -1;
    }
    if (s == "multi_poisoned") return h->multi_poisoned ? 1 : 0;
    if (s == "host_chunks") return h->hpipe ? h->hpipe->last_chunks : 0;
    if (s == "host_chunk_cw") return h->hpipe ? h->hpipe->last_chunk_cw : 0;
    if (s == "host_lanes") return h->hpipe ? h->hpipe->last_lanes : 0;
    if (s == "host_threads") return h->hpipe ? h->hpipe->last_threads : 0;
    if (s == "host_us_copy_in") return h->hpipe ? h->hpipe->us_copy_in : 0;
    if (s == "host_us_wait") return h->hpipe ? h->hpipe->us_wait : 0;
    if (s == "host_us_copy_out") return h->hpipe ? h->hpipe->us_copy_out : 0;
    if (s == "host_us_total") return h->hpipe ? h->hpipe->us_total : 0;
    return -1;
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

// ------------------------------------------------------------------------------------------
int polar_decode_scl_llr_batch_dev(polar_code_t *h, const double *d_llr, long B, int L, uint8_t *d_out,
                                   double *d_pm, void *stream) {
    return polar_decode_scl_llr_batch_dev_ev(h, d_llr, B, L, d_out, d_pm, stream, nullptr, nullptr);
}

// list size 1, small batches: one codeword per wave, whole state in LDS (sc_lat_kernel)
static bool use_sc_lat(const polar_code_t *h, long B) {
    return h->n <= polar_sc_lat_max_log() && polar_sc_lat_lds_bytes(h->N, (int)h->sc_lat_ops.size()) <= h->lds_per_block &&
           h->knobs.lat_max_b >= 0 && B <= (h->knobs.lat_max_b ? h->knobs.lat_max_b : 2048);
}
// phase (list size 1 with the one-codeword-per-wave kernel only): 0 = everything; 1 = the decode kernel alone — the caller
// looks at the flag words itself and runs phase 2 (work list + general kernel over the flagged codewords) only when one is
// set; *deferred reports whether phase 1 really left the fallback out
static int decode_impl(polar_code_t *h, const void *d_llr, int llr_f32, long B, const unsigned int *n_dev, int L, uint8_t *d_out,
                       double *d_pm, void *stream, void *ev_start, void *ev_stop, int phase = 0, int *deferred = nullptr);

int polar_decode_scl_llr_batch_dev_ev(polar_code_t *h, const double *d_llr, long B, int L, uint8_t *d_out,
                                      double *d_pm, void *stream, void *ev_start, void *ev_stop) {
    return decode_impl(h, d_llr, 0, B, nullptr, L, d_out, d_pm, stream, ev_start, ev_stop);
}

// B rows are allocated; when n_dev != nullptr only the first min(B, *n_dev) exist (count read on the device)
static int decode_impl(polar_code_t *h, const void *d_llr, int llr_f32, long B, const unsigned int *n_dev, int L, uint8_t *d_out,
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
    const int pipe = (wpc > 8) ? 0 : 1;
    int lds_log = h->lds_log ? h->lds_log : (pipe ? 4 : 3);
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
    {   // the per-wave state scratch is big*512 B per resident wave (1 MiB at N=2048, 16 MiB at N=32768):
        // cap it at 24 GiB of the 288 GB HBM by running fewer persistent waves for very long codes
        const size_t per_wave = big * 64 * sizeof(double) + 1;
        const long cap = (long)((24ull << 30) / per_wave);
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
                          B <= (h->knobs.lat_max_b ? h->knobs.lat_max_b : lat_resident);
    // (measured, N = 2048: L = 4 B = 1 ... 256 2.45 ... 2.59 ms against 3.87 ... 4.36 ms for the batch kernel, L = 2 2.9 ... 3.0 against
    // 5.9 ... 6.9 ms; beyond the waves the LDS lets the device hold at once — one per CU for lists of 4 and 8 at N = 2048, three for
    // lists of 2 — the batch kernel wins: that is the default threshold)
    if (lat_list) {
        PolarDecodeParams pl = p;
        pl.prefix_q = 0; pl.prefix_len = 0; pl.pre = nullptr;
        const int blocks = (int)std::min<long>(B, lat_resident);
        if (lat_ed) {
            // (sized for the largest batch this path ever takes — a few hundred entries — so that the first call reserves it)
            const size_t cap = (size_t)std::max<long>(B, lat_resident);
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

// The host-pointer entry points: H2D copy, decode, wait, copy the bits back. One codeword at a time is the reference's own
// call pattern (PolarCode.cpp:756, PolarM/main_MC_CC_Comparison.m:96), so the smallest batches of list size 1 are kept to
// the fewest driver calls: the rows are staged in PINNED, device-mapped host memory that the one-codeword-per-wave kernel
// reads and writes directly (no DMA copies: a 16-KB hipMemcpy costs more than moving the bytes), the flags come back with
// the bits, and the work list + general kernel over the flagged codewords (normally none) are launched only when a flag is
// set — the device-resident entry points, which must not wait, always launch them.
//
// LARGE batches are pipelined (round 5): the batch is cut into chunks of ~64 MiB of LLRs; chunk k is copied by a few host
// threads from the caller's pageable memory into a pinned slot (a pageable hipMemcpy is a single-threaded staging loop
// inside the runtime: a fraction of the link), moved by the copy engine on a copy stream, decoded on one of TWO decode
// lanes — the handle and a private copy of its tables with its own scratch, each on its own stream: the persistent waves of
// chunk k + 1 take the slots chunk k's waves leave, so a launch's tail overlaps the next launch's head instead of idling
// the device once per chunk — and its bits come back through a pinned slot: H2D(k + 1) || decode(k) || D2H(k - 1).
static void hostpipe_release(polar_code_t *h) {
    HostPipe *hp = h->hpipe;
    if (!hp) return;
    h->hpipe = nullptr;
    for (polar_code *c : hp->ctx) if (c) polar_destroy(c);
    for (int i = 0; i < HostPipe::kMaxSlots; ++i) {
        if (hp->pin_in[i]) (void)hipHostFree(hp->pin_in[i]);
        if (hp->pin_out[i]) (void)hipHostFree(hp->pin_out[i]);
        if (hp->d_in[i]) (void)hipFree(hp->d_in[i]);
        if (hp->d_out[i]) (void)hipFree(hp->d_out[i]);
        if (hp->h2d[i]) (void)hipEventDestroy(hp->h2d[i]);
        if (hp->done[i]) (void)hipEventDestroy(hp->done[i]);
    }
    if (hp->copy) (void)hipStreamDestroy(hp->copy);
    for (hipStream_t s : hp->lane) if (s) (void)hipStreamDestroy(s);
    delete hp;
}

static int hostpipe_ensure(polar_code_t *h, size_t in_slot, size_t out_slot, int lanes, int threads) {
    if (!h->hpipe) h->hpipe = new HostPipe;
    HostPipe *hp = h->hpipe;
    const int R = lanes + 2;
    // HIP multiplexes its streams onto a few hardware queues PER PRIORITY LEVEL (four by default), and two streams that share
    // a queue take turns: a 64-MiB copy queued behind a 5-ms decode kernel, or the two decode lanes behind each other, and
    // nothing overlaps (measured: rocprofv3 kernel trace of eight lanes — three queue ids, two kernels at a time). The three
    // priority levels have queue pools of their own: the copy stream takes the high one, the first two lanes normal and low.
    int prio_least = 0, prio_greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
    if (!hp->copy) HIP_TRY(hipStreamCreateWithPriority(&hp->copy, hipStreamNonBlocking, prio_greatest));
    for (int l = 0; l < lanes; ++l) {
        // lanes beyond the third (polar_debug_set "host_lanes" only) are spread over the three pools
        static const int pool_of[HostPipe::kMaxLanes] = {0, 1, 0, 2, 1, 2, 0, 1};      // 0 normal, 1 low, 2 high
        const int prio = pool_of[l] == 1 ? prio_least : pool_of[l] == 2 ? prio_greatest : (prio_least + prio_greatest) / 2;
        if (!hp->lane[l]) HIP_TRY(hipStreamCreateWithPriority(&hp->lane[l], hipStreamNonBlocking, prio));
    }
    for (int i = 0; i < R; ++i) {
        if (!hp->h2d[i]) HIP_TRY(hipEventCreateWithFlags(&hp->h2d[i], hipEventDisableTiming));
        if (!hp->done[i]) HIP_TRY(hipEventCreateWithFlags(&hp->done[i], hipEventDisableTiming));
    }
    // (slots are sized together: a larger chunk replaces all of them, more lanes add slots of the current size)
    if (hp->in_cap < in_slot || hp->out_cap < out_slot) {
        for (int i = 0; i < HostPipe::kMaxSlots; ++i) {
            if (hp->pin_in[i]) (void)hipHostFree(hp->pin_in[i]);
            if (hp->d_in[i]) (void)hipFree(hp->d_in[i]);
            if (hp->pin_out[i]) (void)hipHostFree(hp->pin_out[i]);
            if (hp->d_out[i]) (void)hipFree(hp->d_out[i]);
            hp->pin_in[i] = nullptr; hp->d_in[i] = nullptr; hp->pin_out[i] = nullptr; hp->d_out[i] = nullptr;
        }
        hp->in_cap = std::max(hp->in_cap, in_slot); hp->out_cap = std::max(hp->out_cap, out_slot);
    }
    for (int i = 0; i < R; ++i) {
        if (hp->pin_in[i]) continue;
        ++g_allocs;
        HIP_TRY(hipHostMalloc(&hp->pin_in[i], hp->in_cap, hipHostMallocDefault));
        HIP_TRY(hipMalloc(&hp->d_in[i], hp->in_cap));
        HIP_TRY(hipHostMalloc((void **)&hp->pin_out[i], hp->out_cap, hipHostMallocDefault));
        HIP_TRY(hipMalloc((void **)&hp->d_out[i], hp->out_cap));
    }
    hp->R = R;
    for (int l = 1; l < lanes; ++l) {
        if (hp->ctx[l]) continue;
        hp->ctx[l] = copy_ctx(h, h->device);
        DevGuard g2;
        int rc = ensure_device(hp->ctx[l], g2);
        g2.prev = -1;
        if (rc) return rc;
    }
    if (!hp->pool || (int)hp->pool->threads.size() != threads - 1) {
        hp->pool.reset(new CopyPool);
        hp->pool->start(threads - 1);
    }
    return POLAR_OK;
}

static int host_decode_pipelined(polar_code_t *h, const void *llr, int llr_f32, long B, int L, uint8_t *out, long chunk_cw, int lanes, int threads, bool ramp) {
    using clk = std::chrono::steady_clock;
    auto us = [](clk::time_point a, clk::time_point b) { return (long)std::chrono::duration_cast<std::chrono::microseconds>(b - a).count(); };
    const clk::time_point t_begin = clk::now();
    const size_t esz = llr_f32 ? sizeof(float) : sizeof(double);
    const size_t row_in = (size_t)h->N * esz, row_out = (size_t)h->K;
    // The chunks. A launch of the list kernels takes milliseconds whatever it carries (the N-step chain of one wave), so
    // (a) the first chunks are SMALL — an eighth of the full size, doubling: the device starts a fraction of a millisecond
    // after the call instead of after the first 128 MiB — and (b) the full-size chunks are equal (a short last one would be
    // one more launch latency at the end of the call).
    std::vector<long> start, size;
    {
        long done = 0;
        if (ramp)
            for (long c = std::max<long>(8, chunk_cw / 8 / 8 * 8); c < chunk_cw && B - done > 2 * c; c *= 2) { start.push_back(done); size.push_back(c); done += c; }
        const long rest = B - done, n_full = (rest + chunk_cw - 1) / chunk_cw;
        const long C = std::min(chunk_cw, (((rest + n_full - 1) / n_full) + 7) / 8 * 8);
        for (; done < B; done += C) { start.push_back(done); size.push_back(std::min(C, B - done)); }
    }
    const long n_chunks = (long)start.size();
    const long C = *std::max_element(size.begin(), size.end());
    lanes = (int)std::min<long>(lanes, n_chunks);
    int rc = hostpipe_ensure(h, (size_t)C * row_in, (size_t)C * row_out, lanes, threads);
    if (rc) return rc;
    HostPipe *hp = h->hpipe;
    hp->last_chunks = n_chunks; hp->last_chunk_cw = C; hp->last_lanes = lanes; hp->last_threads = threads;
    hp->us_copy_in = hp->us_wait = hp->us_copy_out = 0;
    // (the handle's scratch may still be in use by work the caller put on the null stream through this handle)
    HIP_TRY(hipStreamSynchronize(nullptr));
    const int R = hp->R;
    auto drain = [&] { (void)hipStreamSynchronize(hp->copy); for (int l = 0; l < lanes; ++l) (void)hipStreamSynchronize(hp->lane[l]); };
    auto finish = [&](long j) -> int {                      // chunk j: wait for its bits, hand them to the caller
        const int slot = (int)(j % R);
        const clk::time_point t0 = clk::now();
        hipError_t e = hipEventSynchronize(hp->done[slot]);
        const clk::time_point t1 = clk::now();
        if (e != hipSuccess) return fail(POLAR_E_DEVICE, "host pipeline: chunk %ld failed: %s", j, hipGetErrorString(e));
        hp->pool->copy(out + (size_t)start[j] * row_out, hp->pin_out[slot], (size_t)size[j] * row_out);
        hp->us_wait += us(t0, t1); hp->us_copy_out += us(t1, clk::now());
        return POLAR_OK;
    };
    long k = 0;
    for (; k < n_chunks && !rc; ++k) {
        const int slot = (int)(k % R);
        const long b0 = start[k], nb = size[k];
        if (k >= R && (rc = finish(k - R))) break;          // (frees the slot: its H2D, decode and D2H are all behind `done`)
        const clk::time_point t0 = clk::now();
        hp->pool->copy(hp->pin_in[slot], (const char *)llr + (size_t)b0 * row_in, (size_t)nb * row_in);
        hp->us_copy_in += us(t0, clk::now());
        hipError_t e = hipMemcpyAsync(hp->d_in[slot], hp->pin_in[slot], (size_t)nb * row_in, hipMemcpyHostToDevice, hp->copy);
        if (e == hipSuccess) e = hipEventRecord(hp->h2d[slot], hp->copy);
        const int l = (int)(k % lanes);
        hipStream_t st = hp->lane[l];
        if (e == hipSuccess) e = hipStreamWaitEvent(st, hp->h2d[slot], 0);
        if (e != hipSuccess) { rc = fail(POLAR_E_DEVICE, "host pipeline: copy of chunk %ld: %s", k, hipGetErrorString(e)); break; }
        if ((rc = decode_impl(l ? hp->ctx[l] : h, hp->d_in[slot], llr_f32, nb, nullptr, L, hp->d_out[slot], nullptr, st, nullptr, nullptr))) break;
        e = hipMemcpyAsync(hp->pin_out[slot], hp->d_out[slot], (size_t)nb * row_out, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipEventRecord(hp->done[slot], st);
        if (e != hipSuccess) { rc = fail(POLAR_E_DEVICE, "host pipeline: result copy of chunk %ld: %s", k, hipGetErrorString(e)); break; }
    }
    if (rc) { const std::string msg = g_err; drain(); g_err = msg; return rc; }
    for (long j = std::max<long>(0, n_chunks - R); j < n_chunks; ++j)
        if ((rc = finish(j))) { const std::string msg = g_err; drain(); g_err = msg; return rc; }
    hp->us_total = us(t_begin, clk::now());
    return POLAR_OK;
}

static int host_decode(polar_code_t *h, const void *llr, int llr_f32, long B, int L, uint8_t *out) {
    if (!h || !llr || !out) return fail(POLAR_E_ARG, "NULL argument");
    if (L < 1 || L > POLAR_MAX_LIST) return fail(POLAR_E_ARG, "list size %d out of range [1, %d]", L, POLAR_MAX_LIST);
    if (B < 0) return fail(POLAR_E_ARG, "negative batch");
    if (B == 0) return POLAR_OK;
    DevGuard dg_;
    int rc = ensure_device(h, dg_);
    if (rc) return rc;
    const size_t esz = llr_f32 ? sizeof(float) : sizeof(double);
    const size_t in_bytes = (size_t)B * h->N * esz, out_bytes = (size_t)B * h->K;
    const int mode = h->knobs.mode_override >= 0 ? h->knobs.mode_override : h->mode;
    if (h->hpipe) h->hpipe->last_chunks = 0;
    {
        const polar_code::Knobs &kn = h->knobs;
        const size_t min_bytes = kn.host_pipe_min_bytes > 0 ? (size_t)kn.host_pipe_min_bytes : (size_t)32 << 20;
        if (kn.host_pipe_min_bytes >= 0 && in_bytes >= min_bytes) {
            // full-size chunks: 64 MiB of LLRs at list size 1 (the link is the bound, its kernel answers in a millisecond). The
            // list kernels' launches take 4 .. 8 ms whatever they carry, and the chunks in flight must cover what the link
            // delivers meanwhile: 8192 codewords of N = 2048 for the lists of 17 and more (what the device holds at a time:
            // 128 MiB of doubles), 128 MiB of doubles or floats in between (three lanes x 16384 float rows); at least four
            // full-size chunks per batch
            const long fill_cw = 8192L * 2048 / h->N;
            const bool fills = L > 1 && (long)h->num_cu * 16 * (64 / pow2ceil(L)) <= fill_cw;
            size_t cb = kn.host_chunk_bytes > 0 ? (size_t)kn.host_chunk_bytes : (L == 1 ? (size_t)64 << 20 : fills ? (size_t)fill_cw * h->N * esz : (size_t)128 << 20);
            if (kn.host_chunk_bytes <= 0) cb = std::min(cb, std::max<size_t>(in_bytes / 4, (size_t)8 << 20));
            const long chunk_cw = std::max<long>(8, (long)(cb / ((size_t)h->N * esz)) / 8 * 8);
            // decode lanes: HIP multiplexes its streams onto four hardware queues, two streams on one queue take turns — two
            // lanes for list size 1 and for the lists whose full-size chunk fills the device, three in between
            int lanes = (int)kn.host_lanes;
            if (lanes <= 0) lanes = (L == 1 || (long)h->num_cu * 16 * (64 / pow2ceil(L)) <= chunk_cw) ? 2 : 3;
            // small first chunks (host_decode_pipelined) unless one full-size chunk already fills the device: the list-of-32
            // kernel runs such a chunk as ONE round of resident waves, and three more launches cost it more than the early start
            // returns (headline, 65536 codewords: 0.87 of the device-resident rate with them, 0.90 .. 0.93 without)
            const bool ramp = kn.host_ramp > 0 || (kn.host_ramp == 0 && (L == 1 || (long)h->num_cu * 16 * (64 / pow2ceil(L)) > chunk_cw));
            const int threads = kn.host_threads > 0 ? (int)std::min<long>(kn.host_threads, 64) : std::max(1, std::min(8, usable_cpus() / 2));
            if (B > chunk_cw) return host_decode_pipelined(h, llr, llr_f32, B, L, out, chunk_cw, lanes, threads, ramp);
        }
    }
    if (L == 1 && mode != 1 && use_sc_lat(h, B) && B <= 64) {
        if (h->pin_in_cap < in_bytes) {
            if (h->pin_in) (void)hipHostFree(h->pin_in);
            h->pin_in = nullptr; h->pin_in_cap = 0;
            const size_t cap = std::max(in_bytes, (size_t)64 * h->N * sizeof(double));
            HIP_TRY(hipHostMalloc(&h->pin_in, cap, hipHostMallocMapped));
            HIP_TRY(hipHostGetDevicePointer(&h->pin_in_dev, h->pin_in, 0));
            h->pin_in_cap = cap;
        }
        if (h->pin_out_cap < out_bytes + (size_t)B) {
            if (h->pin_out) (void)hipHostFree(h->pin_out);
            h->pin_out = nullptr; h->pin_out_cap = 0;
            const size_t cap = (size_t)64 * (h->K + 1);
            HIP_TRY(hipHostMalloc((void **)&h->pin_out, cap, hipHostMallocMapped));
            HIP_TRY(hipHostGetDevicePointer((void **)&h->pin_out_dev, h->pin_out, 0));
            h->pin_out_cap = cap;
        }
        memcpy(h->pin_in, llr, in_bytes);
        int deferred = 0;
        h->lat_flag_bytes = h->pin_out_dev + out_bytes;
        rc = decode_impl(h, h->pin_in_dev, llr_f32, B, nullptr, L, h->pin_out_dev, nullptr, nullptr, nullptr, nullptr, 1, &deferred);
        h->lat_flag_bytes = nullptr;
        if (rc) return rc;
        HIP_TRY(hipStreamSynchronize(nullptr));
        bool any = !deferred;
        for (long i = 0; i < B && !any; ++i) any = h->pin_out[out_bytes + i] != 0;
        if (any && deferred) {
            if ((rc = decode_impl(h, h->pin_in_dev, llr_f32, B, nullptr, L, h->pin_out_dev, nullptr, nullptr, nullptr, nullptr, 2, nullptr))) return rc;
            HIP_TRY(hipStreamSynchronize(nullptr));
        }
        memcpy(out, h->pin_out, out_bytes);
        return POLAR_OK;
    }
    void *d_in;
    if (llr_f32) { if ((rc = h->d_f32.ensure((size_t)B * h->N))) return rc; d_in = h->d_f32.p; }
    else { if ((rc = h->d_in.ensure((size_t)B * h->N))) return rc; d_in = h->d_in.p; }
    if ((rc = h->d_out.ensure(out_bytes))) return rc;
    HIP_TRY(hipMemcpy(d_in, llr, in_bytes, hipMemcpyHostToDevice));
    int deferred = 0;
    if ((rc = decode_impl(h, d_in, llr_f32, B, nullptr, L, h->d_out.p, nullptr, nullptr, nullptr, nullptr, 1, &deferred))) return rc;
    if (deferred) {
        HIP_TRY(hipMemcpy(out, h->d_out.p, out_bytes, hipMemcpyDeviceToHost));          // (waits for the kernel)
        bool any = false;
        if (deferred == 1) {                 // list size 1: flag words
            const size_t nfw = (size_t)(B + 31) / 32;
            std::vector<unsigned int> fw(nfw);
            HIP_TRY(hipMemcpy(fw.data(), h->d_flag_words.p + 4, nfw * sizeof(unsigned int), hipMemcpyDeviceToHost));
            for (unsigned int w : fw) any |= (w != 0);
        } else {                             // small lists: flag bytes
            std::vector<uint8_t> fb((size_t)B);
            HIP_TRY(hipMemcpy(fb.data(), h->d_flags.p, (size_t)B, hipMemcpyDeviceToHost));
            for (uint8_t b : fb) any |= (b != 0);
        }
        if (!any) return POLAR_OK;
        if ((rc = decode_impl(h, d_in, llr_f32, B, nullptr, L, h->d_out.p, nullptr, nullptr, nullptr, nullptr, 2, nullptr))) return rc;
    }
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out, h->d_out.p, out_bytes, hipMemcpyDeviceToHost));
    return POLAR_OK;
}

int polar_decode_scl_llr_batch(polar_code_t *h, const double *llr, long B, int L, uint8_t *out) {
    return host_decode(h, llr, 0, B, L, out);
}

int polar_decode_scl_llr(polar_code_t *h, const double *llr, int L, uint8_t *out) {
    return polar_decode_scl_llr_batch(h, llr, 1, L, out);
}

// single-precision LLRs at the boundary: every float is widened (exactly) in the load stage of the first kernel that
// touches the channel values (ed_front_kernel / prefix_kernel / the layer-1 visits) — no staging copy
int polar_decode_scl_llr_batch_dev_f32(polar_code_t *h, const float *d_llr, long B, int L, uint8_t *d_out,
                                       double *d_pm, void *stream) {
    return decode_impl(h, d_llr, 1, B, nullptr, L, d_out, d_pm, stream, nullptr, nullptr);
}
int polar_decode_scl_llr_batch_f32(polar_code_t *h, const float *llr, long B, int L, uint8_t *out) {
    return host_decode(h, llr, 1, B, L, out);
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
    int grid = (int)std::min<long>(groups, (long)h->num_cu * 16);
    grid = (int)std::max<long>(1, std::min<long>(grid, (long)((24ull << 30) / ((size_t)N * 64 * 2 * sizeof(double)))));
    const size_t cwords = (N >= 128) ? (size_t)(N / 32 - 2) : 0;
    if ((rc = h->d_in.ensure((size_t)B * N * 2))) return rc;
    if ((rc = h->d_out.ensure((size_t)B * h->K))) return rc;
    if ((rc = h->d_llr_scr.ensure((size_t)grid * N * 64 * 2 + 64))) return rc;
    if ((rc = h->d_c_scr.ensure((size_t)grid * 2 * cwords * 64 + 64))) return rc;
    if ((rc = h->d_hist_scr.ensure((size_t)grid * h->W * 64 + 64))) return rc;
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
    int grid = (int)std::min<long>((B + 63) / 64, (long)h->num_cu * 16);
    grid = (int)std::max<long>(1, std::min<long>(grid, (long)((24ull << 30) / ((size_t)N * 64 * 4 * sizeof(double)))));
    if ((rc = h->d_in.ensure((size_t)B * N + (size_t)B * h->K))) return rc;
    if ((rc = h->d_llr_scr.ensure((size_t)grid * 4 * N * 64 + 64))) return rc;
    HIP_TRY(hipMemcpy(h->d_in.p, p1, (size_t)B * N * sizeof(double), hipMemcpyHostToDevice));
    PolarScP1Params p;
    p.n = h->n; p.N = N; p.K = h->K; p.B = B;
    p.p1 = h->d_in.p; p.out = h->d_in.p + (size_t)B * N;
    p.frozen = h->d_frozen.p; p.order = h->d_order.p; p.scr = h->d_llr_scr.p;
    HIP_TRY(polar_launch_sc_p1(p, grid, nullptr));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out, p.out, (size_t)B * h->K * sizeof(double), hipMemcpyDeviceToHost));
    return POLAR_OK;
}
int polar_decode_sc_p1(polar_code_t *h, const double *p1, double *out) { return polar_decode_sc_p1_batch(h, p1, 1, out); }

// ------------------------------------------------------------------------------------------
static void fill_enc(const polar_code *h, PolarEncodeParams &p) {
    memset(&p, 0, sizeof p);
    p.n = h->n; p.N = h->N; p.K = h->K; p.crc = h->crc;
    p.order = h->d_order.p; p.crcm = h->d_crcm.p;
    p.stride = 1;
    p.info_block_div = 100;
}

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

}  // extern "C" (first part)

// ---- PolarCode::get_bler_quick on 1..n GPUs ---------------------------------------------------------------------
namespace {

// RCCL, bound at run time (the library has no link-time dependency on it): the copy that sits next to the HIP
// runtime this process uses (PyTorch bundles both), else the ROCm one.
struct Rccl {
    void *lib = nullptr;
    int (*CommInitAll)(void **, int, const int *) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    int (*CommAbort)(void *) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    bool load() {
        if (lib) return true;
        std::vector<std::string> cand;
        Dl_info info;
        if (dladdr((void *)&hipGetDeviceCount, &info) && info.dli_fname) {
            std::string d(info.dli_fname);
            size_t k = d.rfind('/');
            if (k != std::string::npos) cand.push_back(d.substr(0, k + 1) + "librccl.so");
        }
        cand.push_back("librccl.so");
        cand.push_back("/opt/rocm/lib/librccl.so");
        for (const std::string &c : cand) {
            lib = dlopen(c.c_str(), RTLD_NOW | RTLD_GLOBAL);
            if (lib) break;
        }
        if (!lib) return false;
        CommInitAll = (decltype(CommInitAll))dlsym(lib, "ncclCommInitAll");
        CommDestroy = (decltype(CommDestroy))dlsym(lib, "ncclCommDestroy");
        CommAbort = (decltype(CommAbort))dlsym(lib, "ncclCommAbort");
        AllReduce = (decltype(AllReduce))dlsym(lib, "ncclAllReduce");
        GetErrorString = (decltype(GetErrorString))dlsym(lib, "ncclGetErrorString");
        return CommInitAll && CommDestroy && AllReduce;
    }
};
Rccl g_rccl;
constexpr int kNcclUint64 = 5, kNcclSum = 0;     // rccl.h: ncclUint64, ncclSum
std::atomic<int> g_comm_inits{0};                // test hook (polar_debug_comm_inits): ncclCommInitAll calls so far

// all worker threads of a round meet here before the collective: either every one of them enters ncclAllReduce or none does.
// Abortable: the watchdog of a round that takes too long releases everybody who waits here (wait() then returns false, now
// and for the rest of the context's life — a context whose round timed out is torn down, never reused).
struct HostBarrier {
    std::mutex m; std::condition_variable cv; int n, waiting = 0; unsigned gen = 0; bool aborted = false;
    explicit HostBarrier(int n_) : n(n_) {}
    bool wait() {
        std::unique_lock<std::mutex> lk(m);
        if (aborted) return false;
        const unsigned g = gen;
        if (++waiting == n) { waiting = 0; ++gen; cv.notify_all(); return true; }
        cv.wait(lk, [&] { return gen != g || aborted; });
        return gen != g;
    }
    void abort() {
        { std::lock_guard<std::mutex> lk(m); aborted = true; }
        cv.notify_all();
    }
};

}  // namespace

// streams, communicators and worker threads of a device list, owned by the handle (polar_code::multi)
struct MultiCtx {
    std::vector<int> devs;               // as listed by the caller
    std::vector<hipStream_t> streams;
    bool rccl = false;
    // The communicators (empty without RCCL). Workers read them, a worker that learns of a failed round aborts its own,
    // and the watchdog aborts what is left when a worker does not answer: every access goes through get_comm / take_comm
    // (a mutex; take_comm hands a communicator to exactly ONE caller, so none is aborted or destroyed twice).
    std::vector<void *> comms;
    std::mutex cm;
    void *get_comm(int d) { std::lock_guard<std::mutex> lk(cm); return d < (int)comms.size() ? comms[d] : nullptr; }
    void *take_comm(int d) {
        std::lock_guard<std::mutex> lk(cm);
        if (d >= (int)comms.size()) return nullptr;
        void *c = comms[d]; comms[d] = nullptr; return c;
    }
    // Persistent worker pool: one thread per device, created with the context and parked between rounds (round 3 created
    // and joined n_dev threads every round). run_all() hands every worker the same job and waits for all of them, with a
    // watchdog in three bounded steps when a round takes longer than `timeout_s`:
    //   1. SIGNAL: `abort_req` is raised and the host barrier aborted. Workers wait for their streams by polling
    //      (wait_stream), see the flag and abort their OWN communicator — which releases a stream stuck behind a collective
    //      a peer never entered or never finished; workers waiting in the barrier are released by its abort.
    //   2. after `grace_s`: a worker blocked INSIDE an RCCL call cannot poll; the communicators nobody took yet are aborted
    //      from the waiting thread (ncclCommAbort exists for that).
    //   3. after another `grace_s`: give up. `stuck` is set, the caller returns an error WITHOUT joining: the context, its
    //      threads and the handle's device contexts are leaked on purpose (a thread that never comes back from the driver
    //      cannot be cancelled), the handle refuses further multi-device calls.
    std::vector<std::thread> threads;
    std::mutex m;
    std::condition_variable cv_job, cv_done;
    std::function<void(int)> job;
    unsigned gen = 0;
    int pending = 0;
    bool quit = false, timed_out = false, stuck = false;
    std::atomic<bool> abort_req{false};
    std::unique_ptr<HostBarrier> bar;

    void start_workers(int n, bool force_threads) {
        bar.reset(new HostBarrier(n));
        if (n <= 1 && !force_threads) return;        // a single device runs on the calling thread (no watchdog then)
        for (int d = 0; d < n; ++d)
            threads.emplace_back([this, d] {
                unsigned seen = 0;
                for (;;) {
                    std::function<void(int)> f;
                    {
                        std::unique_lock<std::mutex> lk(m);
                        cv_job.wait(lk, [&] { return quit || gen != seen; });
                        if (quit) return;
                        seen = gen;
                        f = job;
                    }
                    f(d);
                    {
                        std::lock_guard<std::mutex> lk(m);
                        if (--pending == 0) cv_done.notify_all();
                    }
                }
            });
    }
    void abort_own(int d) {                          // (worker d, or the watchdog for whoever did not answer)
        void *c = take_comm(d);
        if (c && g_rccl.CommAbort) (void)g_rccl.CommAbort(c);
    }
    // Wait for a worker's stream without giving up the ability to react: hipStreamSynchronize cannot be interrupted, a
    // polling loop can — when the watchdog raises abort_req the worker aborts its own communicator and keeps waiting (the
    // aborted collective completes with an error, the stream drains).
    hipError_t wait_stream(int d) {
        if (threads.empty()) return hipStreamSynchronize(streams[d]);
        bool aborted_own = false;
        for (unsigned spins = 0;; ++spins) {
            const hipError_t q = hipStreamQuery(streams[d]);
            if (q != hipErrorNotReady) return q;
            if (abort_req.load(std::memory_order_relaxed) && !aborted_own) { abort_own(d); aborted_own = true; }
            if (spins < 2000) std::this_thread::yield();
            else std::this_thread::sleep_for(std::chrono::microseconds(50));
        }
    }
    void run_all(const std::function<void(int)> &f, long timeout_s, long grace_s) {
        const int n = (int)devs.size();
        if (threads.empty()) { for (int d = 0; d < n; ++d) f(d); return; }
        std::unique_lock<std::mutex> lk(m);
        job = f; pending = n; ++gen;
        cv_job.notify_all();
        auto done = [&] { return pending == 0; };
        if (timeout_s <= 0) { cv_done.wait(lk, done); return; }
        if (cv_done.wait_for(lk, std::chrono::seconds(timeout_s), done)) return;
        timed_out = true;
        abort_req.store(true);
        bar->abort();                                // step 1: signal
        if (cv_done.wait_for(lk, std::chrono::seconds(grace_s), done)) return;
        lk.unlock();
        for (int d = 0; d < n; ++d) abort_own(d);    // step 2: whatever no worker took
        lk.lock();
        if (cv_done.wait_for(lk, std::chrono::seconds(grace_s), done)) return;
        stuck = true;                                // step 3: bounded in every case
    }
    void stop_workers() {
        {
            std::lock_guard<std::mutex> lk(m);
            quit = true;
        }
        cv_job.notify_all();
        for (auto &t : threads) t.join();
        threads.clear();
    }
};

namespace {

void multi_release(polar_code *h, bool abort_comms) {
    MultiCtx *m = h->multi;
    if (!m) return;
    h->multi = nullptr;
    if (m->stuck) {
        // a worker never came back from the driver: nothing it may still touch is freed (MultiCtx::run_all step 3)
        for (auto &t : m->threads) t.detach();
        h->multi_poisoned = true;
        return;
    }
    m->stop_workers();
    int prev = -1;
    (void)hipGetDevice(&prev);
    for (size_t d = 0; d < m->comms.size(); ++d)
        if (void *c = m->take_comm((int)d)) {
            // after a failed round a rank may be stuck inside a collective: abort, do not wait for it
            if (abort_comms && g_rccl.CommAbort) (void)g_rccl.CommAbort(c);
            else (void)g_rccl.CommDestroy(c);
        }
    for (size_t d = 0; d < m->streams.size(); ++d)
        if (m->streams[d]) { (void)hipSetDevice(m->devs[d]); (void)hipStreamDestroy(m->streams[d]); }
    if (prev >= 0) (void)hipSetDevice(prev);
    delete m;
}
// the handle's tables on another device (owned by `h`, reused by later calls)
polar_code *clone_on_device(polar_code *h, int dev, bool fresh = false) {
    // fresh: a context of its own even when one exists for this device (test hook share_device)
    if (!fresh) {
        if (dev == h->device) return h;
        for (polar_code *c : h->clones) if (c->device == dev) return c;
    }
    polar_code *c = copy_ctx(h, dev);
    h->clones.push_back(c);
    return c;
}
// a copy of the handle's tables and settings bound to `dev`, with its own (not yet allocated) device state; the caller owns it
polar_code *copy_ctx(polar_code *h, int dev) {
    polar_code *c = new polar_code;
    c->n = h->n; c->N = h->N; c->K = h->K; c->crc = h->crc; c->eps = h->eps;
    c->frozen = h->frozen; c->order = h->order; c->bitrev = h->bitrev; c->crcm = h->crcm;
    c->W = h->W; c->info_rank = h->info_rank; c->crc_mask = h->crc_mask; c->sched = h->sched; c->ctl = h->ctl;
    c->sc_ops = h->sc_ops; c->sc_lat_ops = h->sc_lat_ops; c->sc_fold = h->sc_fold; c->weak_leaves = h->weak_leaves;
    c->device = dev;
    c->waves_per_cu = h->waves_per_cu; c->lds_log = h->lds_log; c->pipe = h->pipe; c->prefix_on = h->prefix_on; c->mode = h->mode;
    c->knobs = h->knobs;
    return c;
}

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
                if (!h->knobs.share_device) return fail(POLAR_E_ARG, "device %d listed twice", dev);
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
        m->start_workers(n_dev, h->knobs.force_workers);
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
    job->fail_dev = h->knobs.fail_device; job->fail_coll = h->knobs.fail_collective;       // (test hooks)
    job->stall_dev = h->knobs.stall_device; job->stall_ms = h->knobs.stall_ms;
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
        else if (d == J.fail_dev && J.step_no == 1) { rc = POLAR_E_DEVICE; msg = "injected failure (fail_device)"; }
        else {
            // (test hook: this worker does not answer for stall_ms in its second step — a hang outside every collective)
            if (d == J.stall_dev && J.step_no == 1) std::this_thread::sleep_for(std::chrono::milliseconds(J.stall_ms));
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
            if (d == J.fail_coll && J.step_no == 1) coll_failed = true;       // (test hook: the enqueue "fails" on this rank only)
            else if (J.rccl) {
                void *comm = mc->get_comm(d);
                if (!comm || g_rccl.AllReduce(c->d_mc_ctr.p, c->d_mc_ctr.p, (size_t)2 * P, kNcclUint64, kNcclSum, comm, st) != 0) coll_failed = true;
            }
            if (coll_failed) { rc = POLAR_E_DEVICE; msg = (d == J.fail_coll && J.step_no == 1) ? "injected failure (fail_collective)" : "ncclAllReduce failed"; ++J.n_failed_coll; }
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
    std::vector<unsigned long long> tot((size_t)2 * P);
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
        if (mc->stuck) break;                        // (the job's vectors may still be written by the worker that is stuck)
        // report the device that failed first-hand, not a peer that was merely told to stop
        for (int pass = 0; pass < 2 && !rc_all; ++pass)
            for (int d = 0; d < n_dev; ++d)
                if (job->rcs[d] && (pass == 1 || job->msgs[d].compare(0, 13, "round aborted") != 0)) { rc_all = job->rcs[d]; err_msg = "device " + std::to_string(devs[d]) + ": " + job->msgs[d]; break; }
        if (rc_all) break;
        std::fill(tot.begin(), tot.end(), 0ull);
        for (int d = 0; d < (rccl ? 1 : n_dev); ++d)
            for (int i = 0; i < 2 * P; ++i) tot[i] += job->host_ctr[d][i];
        if (reduce && reduce(reduce_user, (uint64_t *)tot.data(), 2 * P) != 0) { rc_all = POLAR_E_DEVICE; err_msg = "the counter reduction over the processes failed"; break; }
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
int polar_debug_comm_inits(void) { return g_comm_inits.load(); }
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
