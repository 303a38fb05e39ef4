// polar_host.h — INTERNAL header of the host side of libpolar_amd.so (the C-ABI of include/polar_amd.h): the handle
// (`struct polar_code`), its device buffers, and the functions the translation units share.
//
//   polar_handle.cpp      construction (the reference's constructor work), tables, device upload, getters / setters
//   polar_decode.cpp      kernel-family dispatch of decode_scl_llr (decode_impl), device-resident entry points, P1 paths, encoder
//   polar_hostpipe.cpp    host-pointer entry points: small-batch staging and the pipelined large-batch path
//   polar_montecarlo.cpp  get_bler_quick: device-side rounds, the pipelined-round scheduler, Monte-Carlo code construction
//   polar_multi.cpp       multi-device context: RCCL binding, worker threads, watchdog, per-device clones
//   polar_debug.cpp       measurement knobs (include/polar_amd_debug.h); fault injection only with -DPOLAR_TEST_HOOKS
#pragma once
#include <hip/hip_runtime.h>

#include <dlfcn.h>
#include <sched.h>
#include <sys/mman.h>
#include <unistd.h>

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

namespace polar_host {

extern thread_local std::string g_err;
int fail(int code, const char *fmt, ...);

#define HIP_TRY(expr)                                                                      \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess)                                                              \
            return fail(POLAR_E_DEVICE, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

extern std::atomic<unsigned long> g_allocs;     // hipMalloc calls of the handles' scratch buffers so far (polar_debug_get "allocs")

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

int usable_cpus();

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

// The caller's result array is usually FRESH (a MEX gateway, a std::vector, numpy: memory the process has never touched):
// every 4-KiB page of it takes a page fault on its first store, and the stores are the result copies at the END of the
// pipeline, on the calling thread's critical path (measured, round 6, 64 MiB of small pages: 6 ms of result copies instead of
// 0.8 ms per call — tools/fresh_out_probe.py). A few parked threads fault the pages in (MADV_POPULATE_WRITE: the contents are
// not touched; kernels without it: a locked add of 0 to one byte per page, atomic against the result copies) while the first
// chunks are copied in and decoded. Pages that are already resident cost a page-table walk. Persistent: creating the threads
// per call costs more than they save on the short calls.
struct PrefaultPool {
    std::vector<std::thread> threads;
    std::mutex m;
    std::condition_variable cv_job, cv_done;
    unsigned gen = 0;
    bool quit = false;
    uintptr_t lo = 0, hi = 0, pg = 4096;
    std::atomic<uintptr_t> next{0};
    int pending = 0;
    static constexpr uintptr_t kStep = (uintptr_t)2 << 20;      // (in pieces, in address order: the first result copies need the FIRST pages first)
    void work() {
        for (;;) {
            const uintptr_t q = next.fetch_add(kStep);
            if (q >= hi) return;
            const size_t len = (size_t)std::min(kStep, hi - q);
#ifdef MADV_POPULATE_WRITE
            if (madvise((void *)q, len, MADV_POPULATE_WRITE) == 0) continue;
#endif
            for (uintptr_t r = q; r < q + len; r += pg) __atomic_fetch_add((volatile unsigned char *)r, (unsigned char)0, __ATOMIC_RELAXED);
        }
    }
    void start_threads(int n) {
        pg = (uintptr_t)sysconf(_SC_PAGESIZE);
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
    // fault in [p, p + bytes) in the background; wait() before the memory may be handed back to the caller
    void start(void *p, size_t bytes) {
        const uintptr_t a = ((uintptr_t)p + pg - 1) & ~(pg - 1), b = ((uintptr_t)p + bytes) & ~(pg - 1);
        if (threads.empty() || b <= a + 64 * pg) return;
        {
            std::lock_guard<std::mutex> lk(m);
            lo = a; hi = b; next = a; pending = (int)threads.size(); ++gen;
        }
        cv_job.notify_all();
    }
    void wait() {
        std::unique_lock<std::mutex> lk(m);
        cv_done.wait(lk, [&] { return pending == 0; });
    }
    ~PrefaultPool() {
        { std::lock_guard<std::mutex> lk(m); quit = true; }
        cv_job.notify_all();
        for (auto &t : threads) t.join();
    }
};

}  // namespace polar_host
using namespace polar_host;

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
    std::unique_ptr<PrefaultPool> prefault;  // (parked threads that fault the caller's fresh result pages in)
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
        long host_fail_alloc = 0;      // (test hook) the staging slot with this number (1-based) cannot be allocated
        long host_prefault = 0;        // threads that fault the caller's (fresh) output pages in while the first chunks decode (0 = default, -1 = none)
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
    bool ctx_stuck = false;          //   ... and the worker that never came back was working on THIS context: it computes nothing any more
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

namespace polar_host {

// Every compute entry point runs on the handle's device (the one current at creation) and leaves the
// caller's current device as it found it.
struct DevGuard {
    int prev = -1;
    ~DevGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

inline int pow2ceil(int v) { int p = 1; while (p < v) p <<= 1; return p; }

// polar_handle.cpp
int derive_tables(polar_code *h);
int ensure_device(polar_code *h, DevGuard &dg);
void drop_clones(polar_code *h);          // per-device contexts and extra decode lanes carry a copy of the settings: every setter drops them
template <typename T>
int upload(DevBuf<T> &d, const std::vector<T> &v) {
    size_t n = v.size() ? v.size() : 1;
    int rc = d.ensure(n);
    if (rc) return rc;
    if (v.size()) HIP_TRY(hipMemcpy(d.p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return POLAR_OK;
}
// polar_decode.cpp
// phase (list size 1 with the one-codeword-per-wave kernel only): 0 = everything; 1 = the decode kernel alone — the caller
// looks at the flag words itself and runs phase 2 (work list + general kernel over the flagged codewords) only when one is
// set; *deferred reports whether phase 1 really left the fallback out
int decode_impl(polar_code *h, const void *d_llr, int llr_f32, long B, const unsigned int *n_dev, int L, uint8_t *d_out,
                double *d_pm, void *stream, void *ev_start, void *ev_stop, int phase = 0, int *deferred = nullptr);
bool use_sc_lat(const polar_code *h, long B);
void fill_enc(const polar_code *h, PolarEncodeParams &p);
// polar_hostpipe.cpp
void hostpipe_release(polar_code *h);
// polar_multi.cpp
void multi_release(polar_code *h, bool abort_comms);
polar_code *copy_ctx(polar_code *h, int dev);            // a copy of the handle's tables and settings bound to `dev`; the caller owns it
polar_code *clone_on_device(polar_code *h, int dev, bool fresh = false);

}  // namespace polar_host
