// polar_multi.h — INTERNAL: the multi-device context of get_bler_quick (one host process driving several GPUs): RCCL bound at
// run time, the abortable host barrier, and the per-handle context (streams, communicators, persistent worker threads,
// watchdog). Used by polar_montecarlo.cpp (the scheduler) and polar_multi.cpp (lifetime).
#pragma once
#include "polar_host.h"

namespace polar_host {

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
extern Rccl g_rccl;
constexpr int kNcclUint64 = 5, kNcclSum = 0;     // rccl.h: ncclUint64, ncclSum
extern std::atomic<int> g_comm_inits;           // (polar_debug_comm_inits): ncclCommInitAll calls so far

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

}  // namespace polar_host

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
    std::unique_ptr<std::atomic<char>[]> busy;       // busy[d]: worker d is inside a job (who is stuck when a round is given up)

    void start_workers(int n, bool force_threads) {
        bar.reset(new HostBarrier(n));
        busy.reset(new std::atomic<char>[n]);
        for (int d = 0; d < n; ++d) busy[d] = 0;
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
                    busy[d] = 1;
                    f(d);
                    busy[d] = 0;
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
