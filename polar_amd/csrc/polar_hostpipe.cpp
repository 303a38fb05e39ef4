// polar_hostpipe.cpp — the host-pointer entry points of decode_scl_llr (what a MEX gateway or the C++ class calls:
// PolarCode.cpp:130-148, PolarM/PolarCode.m:312-322): zero-copy staging of the smallest batches, one copy in / decode /
// one copy out in between, the chunked pipeline for large batches.
#include "polar_host.h"

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
void polar_host::hostpipe_release(polar_code *h) {
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
    auto free_slot = [&](int i) {
        if (hp->pin_in[i]) (void)hipHostFree(hp->pin_in[i]);
        if (hp->d_in[i]) (void)hipFree(hp->d_in[i]);
        if (hp->pin_out[i]) (void)hipHostFree(hp->pin_out[i]);
        if (hp->d_out[i]) (void)hipFree(hp->d_out[i]);
        hp->pin_in[i] = nullptr; hp->d_in[i] = nullptr; hp->pin_out[i] = nullptr; hp->d_out[i] = nullptr;
    };
    if (hp->in_cap < in_slot || hp->out_cap < out_slot) {
        for (int i = 0; i < HostPipe::kMaxSlots; ++i) free_slot(i);
        hp->in_cap = std::max(hp->in_cap, in_slot); hp->out_cap = std::max(hp->out_cap, out_slot);
    }
    // A slot is its four buffers or nothing: when one allocation fails (128 MiB of pinned memory is not always to be had) the
    // partial slot is released, the capacities forget what they promised, and the caller falls back to the unpipelined path
    // (POLAR_E_NOMEM here is not an error of the decode).
    for (int i = 0; i < R; ++i) {
        if (hp->pin_in[i] && hp->d_in[i] && hp->pin_out[i] && hp->d_out[i]) continue;
        free_slot(i);
        ++g_allocs;
        hipError_t e = (h->knobs.host_fail_alloc == i + 1) ? hipErrorOutOfMemory : hipHostMalloc(&hp->pin_in[i], hp->in_cap, hipHostMallocDefault);
        if (e == hipSuccess) e = hipMalloc(&hp->d_in[i], hp->in_cap);
        if (e == hipSuccess) e = hipHostMalloc((void **)&hp->pin_out[i], hp->out_cap, hipHostMallocDefault);
        if (e == hipSuccess) e = hipMalloc((void **)&hp->d_out[i], hp->out_cap);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            for (int j = 0; j < HostPipe::kMaxSlots; ++j) free_slot(j);
            hp->in_cap = hp->out_cap = 0;
            return fail(POLAR_E_NOMEM, "host pipeline: staging slot %d (%zu + %zu bytes, pinned and device): %s", i, in_slot, out_slot, hipGetErrorString(e));
        }
    }
    hp->R = R;
    for (int l = 1; l < lanes; ++l) {
        if (hp->ctx[l]) continue;
        polar_code *c = copy_ctx(h, h->device);
        DevGuard g2;
        int rc = ensure_device(c, g2);
        g2.prev = -1;
        if (rc) { const std::string msg = g_err; polar_destroy(c); g_err = msg; return rc; }
        hp->ctx[l] = c;
    }
    const int n_pf = h->knobs.host_prefault < 0 ? 0 : h->knobs.host_prefault > 0 ? (int)h->knobs.host_prefault : std::max(1, std::min(2, usable_cpus() / 4));
    if ((hp->prefault ? (int)hp->prefault->threads.size() : 0) != n_pf) {
        hp->prefault.reset();
        if (n_pf > 0) { hp->prefault.reset(new PrefaultPool); hp->prefault->start_threads(n_pf); }
    }
    if (!hp->pool || (int)hp->pool->threads.size() != threads - 1) {
        hp->pool.reset(new CopyPool);
        hp->pool->start(threads - 1);
    }
    return POLAR_OK;
}

// *no_staging is set when the call failed BEFORE anything was decoded because the staging slots / decode lanes could not be set
// up (host_decode then takes the unpipelined path)
static int host_decode_pipelined(polar_code_t *h, const void *llr, int llr_f32, long B, int L, uint8_t *out, long chunk_cw, int lanes, int threads, bool ramp, bool *no_staging) {
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
    if (rc) { *no_staging = true; return rc; }
    HostPipe *hp = h->hpipe;
    hp->last_chunks = n_chunks; hp->last_chunk_cw = C; hp->last_lanes = lanes; hp->last_threads = threads;
    hp->us_copy_in = hp->us_wait = hp->us_copy_out = 0;
    // (the handle's scratch may still be in use by work the caller put on the null stream through this handle)
    HIP_TRY(hipStreamSynchronize(nullptr));
    // (the caller's result pages are faulted in behind the first chunks; every way out waits for the helper threads first)
    struct PrefaultWait { PrefaultPool *p; ~PrefaultWait() { if (p) p->wait(); } } pf_wait{hp->prefault.get()};
    if (hp->prefault) hp->prefault->start(out, (size_t)B * row_out);
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
        // From where the pipeline pays (round 6, tools/host_pipe_crossover.py, profiles/r06/host_pipe_crossover.json): n chunks on k lanes
        // are n / k launch latencies of 4 .. 8 ms for the list kernels, so while the whole batch fits the device's resident waves and
        // its copy takes less than about one launch, ONE copy in, ONE launch and one copy out is faster — 2048 .. 16384 codewords of
        // N = 2048 at L = 4: 5.0 .. 10.5 ms against 16.0 .. 14.9 ms pipelined (the 32-MiB rule of round 5), L = 32 up to 8192: 5.8 ..
        // 13.2 against 15.3 .. 17.7 ms. Measured crossovers: L = 1 32 MiB; L = 2 (the LLR-domain 2-lane kernel: the slowest launches)
        // 1 GiB; L = 3 .. 8 half a GiB of LLRs or one full round of resident waves, whichever comes first (N = 1024, L = 8, 65536
        // float rows = 256 MiB = two rounds: 12.0 ms pipelined, 17.7 in one copy); larger lists half a GiB or two rounds.
        const long resident_cw = (long)h->num_cu * 16 * (64 / pow2ceil(L));
        const bool pays = L == 1 ? in_bytes >= ((size_t)32 << 20) : L == 2 ? in_bytes >= ((size_t)1 << 30)
                                 : (in_bytes >= ((size_t)512 << 20) || B >= (L <= 8 ? 1 : 2) * resident_cw);
        const size_t min_bytes = kn.host_pipe_min_bytes > 0 ? (size_t)kn.host_pipe_min_bytes : (pays ? 0 : ~(size_t)0);
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
            if (B > chunk_cw) {
                bool no_staging = false;
                rc = host_decode_pipelined(h, llr, llr_f32, B, L, out, chunk_cw, lanes, threads, ramp, &no_staging);
                // (no staging memory: the batch is decoded by the unpipelined path below, as round 4 decoded every batch)
                if (!no_staging) return rc;
                if (h->hpipe) h->hpipe->last_chunks = -1;          // (polar_debug_get "host_chunks" = -1: the fallback was taken)
            }
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

int polar_decode_scl_llr_batch_f32(polar_code_t *h, const float *llr, long B, int L, uint8_t *out) {
    return host_decode(h, llr, 1, B, L, out);
}

