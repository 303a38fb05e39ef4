// polar_debug.cpp — measurement and test hooks (include/polar_amd_debug.h; not part of the reference's surface and not
// declared by include/polar_amd.h). The product library carries the measurement knobs only; the fault-injection keys
// (fail_device, fail_collective, stall_device, stall_ms, share_device, force_workers) exist only in the library built with
// -DPOLAR_TEST_HOOKS (libpolar_amd_test.so, what the GPU tests of the failure protocol load).
#include "polar_multi.h"
#include "polar_amd_debug.h"

extern "C" {

int polar_debug_weak_leaves(const polar_code_t *h) { return polar_get_weak_leaves(h); }

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
#ifdef POLAR_TEST_HOOKS
    else if (s == "host_fail_alloc") k.host_fail_alloc = value;
    else if (s == "share_device") { k.share_device = value != 0; multi_release(h, false); }
    else if (s == "fail_device") k.fail_device = (int)value;
    else if (s == "fail_collective") k.fail_collective = (int)value;
#endif
    else if (s == "lat_max_b") k.lat_max_b = value;
    else if (s == "host_pipe_min_bytes") k.host_pipe_min_bytes = value;
    else if (s == "host_chunk_bytes") k.host_chunk_bytes = value;
    else if (s == "host_lanes") { if (value < 0 || value > HostPipe::kMaxLanes) return fail(POLAR_E_ARG, "host_lanes must be 0 (default) or 1 .. %d", HostPipe::kMaxLanes); k.host_lanes = value; }
    else if (s == "host_ramp") k.host_ramp = value;
    else if (s == "host_prefault") { if (value < -1 || value > 16) return fail(POLAR_E_ARG, "host_prefault must be -1 (none), 0 (default) or 1 .. 16 threads"); k.host_prefault = value; }
    else if (s == "host_threads") { if (value < 0) return fail(POLAR_E_ARG, "host_threads must be >= 0"); k.host_threads = value; }
    else if (s == "multi_grace_s") { if (value < 0) return fail(POLAR_E_ARG, "multi_grace_s must be >= 0"); k.multi_grace_s = value; }
#ifdef POLAR_TEST_HOOKS
    else if (s == "force_workers") { k.force_workers = value != 0; multi_release(h, false); }
    else if (s == "stall_device") k.stall_device = (int)value;
    else if (s == "stall_ms") { if (value < 0) return fail(POLAR_E_ARG, "stall_ms must be >= 0"); k.stall_ms = value; }
#endif
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
    if (s == "test_hooks") {
#ifdef POLAR_TEST_HOOKS
        return 1;
#else
        return 0;
#endif
    }
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
int polar_debug_comm_inits(void) { return g_comm_inits.load(); }

}  // extern "C"
