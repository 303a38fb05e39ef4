/* polar_amd_debug.h — measurement knobs and test hooks of libpolar_amd.so. NOT part of the drop-in interface
 * (include/polar_amd.h does not include this file; nothing here has a counterpart in the reference's PolarCode class):
 * tools/, bench.py's diagnostic records and tests/ use it.
 *
 * Two libraries are built from the same sources (polar_amd/build.py):
 *   libpolar_amd.so       the product: the MEASUREMENT knobs below only. A fault-injection key is an unknown key.
 *   libpolar_amd_test.so  the same code compiled with -DPOLAR_TEST_HOOKS: additionally the FAULT-INJECTION keys, which make a
 *                         Monte-Carlo sweep fail, stall or share a device on purpose (tests of the failure protocol and of the
 *                         multi-device paths on a single-GPU box load this one).
 */
#ifndef POLAR_AMD_DEBUG_H
#define POLAR_AMD_DEBUG_H
#include "polar_amd.h"
#ifdef __cplusplus
extern "C" {
#endif

/* polar_debug_set(h, key, value): negative return = error (unknown key, value out of range).
 * Measurement knobs (both libraries; results never depend on them):
 *   "mode_override" -1|0|1|2      replaces polar_set_mode's value (-1 = none); POLAR_MODE in the environment at creation
 *   "sc_no_fold", "no_tables", "no_fuse_front", "no_prefix"     alternative (older) forms of single passes, for A/B timing
 *   "no_rccl", "force_rccl"       counter reduction of the single-process multi-device driver (drops its cached context)
 *   "lat_max_b"                   largest batch that takes the one-codeword-per-wave kernels (0 = default, -1 = never)
 *   "host_pipe_min_bytes", "host_chunk_bytes", "host_lanes", "host_threads", "host_ramp", "host_prefault"
 *                                 the pipelined host-pointer path (0 = default everywhere)
 *   "multi_timeout_s", "multi_grace_s"    watchdog of a multi-device step and its grace periods
 * Fault injection (libpolar_amd_test.so only; no environment form):
 *   "share_device"                one GPU may be listed several times in a device list (separate contexts, host-side sum)
 *   "force_workers"               worker threads (and so the watchdog) even with one device
 *   "fail_device" = d, "fail_collective" = d    worker d fails in its second step before / after the barrier (-1 = off)
 *   "stall_device" = d, "stall_ms"              worker d sleeps in its second step before it launches anything
 * Like every entry point, the hooks must not run concurrently with another call on the same handle. Every setter drops
 * the handle's per-device contexts and extra decode lanes (they carry a copy of the knobs). */
int polar_debug_set(polar_code_t *h, const char *key, long value);
/* polar_debug_get(h, key): "allocs" (hipMalloc calls of all handles' scratch so far), "comm_inits", "weak_leaves",
 * "mode_override", "last_rounds", "last_round_max_per_device", "worker_threads_started", "multi_poisoned",
 * "round_us_first|min|median|max|count" (steps of the handle's last sweep), "host_chunks", "host_chunk_cw", "host_lanes",
 * "host_threads", "host_us_copy_in|wait|copy_out|total" (the last pipelined host-pointer call), "test_hooks" (1 in the test
 * build); -1 = unknown key. */
long polar_debug_get(const polar_code_t *h, const char *key);
/* number of ncclCommInitAll calls made by this library so far (the communicators of a device list are cached on the handle) */
int polar_debug_comm_inits(void);
/* older name of polar_get_weak_leaves (include/polar_amd.h) */
int polar_debug_weak_leaves(const polar_code_t *h);
/* instrumented development builds (POLAR_DEFS, polar_amd/build.py): where their kernels leave their counters */
void *polar_debug_scratch_ptr(polar_code_t *h);

#ifdef __cplusplus
}
#endif
#endif /* POLAR_AMD_DEBUG_H */
