"""GPU, TEST BUILD of the library (libpolar_amd_test.so, fixture `hooks_lib`): the multi-device paths of get_bler_quick exercised
on a single-GPU box and the failure protocol of a multi-device step — one GPU standing in for several ("share_device"), injected
failures before / after the barrier ("fail_device", "fail_collective"), stalls inside and beyond the watchdog's grace periods
("stall_device", "stall_ms", "force_workers"). These hooks are fault injection: they exist only in the test build
(include/polar_amd_debug.h); the product library rejects their keys (tests/test_abi.py)."""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("hooks_lib")]


def _pair(n, K, crc):
    import polar_amd
    from oracle_lib import Oracle
    o = Oracle(n, K, 0.32, crc, srand=1)
    C.CDLL(None).srand(C.c_uint(1))
    return o, polar_amd.PolarCode(n, K, 0.32, crc)


def test_multi_device_partition_with_one_gpu_standing_in_for_several(built_lib, oracle_built, monkeypatch):
    """The single-process multi-GPU driver (per-device table clones, one worker thread and stream per device, trials
    done + d, done + d + n_dev, ..., counter sum) with ONE GPU listed two / three / five times (test hook
    "share_device"; RCCL cannot have two ranks on a device, so the counters are summed on the host): the
    estimates must be those of the single-device run — the union of the trials does not depend on the partition."""
    o, g = _pair(8, 128, 8)
    ebno, Ls = [0.5, 2.0], [1, 4, 8]
    want, want_ber = g.get_bler_quick(ebno, Ls, max_runs=1500, max_err=60, seed=77, batch=250, return_ber=True)
    g.debug_set("share_device", 1)
    for devs in ([0, 0], [0, 0, 0], [0] * 5):
        got, got_ber = g.get_bler_quick(ebno, Ls, max_runs=1500, max_err=60, seed=77, batch=250, return_ber=True, devices=devs)
        assert np.array_equal(np.asarray(got), np.asarray(want)), devs
        assert np.array_equal(np.asarray(got_ber), np.asarray(want_ber)), devs
    # a setter after the per-device contexts exist: they must not keep the old CRC matrix
    m = g.crc_matrix.copy()
    m[:, ::3] ^= 1
    g.crc_matrix = m
    want2 = g.get_bler_quick(ebno, [8], max_runs=500, max_err=10**6, seed=5, batch=500)
    got2 = g.get_bler_quick(ebno, [8], max_runs=500, max_err=10**6, seed=5, batch=500, devices=[0, 0, 0])
    assert np.array_equal(np.asarray(got2), np.asarray(want2))
    assert not np.array_equal(np.asarray(want2), np.asarray(want)[2:3])     # (the matrix matters: different estimates)


def test_a_failing_device_aborts_the_round_for_every_device(built_lib, monkeypatch):
    """Device 1 of three fails in the second round (test hook): nobody enters the round's counter reduction (a lone
    rank skipping ncclAllReduce used to leave the others blocked in it for good), the call returns the error, and the
    handle works again afterwards."""
    import polar_amd
    o, g = _pair(8, 128, 8)
    g.debug_set("share_device", 1)
    want = g.get_bler_quick([1.0], [1, 4], max_runs=900, max_err=10**6, seed=3, batch=300, devices=[0, 0, 0])
    g.debug_set("fail_device", 1)
    with pytest.raises(polar_amd.PolarError, match=r"device 0: injected failure \(fail_device\)"):
        g.get_bler_quick([1.0], [1, 4], max_runs=900, max_err=10**6, seed=3, batch=300, devices=[0, 0, 0])
    g.debug_set("fail_device", -1)
    again = g.get_bler_quick([1.0], [1, 4], max_runs=900, max_err=10**6, seed=3, batch=300, devices=[0, 0, 0])
    assert np.array_equal(np.asarray(again), np.asarray(want))


def test_a_failing_collective_enqueue_aborts_before_anyone_synchronises(built_lib):
    """Worker 2 of three gets past the barrier that precedes the counter reduction and THEN fails (its collective enqueue,
    test hook "fail_collective"): its peers have their share of the collective on their streams and would wait in
    hipStreamSynchronize for a rank that never arrives. Every worker meets again after the enqueue and, since one failed,
    aborts its own communicator before synchronising; the call returns the error (naming the device that failed first-hand)
    and the next call rebuilds communicators and worker threads. Host-sum form with a shared device, and the RCCL form with
    one rank (the communicator is aborted and a new ncclCommInitAll happens on the next call)."""
    import polar_amd
    o, g = _pair(8, 128, 8)
    g.debug_set("share_device", 1)
    want = g.get_bler_quick([1.0], [1, 4], max_runs=900, max_err=10**6, seed=3, batch=300, devices=[0, 0, 0])
    t0 = g.debug_get("worker_threads_started")
    g.debug_set("fail_collective", 2)
    with pytest.raises(polar_amd.PolarError, match=r"injected failure \(fail_collective\)"):
        g.get_bler_quick([1.0], [1, 4], max_runs=900, max_err=10**6, seed=3, batch=300, devices=[0, 0, 0])
    g.debug_set("fail_collective", -1)
    again = g.get_bler_quick([1.0], [1, 4], max_runs=900, max_err=10**6, seed=3, batch=300, devices=[0, 0, 0])
    assert np.array_equal(np.asarray(again), np.asarray(want))
    assert g.debug_get("worker_threads_started") == t0 + 3          # the pool of the failed context was torn down, one new pool
    g.debug_set("share_device", 0)
    g.debug_set("force_rccl", 1)
    one = g.get_bler_quick([1.0], [1, 4], max_runs=900, max_err=10**6, seed=3, batch=300, devices=[0])
    assert g.last_used_rccl and np.array_equal(np.asarray(one), np.asarray(want))
    n0 = g.debug_get("comm_inits")
    g.debug_set("fail_collective", 0)
    with pytest.raises(polar_amd.PolarError, match="fail_collective"):
        g.get_bler_quick([1.0], [1, 4], max_runs=900, max_err=10**6, seed=3, batch=300, devices=[0])
    g.debug_set("fail_collective", -1)
    two = g.get_bler_quick([1.0], [1, 4], max_runs=900, max_err=10**6, seed=3, batch=300, devices=[0])
    assert g.last_used_rccl and np.array_equal(np.asarray(two), np.asarray(want))
    assert g.debug_get("comm_inits") == n0 + 1


def test_a_round_that_takes_too_long_is_aborted_by_the_watchdog_and_the_next_call_works(built_lib):
    """Worker 1 of three does not answer for 2.5 s in its second round (test hook "stall_device": a hang OUTSIDE every
    collective — the watchdog of round 4 only covered hangs inside one, and its second wait had no bound). Watchdog 1 s:
    step 1 raises the abort flag and aborts the host barrier — the two peers waiting there are released and report "round
    aborted: watchdog" —, the sleeper wakes up inside the grace period, sees the flag, stays out of the reduction; the call
    returns POLAR_E_DEVICE in about the stall time, nothing is leaked, the next call rebuilds the context and gives the
    undisturbed counters. Then the same with ONE device through RCCL and a worker thread forced (force_workers): the
    worker aborts its own communicator, the next call makes a new one."""
    import time
    import polar_amd
    o, g = _pair(8, 128, 8)
    g.debug_set("share_device", 1)
    args = dict(max_runs=900, max_err=10**6, seed=3, batch=300)
    want = g.get_bler_quick([1.0], [1, 4], devices=[0, 0, 0], **args)
    g.debug_set("multi_timeout_s", 1); g.debug_set("multi_grace_s", 20)
    g.debug_set("stall_device", 1); g.debug_set("stall_ms", 2500)
    t = time.perf_counter()
    with pytest.raises(polar_amd.PolarError, match="exceeded the watchdog"):
        g.get_bler_quick([1.0], [1, 4], devices=[0, 0, 0], **args)
    dt = time.perf_counter() - t
    assert 2.0 < dt < 10.0 and g.debug_get("multi_poisoned") == 0
    g.debug_set("stall_device", -1)
    g.debug_set("multi_timeout_s", 1800)
    assert np.array_equal(np.asarray(g.get_bler_quick([1.0], [1, 4], devices=[0, 0, 0], **args)), np.asarray(want))
    # one device, RCCL, a worker thread of its own
    g.debug_set("share_device", 0); g.debug_set("force_rccl", 1); g.debug_set("force_workers", 1)
    g.debug_set("multi_timeout_s", 1800)         # (the communicator's first collective sets its connections up: not under a 1-s watchdog)
    one = g.get_bler_quick([1.0], [1, 4], devices=[0], **args)
    assert g.last_used_rccl and np.array_equal(np.asarray(one), np.asarray(want)) and g.debug_get("worker_threads_started") >= 1
    n0 = g.debug_get("comm_inits")
    g.debug_set("multi_timeout_s", 1)
    g.debug_set("stall_device", 0)
    with pytest.raises(polar_amd.PolarError, match="exceeded the watchdog"):
        g.get_bler_quick([1.0], [1, 4], devices=[0], **args)
    g.debug_set("stall_device", -1)
    g.debug_set("multi_timeout_s", 1800)
    two = g.get_bler_quick([1.0], [1, 4], devices=[0], **args)
    assert g.last_used_rccl and np.array_equal(np.asarray(two), np.asarray(want)) and g.debug_get("comm_inits") == n0 + 1


def test_a_worker_that_never_answers_costs_the_handle_not_the_caller(built_lib):
    """The stall outlasts the watchdog AND both grace periods (1 s each): the call still returns — after about three seconds,
    with the error — instead of waiting for ever (round 4: unbounded second wait), the handle refuses further Monte-Carlo
    calls and frees nothing (a thread that does not come back from the driver cannot be cancelled: what it may still touch is
    leaked on purpose), and decoding through the handle still works."""
    import time
    import polar_amd
    o, g = _pair(8, 128, 8)
    g.debug_set("share_device", 1)
    args = dict(max_runs=900, max_err=10**6, seed=3, batch=300)
    g.get_bler_quick([1.0], [1, 4], devices=[0, 0], **args)
    g.debug_set("multi_timeout_s", 1); g.debug_set("multi_grace_s", 1)
    g.debug_set("stall_device", 1); g.debug_set("stall_ms", 6000)
    t = time.perf_counter()
    with pytest.raises(polar_amd.PolarError, match="never returned"):
        g.get_bler_quick([1.0], [1, 4], devices=[0, 0], **args)
    assert time.perf_counter() - t < 5.0 and g.debug_get("multi_poisoned") == 1
    with pytest.raises(polar_amd.PolarError, match="never returned"):
        g.get_bler_quick([1.0], [1], **args)
    llr, _ = o.synth_llr(5, 0, 16, o.snr_sqrt_linear(2.0))
    assert (g.decode_scl_llr(llr, 4) == o.decode_scl_llr(llr, 4)).all()     # (the stuck worker had a per-device copy, not the handle)
    g.debug_set("lat_max_b", -1)                                            # setters leave the leaked per-device contexts alone
    assert (g.decode_scl_llr(llr, 4) == o.decode_scl_llr(llr, 4)).all()
    # the same with the worker that works on the HANDLE'S OWN context (device 0 of the list) stuck: the handle computes nothing
    # any more — its scratch may still be in that worker's hands (round-5 advisor)
    _, g2 = _pair(8, 128, 8)
    g2.debug_set("share_device", 1)
    g2.get_bler_quick([1.0], [1, 4], devices=[0, 0], **args)
    g2.debug_set("multi_timeout_s", 1); g2.debug_set("multi_grace_s", 1)
    g2.debug_set("stall_device", 0); g2.debug_set("stall_ms", 6000)
    with pytest.raises(polar_amd.PolarError, match="never returned"):
        g2.get_bler_quick([1.0], [1, 4], devices=[0, 0], **args)
    with pytest.raises(polar_amd.PolarError, match="accepts no further calls"):
        g2.decode_scl_llr(llr, 4)
    time.sleep(5.0)              # (let the sleepers finish their rounds on the leaked contexts before the process goes on)
    g.close(); g2.close()


def test_worker_threads_live_on_the_handle_between_calls(built_lib):
    """One thread per device, created with the device list's context and parked between rounds and calls (round 3 created
    and joined n_dev threads per ROUND)."""
    o, g = _pair(8, 128, 0)
    g.debug_set("share_device", 1)
    a, ca = g.get_bler_quick([1.0, 3.0], [1], max_runs=4000, max_err=10**6, seed=4, batch=500, devices=[0] * 4, return_counters=True)
    assert ca["rounds"] == 8 and g.debug_get("worker_threads_started") == 4
    b = g.get_bler_quick([1.0, 3.0], [1], max_runs=4000, max_err=10**6, seed=4, batch=500, devices=[0] * 4)
    assert g.debug_get("worker_threads_started") == 4 and np.array_equal(a, b)
    g.get_bler_quick([1.0], [1], max_runs=500, max_err=10**6, seed=4, batch=500, devices=[0] * 2)     # another list: another pool
    assert g.debug_get("worker_threads_started") == 6


def test_automatic_rounds_grow_with_the_device_count(built_lib):
    """batch = 0: a round is capped at 262144 trials PER DEVICE (round 3 capped it over all devices: at 8 GPUs each got
    32768 per round — less than one resident round of the list-size-1 kernel). Five contexts on one GPU, a short code: the
    last rounds hand every context 262144 trials; one device alone reaches the same cap; and the estimates of the two runs
    are those of the same trials whenever the run counts agree (no early stop here)."""
    o, g = _pair(6, 32, 0)
    total = 3 * 5 * 262144
    one, c1 = g.get_bler_quick([7.0], [1], max_runs=total, max_err=50000, seed=8, return_counters=True)
    assert g.debug_get("last_round_max_per_device") == 262144
    g.debug_set("share_device", 1)
    five, c5 = g.get_bler_quick([7.0], [1], max_runs=total, max_err=50000, seed=8, devices=[0] * 5, return_counters=True)
    assert g.debug_get("last_round_max_per_device") == 262144
    assert c5["rounds"] < c1["rounds"]
    assert int(c1["run"][0, 0]) == int(c5["run"][0, 0]) == total and int(c1["err"][0, 0]) == int(c5["err"][0, 0]) > 0


def test_bicm_sweep_sharded_over_devices_equals_one_device(built_lib, oracle_built):
    """BASELINE configuration 5's shape behind the native multi-device entry point (polar_get_bler_quick_multi_ex with an
    ASK Gray constellation: PolarM/main_MC_CC_Comparison.m:44-119): the counters of three contexts equal those of one
    device, and those of the step-wise engine (polar_mc_batch_bicm) the multi-process driver uses."""
    o, g = _pair(8, 128, 0)
    snr, Ls = [9.0, 11.0, 13.0], [1, 8]
    want, cw = g.get_bler_quick(snr, Ls, max_runs=1200, max_err=80, seed=21, batch=300, constellation="ask16-gray", return_counters=True)
    assert 0 < want[1, 2] < want[1, 0] <= 1
    g.debug_set("share_device", 1)
    got, cg = g.get_bler_quick(snr, Ls, max_runs=1200, max_err=80, seed=21, batch=300, constellation="ask16-gray",
                               devices=[0, 0, 0], return_counters=True)
    assert np.array_equal(cw["err"], cg["err"]) and np.array_equal(cw["run"], cg["run"]) and cw["rounds"] == cg["rounds"]
    from polar_amd.montecarlo import get_bler_quick_sharded
    eng = lambda seed, t0, T, stride, ax, L_, en, e, r: g.mc_batch_bicm("ask16-gray", seed, t0, T, stride, ax, L_, en, e, r)
    bler, err, run = get_bler_quick_sharded(eng, snr, Ls, max_runs=1200, max_err=80, seed=21, global_batch=300)
    assert np.array_equal(err, cw["err"]) and np.array_equal(run, cw["run"])
    # the BPSK sweep is a different workload (same code, Eb/N0 axis)
    other = g.get_bler_quick(snr, Ls, max_runs=300, max_err=80, seed=21, batch=300)
    assert not np.array_equal(other, want)



def test_a_rank_that_fails_locally_takes_every_rank_out_of_the_sweep_together(built_lib):
    """One process per GPU (polar_get_bler_quick_rank): rank 1's own step fails (injected) in its second step. It still makes
    the step's collective reduce call — with a failure flag in the last counter — so the other ranks are not left waiting in
    their all-reduce (round-5 advisor: they blocked for good): all three return an error from the SAME step, the failing rank
    its own reason, the others that a peer failed."""
    import threading
    import polar_amd
    world = 3
    codes = [_pair(8, 128, 8)[1] for _ in range(world)]
    codes[1].debug_set("fail_device", 0)                 # (its single worker, device index 0 of its own list, fails in step 2)
    bar = threading.Barrier(world)
    acc, lock = {}, threading.Lock()
    calls = [0] * world
    errors = [None] * world

    def reduce_for(rank):
        def reduce(a):
            calls[rank] += 1
            bar.wait(timeout=60)
            with lock:
                acc.setdefault("sum", np.zeros_like(a))
                acc["sum"] += a
            bar.wait(timeout=60)
            a[:] = acc["sum"]
            if bar.wait(timeout=60) == 0:
                acc.pop("sum")
            bar.wait(timeout=60)
        return reduce

    def run(rank):
        try:
            codes[rank].get_bler_quick_rank([0.5, 1.5], [1, 4], rank, world, reduce_for(rank), max_runs=3000, max_err=10**9, seed=9, batch=300)
            errors[rank] = "returned normally"
        except polar_amd.PolarError as ex:
            errors[rank] = str(ex)
    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    [t.start() for t in th]
    [t.join(timeout=120) for t in th]
    assert not any(t.is_alive() for t in th), "a rank is still waiting in the reduction"
    assert calls[0] == calls[1] == calls[2] == 2, calls                     # the failing step's reduction was made by everybody
    assert "injected failure (fail_device)" in errors[1], errors
    assert "another process of the sweep reported a failure" in errors[0] and "another process" in errors[2], errors
    codes[1].debug_set("fail_device", -1)
    # the handles are usable again (nothing was left half-way)
    b = codes[0].get_bler_quick([1.0], [1], max_runs=300, max_err=10**9, seed=9, batch=300)
    assert 0 < b[0, 0] < 1


def test_staging_that_cannot_be_allocated_falls_back_to_the_single_copy_path(built_lib, oracle_built):
    """The pipelined host-pointer path needs pinned and device staging slots (hundreds of MiB): when one of them cannot be
    allocated (injected: slot 2) nothing half-built is kept (round-5 advisor: a later call ran with NULL buffers) and the batch
    is decoded by the unpipelined path — same bits; the next call, with memory available again, is pipelined."""
    o, g = _pair(9, 256, 8)
    llr = o.synth_llr(3, 0, 4096, o.snr_sqrt_linear(1.5))[0]
    want = g.decode_scl_llr(llr, 4)
    assert (want[:64] == o.decode_scl_llr(llr[:64], 4)).all()
    g.debug_set("host_pipe_min_bytes", 1 << 20)
    g.debug_set("host_chunk_bytes", 1 << 22)
    g.debug_set("host_fail_alloc", 2)
    got = g.decode_scl_llr(llr, 4)
    assert (got == want).all() and g.debug_get("host_chunks") == -1          # (-1: the fallback was taken)
    g.debug_set("host_fail_alloc", 0)
    got = g.decode_scl_llr(llr, 4)
    assert (got == want).all() and g.debug_get("host_chunks") >= 4
