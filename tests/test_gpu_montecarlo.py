"""GPU: the Monte-Carlo engine behind get_bler_quick (PolarCode.cpp:658-785; PolarM/PolarCode.m:781-850) —
the device-side rounds, PolarM's `ber` output, the early-stop rounds, and the single-process multi-GPU entry point
(polar_get_bler_quick_multi) driven through RCCL with one device."""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _pair(n, K, crc):
    import polar_amd
    from oracle_lib import Oracle
    o = Oracle(n, K, 0.32, crc, srand=1)
    C.CDLL(None).srand(C.c_uint(1))
    return o, polar_amd.PolarCode(n, K, 0.32, crc)


def test_ber_is_bit_errors_per_run_as_polarm(built_lib, oracle_built):
    """One (L, Eb/N0) point, nothing stops early: every trial is simulated, so
    ber = sum(differing info bits) / runs and bler = block errors / runs, both computable from the oracle's
    decodes of the same counter-based trials (PolarM/PolarCode.m:836-848: per run, not per bit)."""
    o, g = _pair(8, 128, 4)
    T, L, ebno, seed = 300, 4, 1.0, 77
    llr, info = o.synth_llr(seed, 0, T, o.snr_sqrt_linear(ebno))
    dec = o.decode_scl_llr(llr, L)
    nd = (dec != info).sum(axis=1)
    bler, ber = g.get_bler_quick([ebno], [L], max_runs=T, max_err=10**9, seed=seed, batch=T, return_ber=True)
    assert bler[0, 0] == (nd > 0).sum() / T
    assert ber[0, 0] == nd.sum() / T
    assert ber[0, 0] > bler[0, 0] > 0
    # step-wise form with the bit-error accumulator
    e, b, r = (np.zeros((1, 1), np.uint64) for _ in range(3))
    g.mc_batch_ber(seed, 0, T, 1, [ebno], [L], np.ones((1, 1), np.uint8), e, b, r)
    assert (int(e[0, 0]), int(b[0, 0]), int(r[0, 0])) == (int((nd > 0).sum()), int(nd.sum()), T)


def test_default_rounds_respect_max_err(built_lib, oracle_built):
    """Reference defaults (1000 runs, 100 errors): the library's own rounds (batch = 0) must stop a high-BLER point
    early — run count well below max_runs — and leave low-BLER points at max_runs; the estimate stays consistent
    with the all-trials estimate."""
    o, g = _pair(8, 128, 0)
    ebno = [0.0, 6.0]
    full = g.get_bler_quick(ebno, [1], max_runs=1000, max_err=10**9, seed=3, batch=1000)
    auto = g.get_bler_quick(ebno, [1], max_runs=1000, max_err=100, seed=3)
    assert full[0, 0] > 0.3                                   # > 100 errors long before 1000 runs
    # geometric rounds 256, 256, 512: the 0 dB point stops after the first round(s); same trials => the prefix estimate
    first = g.get_bler_quick(ebno, [1], max_runs=256, max_err=10**9, seed=3, batch=256)
    assert auto[0, 0] == first[0, 0]
    assert auto[0, 1] == full[0, 1]                           # the clean point ran all 1000 trials
    # batch = 1 reproduces the reference's per-run granularity: stops right after the (max_err+1)-th error
    one = g.get_bler_quick([0.0], [1], max_runs=400, max_err=20, seed=3, batch=1)
    runs = round(21 / one[0, 0])
    assert abs(21 / runs - one[0, 0]) < 1e-12 and runs < 400


def test_multi_gpu_entry_point_one_device_through_rccl(built_lib, oracle_built):
    """polar_get_bler_quick_multi with n_dev = 1 and the RCCL path forced: the counters go through
    ncclAllReduce(uint64, sum) and must equal the single-GPU entry point exactly (bler and ber)."""
    o, g = _pair(9, 256, 8)
    ebno, Ls = [1.0, 2.0, 3.0], [1, 8]
    want, want_ber = g.get_bler_quick(ebno, Ls, max_runs=600, max_err=40, seed=9, batch=200, return_ber=True)
    g.debug_set("force_rccl", 1)
    got, got_ber = g.get_bler_quick(ebno, Ls, max_runs=600, max_err=40, seed=9, batch=200, return_ber=True, devices=[0])
    used = g.last_used_rccl
    g.debug_set("force_rccl", 0)
    assert used, "RCCL could not be loaded/initialised on the GPU box"
    assert (got == want).all() and (got_ber == want_ber).all()
    # and the host-sum fallback
    g.debug_set("no_rccl", 1)
    got2 = g.get_bler_quick(ebno, Ls, max_runs=600, max_err=40, seed=9, batch=200, devices=[0])
    assert not g.last_used_rccl
    g.debug_set("no_rccl", 0)
    assert (got2 == want).all()
    # the environment form of the knobs is read once, at handle creation: a variable that appears later changes nothing
    os.environ["POLAR_NO_RCCL"] = "1"
    try:
        g.debug_set("force_rccl", 1)
        g.get_bler_quick(ebno, Ls, max_runs=200, max_err=40, seed=9, batch=200, devices=[0])
        assert g.last_used_rccl
        o2, g2 = _pair(9, 256, 8)                   # (a handle created with the variable set takes it)
        g2.debug_set("force_rccl", 1)
        g2.get_bler_quick(ebno, Ls, max_runs=200, max_err=40, seed=9, batch=200, devices=[0])
        assert not g2.last_used_rccl
    finally:
        del os.environ["POLAR_NO_RCCL"]


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


def test_communicators_and_streams_are_cached_on_the_handle(built_lib, monkeypatch):
    """A second multi-device call with the same device list makes no ncclCommInitAll (an 8-rank init costs about as long
    as a short sweep runs); another list, or a handle of its own, makes one."""
    import polar_amd
    L = polar_amd.lib()
    L.polar_debug_comm_inits.restype = C.c_int
    o, g = _pair(8, 128, 8)
    g.debug_set("force_rccl", 1)
    n0 = L.polar_debug_comm_inits()
    a = g.get_bler_quick([1.0, 2.0], [1, 4], max_runs=400, max_err=30, seed=3, batch=100, devices=[0])
    assert g.last_used_rccl, "RCCL could not be loaded/initialised on the GPU box"
    n1 = L.polar_debug_comm_inits()
    b = g.get_bler_quick([1.0, 2.0], [1, 4], max_runs=400, max_err=30, seed=3, batch=100, devices=[0])
    n2 = L.polar_debug_comm_inits()
    assert n1 == n0 + 1 and n2 == n1
    assert np.array_equal(np.asarray(a), np.asarray(b)) and g.last_used_rccl


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
    assert (g.decode_scl_llr(llr, 4) == o.decode_scl_llr(llr, 4)).all()
    time.sleep(4.0)              # (let the sleeper finish its round on the leaked context before the process goes on)
    g.close()


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


def test_python_sharded_driver_and_strided_engine_on_the_gpu(built_lib, oracle_built):
    """polar_amd/montecarlo.py (the multi-process form bench.py / torchrun use) with the real GPU engine `mc_batch`:
    world size 1 equals the native driver round for round; and the engine's strided trial partition — rank r of a
    world of 3 simulates trials base + r, base + r + 3, ... — sums to the unpartitioned counters."""
    from polar_amd.montecarlo import get_bler_quick_sharded
    o, g = _pair(8, 128, 8)
    ebno, Ls = [0.5, 2.0], [1, 4]
    want = g.get_bler_quick(ebno, Ls, max_runs=1200, max_err=50, seed=11, batch=300)
    bler, err, run = get_bler_quick_sharded(g.mc_batch, ebno, Ls, max_runs=1200, max_err=50, seed=11, global_batch=300)
    assert np.array_equal(np.asarray(bler), np.asarray(want))
    P = (len(Ls), len(ebno))
    en = np.ones(P, np.uint8)
    e_all, r_all = np.zeros(P, np.uint64), np.zeros(P, np.uint64)
    g.mc_batch(11, 40, 501, 1, ebno, Ls, en, e_all, r_all)
    e_sum, r_sum = np.zeros(P, np.uint64), np.zeros(P, np.uint64)
    for r in range(3):
        g.mc_batch(11, 40 + r, len(range(r, 501, 3)), 3, ebno, Ls, en, e_sum, r_sum)
    assert np.array_equal(e_sum, e_all) and np.array_equal(r_sum, r_all)


def test_pipelined_rounds_equal_the_round_after_round_loop(built_lib, oracle_built):
    """The native driver pipelines the rounds on the device (a step decodes point 1 of the newest round together with the later
    points of the rounds before: polar_host.cpp mc_step_launch); the step-wise engine (polar_mc_batch, driven round after
    round by montecarlo.get_bler_quick_sharded) does not. Same counters, to the last one: with the early stop biting at different
    rounds for different points and list sizes, with points that are disabled from the start (max_err = 0 after the first
    round), with fixed and geometric rounds — and equal to the CPU restatement's step-wise engine."""
    from polar_amd.montecarlo import get_bler_quick_sharded
    o, g = _pair(8, 128, 8)
    ebno, Ls = [0.0, 1.0, 2.0, 3.0, 4.0], [1, 2, 8]
    for max_err, batch, max_runs in ((20, 200, 3000), (5, 64, 2000), (0, 100, 700), (40, 0, 5000), (10**6, 333, 2000)):
        b1, c1 = g.get_bler_quick(ebno, Ls, max_runs=max_runs, max_err=max_err, seed=21, batch=batch, return_counters=True)
        b2, e2, r2 = get_bler_quick_sharded(g.mc_batch, ebno, Ls, max_runs=max_runs, max_err=max_err, seed=21, global_batch=batch)
        assert np.array_equal(c1["err"], e2) and np.array_equal(c1["run"], r2), (max_err, batch)
        b3, e3, r3 = get_bler_quick_sharded(o.mc_batch, ebno, Ls, max_runs=max_runs, max_err=max_err, seed=21, global_batch=batch)
        assert np.array_equal(c1["err"], e3) and np.array_equal(c1["run"], r3), (max_err, batch)


def test_rank_driver_three_processes_stand_ins_equal_one_device(built_lib):
    """polar_get_bler_quick_rank (what polar_amd/montecarlo.py's get_bler_quick_ranks and `bench.py --gpus N` call): three
    "ranks" — three handles on the one GPU, one thread each, a reduce callback that sums over the three through a barrier —
    return, each of them, the counters of one device alone; early stop included."""
    import threading
    o, g = _pair(8, 128, 8)
    ebno, Ls = [0.5, 1.5, 2.5], [1, 4]
    kw = dict(max_runs=2400, max_err=30, seed=9, batch=300)
    want, cw = g.get_bler_quick(ebno, Ls, return_counters=True, **kw)
    world = 3
    codes = [g] + [_pair(8, 128, 8)[1] for _ in range(world - 1)]
    bar = threading.Barrier(world)
    acc = {}
    lock = threading.Lock()
    res = [None] * world

    def reduce_for(rank):
        def reduce(a):
            k = bar.wait()                               # (everybody has arrived: the accumulator of the previous step is free)
            with lock:
                acc.setdefault("sum", np.zeros_like(a))
                acc["sum"] += a
            bar.wait()
            a[:] = acc["sum"]
            if bar.wait() == 0:
                acc.pop("sum")
            bar.wait()
        return reduce

    def run(rank):
        res[rank] = codes[rank].get_bler_quick_rank(ebno, Ls, rank, world, reduce_for(rank), **kw)
    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    [t.start() for t in th]
    [t.join(timeout=120) for t in th]
    for r in range(world):
        assert res[r] is not None
        assert np.array_equal(res[r][2]["err"], cw["err"]) and np.array_equal(res[r][2]["run"], cw["run"]) and np.array_equal(res[r][0], want)


def test_handle_keeps_callers_device_and_rejects_bad_devices(built_lib):
    import torch
    import polar_amd
    g = polar_amd.PolarCode(7, 64, 0.32, 0)
    assert torch.cuda.current_device() == 0
    with pytest.raises(polar_amd.PolarError):
        g.get_bler_quick([1.0], [1], max_runs=10, devices=[0, 0])
    with pytest.raises(polar_amd.PolarError):
        g.get_bler_quick([1.0], [1], max_runs=10, devices=[torch.cuda.device_count()])
    with pytest.raises(polar_amd.PolarError):
        g.set_tuning(8, 2)          # lds_log = 2 exists only for the 4-wave-block kernels
