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
    points of the rounds before: polar_montecarlo.cpp mc_step_launch); the step-wise engine (polar_mc_batch, driven round after
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
