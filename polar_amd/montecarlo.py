"""Monte-Carlo BLER driver: PolarCode::get_bler_quick (PolarCode.cpp:658-785) sharded over GPUs.

One process per GPU; rank r simulates trials {round_base + r + i*world}. Because the synthetic
workload is counter-based (include/polar_synth.h), the union of trials — and therefore every
counter — is independent of the world size. The only communication is one all-reduce (sum, int64)
of the 2*n_L*n_e error/run counters per round (RCCL over xGMI when the backend is "nccl"); the
early stop `num_err > max_err` (PolarCode.cpp:725) is evaluated on the reduced counters between
rounds of `global_batch` trials (default: the native driver's geometric rounds, up to 262144 trials
PER RANK and round — polar_montecarlo.cpp next_round(); the two drivers then take the same rounds and
return the same counters).
"""
import numpy as np

try:
    import torch
    import torch.distributed as dist
except Exception:  # pragma: no cover
    torch = None
    dist = None


def _world():
    if dist is not None and dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def next_round(batch, max_err, done, max_runs, world):
    """Trials of the next round over all ranks: polar_montecarlo.cpp next_round() (fixed `batch`, or geometric: first
    max(256, 2 max_err) rounded up to a multiple of the world size, then as many as all rounds before, at most 262144
    per rank)."""
    if batch:
        t = batch
    elif done == 0:
        t = max(256, 2 * max_err)
        t = -(-t // world) * world
    else:
        t = min(done, 262144 * world)
    return min(t, max_runs - done)


def get_bler_quick_sharded(engine, ebno_vec, list_size_vec, max_runs=1000, max_err=100, seed=1,
                           global_batch=None, device=None, stats=None):
    """engine(seed, t0, T, stride, ebno, Ls, enabled, err, run): adds this rank's counts into the
    uint64 arrays err/run — polar_amd.PolarCode.mc_batch on a GPU. Returns (bler, err, run).
    global_batch=None / 0: geometric rounds (next_round). stats (optional dict) receives "rounds"."""
    rank, world = _world()
    ebno = np.ascontiguousarray(ebno_vec, np.float64)
    Ls = np.ascontiguousarray(list_size_vec, np.uint8)
    P = (len(Ls), len(ebno))
    err = np.zeros(P, np.uint64)
    run = np.zeros(P, np.uint64)
    base = 0
    rounds = 0
    while base < max_runs:
        gb = next_round(global_batch or 0, max_err, base, max_runs, world)
        enabled = (err <= np.uint64(max_err)).astype(np.uint8)            # PolarCode.cpp:725
        if not enabled.any():
            break
        # trials base .. base+gb-1, round-robin over ranks
        mine = len(range(rank, gb, world))
        d_err = np.zeros(P, np.uint64)
        d_run = np.zeros(P, np.uint64)
        if mine:
            engine(seed, base + rank, mine, world, ebno, Ls, enabled, d_err, d_run)
        if world > 1:
            t = torch.from_numpy(np.stack([d_err, d_run]).astype(np.int64))
            if device is not None:
                t = t.to(device)
            dist.all_reduce(t)                                            # sum over ranks
            t = t.cpu().numpy().astype(np.uint64)
            d_err, d_run = t[0], t[1]
        err += d_err
        run += d_run
        base += gb
        rounds += 1
    if stats is not None:
        stats["rounds"] = rounds
    bler = np.where(run > 0, err.astype(np.float64) / np.maximum(run, 1).astype(np.float64), 0.0)
    return bler, err, run


def get_bler_quick_ranks(code, ebno_vec, list_size_vec, max_runs=1000, max_err=100, seed=1, global_batch=None, device=None,
                         stats=None, constellation=None):
    """The same sweep through the library's own driver (polar_get_bler_quick_rank): one process per GPU, this rank's device
    simulates the trials rank, rank + world, ... of every round, the rounds are PIPELINED on the device (a step decodes point 1
    of the newest round together with the later points of the rounds before it — polar_montecarlo.cpp mc_step_launch) and the
    counters are summed by one all-reduce per step. Same counters as get_bler_quick_sharded and as one GPU alone.
    Returns (bler, err, run)."""
    rank, world = _world()

    def reduce(a):
        if world > 1:
            t = torch.from_numpy(a.astype(np.int64))
            if device is not None:
                t = t.to(device)
            dist.all_reduce(t)
            a[:] = t.cpu().numpy().astype(np.uint64)
    bler, _, c = code.get_bler_quick_rank(ebno_vec, list_size_vec, rank, world, reduce, max_runs=max_runs, max_err=max_err, seed=seed,
                                          batch=global_batch or 0, constellation=constellation)
    if stats is not None:
        stats["rounds"] = c["rounds"]
        stats["steps"] = c["steps"]
        for k in ("min", "median", "max", "first"):
            stats["step_ms_" + k] = code.debug_get("round_us_" + k) / 1e3
    return bler, c["err"], c["run"]


def mc_construction_sharded(counter, num_layers, design_snr_db, num_runs, constellation, seed=1, device=None):
    """Monte-Carlo code construction (PolarCode.m:143-196) sharded over GPUs: runs 0..num_runs-1 are
    split into `world` contiguous ranges, rank r counts its range with
    counter(num_layers, design_snr_db, runs, constellation, seed=, trial0=) -> uint64[N]
    (polar_amd.mc_construction on a GPU) and ONE all-reduce (sum, int64[N]) merges the tables.
    The runs are counter-based, so the table does not depend on the world size."""
    rank, world = _world()
    lo = (num_runs * rank) // world
    hi = (num_runs * (rank + 1)) // world
    N = 1 << num_layers
    cnt = np.zeros(N, np.uint64)
    if hi > lo:
        cnt = np.asarray(counter(num_layers, design_snr_db, hi - lo, constellation, seed=seed, trial0=lo), np.uint64)
    if world > 1:
        t = torch.from_numpy(cnt.astype(np.int64))
        if device is not None:
            t = t.to(device)
        dist.all_reduce(t)
        cnt = t.cpu().numpy().astype(np.uint64)
    return cnt
