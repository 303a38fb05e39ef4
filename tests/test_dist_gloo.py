"""CPU suite: the multi-GPU Monte-Carlo sharding logic (polar_amd/montecarlo.py) under
torch.distributed with the gloo backend, world_size 2 and 4. The per-rank engine is the CPU oracle
(test-only) standing in for the GPU engine: the sharded counters must equal the unsharded ones
exactly, because the synthetic trials are counter-based."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import os, sys, json
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import ctypes as C
import numpy as np
import torch, torch.distributed as dist
from oracle_lib import Oracle
from polar_amd.montecarlo import get_bler_quick_sharded, mc_construction_sharded
import oracle_lib
world = int(os.environ.get("WORLD_SIZE", "1"))
if world > 1:
    dist.init_process_group(backend="gloo")
C.CDLL(None).srand(1)
o = Oracle(7, 64, 0.32, 4)
bler, err, run = get_bler_quick_sharded(o.mc_batch, [1.0, 2.5, 4.0], [1, 4], max_runs=96, max_err=10, seed=11, global_batch=24)
cnt = mc_construction_sharded(lambda n, snr, runs, cid, seed, trial0: oracle_lib.mc_construction(n, cid, snr, seed, trial0, runs),
                              6, 1.0, 75, 4, seed=5)
if (not dist.is_initialized()) or dist.get_rank() == 0:
    print("RESULT " + json.dumps({"err": err.tolist(), "run": run.tolist(), "bler": bler.tolist(), "cnt": cnt.tolist()}))
if dist.is_initialized():
    dist.destroy_process_group()
"""


def _run(world, tmp_path):
    w = tmp_path / "worker.py"
    w.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    if world == 1:
        cmd = [sys.executable, str(w), ROOT]
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
               "--master-addr", "127.0.0.1", "--master-port", "29517", str(w), ROOT]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("RESULT ")][-1]
    import json
    return json.loads(line[7:])


def test_sharded_counters_equal_unsharded(oracle_built, tmp_path):
    a = _run(1, tmp_path)
    b = _run(2, tmp_path)
    assert a == b
    c = _run(4, tmp_path)          # 24 trials per round over 4 ranks, 75 construction runs over 4 ranges
    assert a == c
    run = np.array(a["run"])
    err = np.array(a["err"])
    assert run.max() <= 96 and (err <= run).all() and run.min() >= 24
    # early stop happened somewhere (low Eb/N0, max_err = 10) and not everywhere
    assert (run < 96).any() and (run == 96).any()
    # Monte-Carlo construction table: identical for 1 and 2 ranks (checked by a == b), and non-trivial
    cnt = np.array(a["cnt"])
    assert cnt.shape == (64,) and cnt[0] > 20 and cnt[-1] == 0


def test_automatic_rounds_follow_the_native_driver():
    """polar_amd/montecarlo.py takes the rounds of polar_montecarlo.cpp next_round(): `batch` trials over all ranks, or geometric —
    max(256, 2 max_err) first (rounded up to a multiple of the world size), then as many as all rounds before, at most 262144 PER
    RANK (round 3 capped the round over all ranks: eight GPUs got 32768 trials each)."""
    from polar_amd.montecarlo import next_round

    def rounds(batch, max_err, max_runs, world):
        done, out = 0, []
        while done < max_runs:
            t = next_round(batch, max_err, done, max_runs, world)
            out.append(t); done += t
        return out

    assert rounds(0, 100, 1000, 1) == [256, 256, 488]                      # the reference's defaults: 1000 runs, 100 errors
    assert rounds(0, 100, 3 * 262144, 1)[-2:] == [262144, 262144]
    r8 = rounds(0, 100, 40 * 262144, 8)
    assert r8[0] == 256 and max(r8) == 8 * 262144 and r8.count(8 * 262144) >= 3   # every rank reaches 262144 trials per round
    assert rounds(0, 100, 2000, 3)[0] == 258                               # first round: a multiple of the world size
    assert rounds(500, 100, 1200, 4) == [500, 500, 200]                    # a fixed batch is the round over ALL ranks
    assert sum(rounds(0, 7, 12345, 5)) == 12345
