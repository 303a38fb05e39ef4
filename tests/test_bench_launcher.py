"""bench.py --gpus N: a plain launch must really run N ranks — or refuse. No GPU needed:
the gloo dry run goes through the same launcher, rendezvous, barrier and counter all-reduce as the timed run."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(*a, env=None):
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, BENCH, *a], capture_output=True, text=True, timeout=300, env=e, cwd=ROOT)


def test_plain_launch_with_gpus_2_runs_two_ranks():
    r = _run("--gpus", "2", "--steps", "2", "--dry-run-gloo")
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["dry_run"] is True
    assert line["runs_all_ranks"] == 2 * 2 * 262144          # both ranks' shards reached the reduction
    # the end-to-end Monte-Carlo leg (strong scaling: the total is fixed, rounds of 262144 trials per rank, one counter
    # all-reduce per round), with a stand-in engine: both ranks' strided shards reach every round's reduction and the
    # sharded counters of the prefix equal the unsharded ones
    mc = line["monte_carlo"]
    assert mc["n_gpus"] == 2 and mc["total_trials"] == 4194304
    assert mc["multiprocess"]["rounds"] == 4194304 // (2 * 262144)
    assert mc["multiprocess"]["runs"] == [4194304] * 5 and mc["mc_trials_per_s"] > 0
    assert mc["counters_equal_single_gpu"] is True and mc["counters_equal_single_gpu_detail"]["multiprocess"] is True
    # round 6: the line says who reduced the counters (ranks of the process group, its backend, the native leg's RCCL flag)
    seen = line["rccl_ranks_seen"]
    assert seen["world_size"] == 2 and seen["backend"] == "gloo" and "native_multi_used_rccl" in seen


def test_world_size_and_gpus_must_agree():
    r = _run("--gpus", "2", "--dry-run-gloo", env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "must agree" in (r.stderr + r.stdout)


def test_more_gpus_than_visible_is_refused():
    import torch
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    r = _run("--gpus", str(have + 1 if have else 2))
    assert r.returncode != 0
    assert "refusing" in (r.stderr + r.stdout)
