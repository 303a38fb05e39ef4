"""GPU: `bench.py --gpus N` with N real ranks on a box that has ONE GPU. The driver's scaling run (N = 2, 4, 8) cannot be
rehearsed here with one rank per device, so the test mode `--shared-gpu-gloo` puts the N ranks of the launch on cuda:0 and
reduces over gloo: the same launcher (torch.distributed.run on 127.0.0.1), the same barrier-bracketed regions, the REAL kernels on
the REAL strided shards — rank r decodes the trials r, r + N, ... of every Monte-Carlo round —, the native single-process driver
from rank 0 over N contexts while the other ranks wait at a host barrier, and the check the N-GPU line carries:
a 65536-trial prefix decoded by one GPU alone == the prefix sharded over the ranks == the prefix over the native driver's
contexts (PolarCode.cpp:696-775 sharded per BASELINE.json north_star). Rates of such a run mean nothing; counters do."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("world", [2, 3])
def test_n_ranks_on_one_gpu_give_the_single_gpu_counters(built_lib, world):
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    total = 2 * 262144 * world
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--shared-gpu-gloo", "--steps", "1",
                        "--warmup", "1", "--batch", "4096", "--mc-trials", str(total)],
                       capture_output=True, text=True, timeout=900, env=e, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == world and line["shared_gpu_test"] is True
    assert line["bler"]["runs"] == world * 4096                     # every rank's headline shard reached the reduction
    mc = line["monte_carlo"]
    assert mc["n_gpus"] == world and mc["total_trials"] == total
    assert mc["multiprocess"]["rounds"] == 2 and mc["multiprocess"]["runs"] == [total] * 5
    assert mc["counters_equal_single_gpu_detail"] == {"multiprocess": True, "native_multi": True, "prefix_trials": 65536}
    assert mc["counters_equal_single_gpu"] is True
    nm = mc["native_multi"]
    assert "error" not in nm, nm
    assert nm["rounds"] == 2 and nm["equals_multiprocess_counters"] is True
    assert nm["block_errors"] == mc["multiprocess"]["block_errors"] and nm["block_errors"][0] > nm["block_errors"][-1] > 0


def test_the_result_line_is_the_last_line_on_stdout_with_rccl_initialised(built_lib):
    """With the RCCL process group up (every N > 1 run; forced here with one rank) RCCL writes a start-up banner through C stdio,
    which is flushed when the process exits — it used to land AFTER the JSON line, the one thing the driver reads. bench.py flushes
    the C buffers, prints the line and points stdout at stderr; ranks other than 0 never own stdout at all."""
    e = dict(os.environ, BENCH_FORCE_DIST="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "1", "--batch", "8192",
                        "--cpu-sample", "0", "--no-other-configs", "--mc-trials", "0"],
                       capture_output=True, text=True, timeout=600, env=e, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    last = r.stdout.strip().splitlines()[-1]
    line = json.loads(last)                                          # (fails when anything follows the line)
    assert line["n_gpus"] == 1 and line["value"] > 0 and line["bler"]["runs"] == 8192
    assert sum(1 for l in r.stdout.splitlines() if l.startswith("{")) == 1


def test_a_native_leg_that_does_not_return_cannot_take_the_result_line_with_it(built_lib):
    """The single-process multi-GPU leg of the Monte-Carlo record runs in a thread of its own with a deadline (the library's watchdog
    covers the rounds, not a communicator set-up that never returns on some node): with the deadline set to (almost) nothing the
    launch still ends with rc 0 and ONE result line — headline, multi-process record and an error entry in place of the native one."""
    world = 2
    e = dict(os.environ, BENCH_NATIVE_DEADLINE_S="0.001")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--shared-gpu-gloo", "--steps", "1",
                        "--warmup", "1", "--batch", "4096", "--mc-trials", str(262144 * world)],
                       capture_output=True, text=True, timeout=900, env=e, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    mc = line["monte_carlo"]
    assert line["n_gpus"] == world and line["value"] > 0
    assert "deadline" in mc["native_multi"]["error"] and mc["native_multi_hung"] is True
    assert mc["counters_equal_single_gpu_detail"]["multiprocess"] is True and mc["multiprocess"]["rounds"] == 1
