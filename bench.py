#!/usr/bin/env python3
"""bench.py — throughput of the polar LLR-SCL hot path on MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 launched by
torch.distributed.run with one rank per GPU (RCCL). Prints ONE JSON line on rank 0.

Workload (BASELINE.json metric config): N=2048, K=1024, 16-bit random-parity CRC, L=32, LLR-SCL,
BPSK/AWGN at Eb/N0 = 2 dB, synthetic trials from include/polar_synth.h generated ON the device
before the timed region (inputs resident in HBM). A "step" = one pass of the hot path over one
batch: decode `batch` codewords per GPU + count block errors. Multi-GPU = Monte-Carlo trial
sharding (rank r owns its own trial range, no data-path collective); the error/run counters are
all-reduced (RCCL, uint64 sum) once inside the timed region. Scaling is weak (per-GPU batch fixed).

Second leg, same line ("monte_carlo"): the END-TO-END path north_star shards — PolarCode::get_bler_quick
(PolarCode.cpp:658-785) over BASELINE configuration 4's grid (Eb/N0 1:0.25:2 dB, L=32 + CRC16), a FIXED total number of
trials (strong scaling), rounds of 262144 trials per GPU, one counter all-reduce per round — through both drivers: the
multi-process one (polar_amd/montecarlo.py, one rank per GPU, torch.distributed all-reduce) and the native one
(polar_get_bler_quick_multi_ex from rank 0, one worker thread per GPU, ncclAllReduce) — with `counters_equal_single_gpu`:
a 65536-trial prefix decoded on ONE GPU must give exactly the sharded counters (counter-based inputs).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np
import torch

HBM_PEAK_GBS = 8000.0  # MI355X spec (MI355X_MICROARCH.md: 8.0 TB/s; ~6.3 TB/s achievable)
LDS_PEAK_GBS = 256 * 128 * 2.4    # 256 CUs x 128 B/clk x 2.4 GHz ~ 78.6 TB/s


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=262144, help="codewords per GPU per step (32 rounds of the 8192 codewords in flight: the tail of a launch — half a wave-decode on average — is amortised; 65536: -6 %%)")
    ap.add_argument("--n", type=int, default=11)
    ap.add_argument("--K", type=int, default=1024)
    ap.add_argument("--crc", type=int, default=16)
    ap.add_argument("--L", type=int, default=32)
    ap.add_argument("--ebno", type=float, default=2.0)
    ap.add_argument("--seed", type=int, default=2024)
    ap.add_argument("--waves-per-cu", type=int, default=0)
    ap.add_argument("--lds-log", type=int, default=0)
    ap.add_argument("--cpu-sample", type=int, default=-1, help="codewords for the CPU baseline (-1 = auto, 0 = skip)")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the short timings of BASELINE.json configs 1, 2, 3, 5")
    ap.add_argument("--only-config", default="", help="run ONE of the other configurations (" + ", ".join(OTHER_CONFIGS) + ") and print its record: the command tools/profile_configs.sh profiles")
    ap.add_argument("--mc-trials", type=int, default=16777216, help="total trials of the end-to-end get_bler_quick leg, strong scaling (0 = skip): 64 rounds of 262144 at one GPU, 8 full rounds per GPU at eight")
    ap.add_argument("--dry-run-gloo", action="store_true",
                    help="launcher / rendezvous / counter all-reduce only, on CPU over gloo: no kernel runs, the line carries "
                         "\"dry_run\": true and no throughput (tests/test_bench_launcher.py)")
    ap.add_argument("--shared-gpu-gloo", action="store_true",
                    help="TEST MODE for boxes with one GPU: the N ranks of the launch all use cuda:0 and reduce over gloo — the real "
                         "kernels, the real strided shards, the same launcher, barriers and legs as an N-GPU run; the line carries "
                         "\"shared_gpu_test\": true and its rates mean nothing (tests/test_gpu_bench_ranks.py)")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: become the launcher (one rank per GPU, RCCL over xGMI). Checked BEFORE
        # spawning, so that a box with fewer GPUs fails here with one clear message instead of N tracebacks.
        if not args.dry_run_gloo and not args.shared_gpu_gloo:
            have = torch.cuda.device_count() if torch.cuda.is_available() else 0
            if have < args.gpus:
                raise SystemExit(f"bench.py: --gpus {args.gpus} requested but {have} GPU(s) visible; refusing to run "
                                 f"a {args.gpus}-GPU benchmark on fewer devices")
        import socket
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.stdout.flush()
        os.execv(sys.executable, cmd)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if rank != 0:
        # stdout belongs to rank 0's ONE JSON line: whatever a library of another rank writes there (RCCL prints a version banner
        # through C stdio, flushed when the process exits — after rank 0's line) goes to stderr instead
        sys.stdout.flush()
        os.dup2(2, 1)
    if world != args.gpus:
        raise SystemExit(f"bench.py: launched with WORLD_SIZE={world} but --gpus {args.gpus}: the two must agree "
                         f"(n_gpus in the result line is the number of ranks that really ran)")
    if args.dry_run_gloo:
        return dry_run(args, world, rank)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    shared = args.shared_gpu_gloo
    if shared:
        local_rank = 0                       # every rank on the one GPU of the box
    if torch.cuda.device_count() <= local_rank:
        raise SystemExit(f"bench.py: rank {rank} needs GPU {local_rank} but only {torch.cuda.device_count()} visible")
    torch.cuda.set_device(local_rank)
    if args.only_config:
        from polar_amd import build
        if not os.environ.get("POLAR_AMD_LIB"):
            build.build()
        print(json.dumps(run_config(args.only_config, args, torch.device("cuda", local_rank), steps=args.steps, with_cpu=False)), flush=True)
        return
    dist = None
    if world > 1 or os.environ.get("BENCH_FORCE_DIST"):
        import torch.distributed as dist_mod
        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if shared:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    def allreduce(t, op=None):
        """Sum / max over the ranks: RCCL on the device tensor; in the shared-GPU test mode through host memory over gloo."""
        if not shared:
            dist.all_reduce(t) if op is None else dist.all_reduce(t, op=op)
            return t
        c = t.cpu()
        dist.all_reduce(c) if op is None else dist.all_reduce(c, op=op)
        t.copy_(c)
        return t

    from polar_amd import build
    if rank == 0 and not os.environ.get("POLAR_AMD_LIB"):
        build.build()            # in-tree .so normally travels prebuilt; only one rank may compile
                                 # (an explicit POLAR_AMD_LIB — A/B runs — is used as it is)
        if shared:
            build.build_test()
    if dist:
        dist.barrier()
    import polar_amd
    if shared:
        # TEST MODE only: one GPU stands in for several ("share_device", a fault-injection hook that exists only in the test
        # build of the library: include/polar_amd_debug.h)
        polar_amd.use_library(build.LIB_TEST)

    # the code: Bhattacharyya construction as the reference's main.cpp (eps = 0.32); the CRC matrix
    # comes from glibc rand() after srand(1) — identical on every rank
    import ctypes as C
    C.CDLL(None).srand(C.c_uint(1))
    code = polar_amd.PolarCode(args.n, args.K, 0.32, args.crc)
    if args.waves_per_cu or args.lds_log:
        code.set_tuning(args.waves_per_cu, args.lds_log)
    N, K, B, L = code.N, code.K, args.batch, args.L
    dev = torch.device("cuda", local_rank)
    s = code.snr_sqrt_linear(args.ebno)

    llr = torch.empty((B, N), dtype=torch.float64, device=dev)
    sent = torch.empty((B, K), dtype=torch.uint8, device=dev)
    out = torch.empty((B, K), dtype=torch.uint8, device=dev)
    counters = torch.zeros(2, dtype=torch.int64, device=dev)   # [block errors, runs]
    trial0 = rank * B                                            # Monte-Carlo shard of this rank
    code.synth_llr_dev(args.seed, trial0, B, s, llr.data_ptr(), sent.data_ptr())
    torch.cuda.synchronize()

    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]

    for a, b_ in ev:            # create the underlying hipEvent_t handles
        a.record(); b_.record()

    def step(i=None):
        # HIP events are recorded by the library on the launch stream right around the dominant
        # kernel (scl_decode_llr_kernel), after the small prefix kernel
        if i is not None:
            code.decode_scl_llr_dev(llr.data_ptr(), B, L, out.data_ptr(), ev_start=ev[i][0].cuda_event, ev_stop=ev[i][1].cuda_event)
        else:
            code.decode_scl_llr_dev(llr.data_ptr(), B, L, out.data_ptr())
        code.count_errors_dev(out.data_ptr(), sent.data_ptr(), B, counters.data_ptr())
        counters[1] += B

    for _ in range(args.warmup):
        step()
    counters.zero_()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    if dist:
        allreduce(counters)              # RCCL sum of the Monte-Carlo counters (xGMI)
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if dist:
        allreduce(tmax, dist.ReduceOp.MAX)
    dt = float(tmax.item())

    kern_ms = [a.elapsed_time(b) for a, b in ev]
    kern_avg_s = (sum(kern_ms) / len(kern_ms)) / 1e3
    total_cw = B * world * args.steps
    value = total_cw / dt
    alg_bytes_per_cw = N * 8 + K                       # doubles consumed, 1 B per info bit out
    achieved = (B * alg_bytes_per_cw) / kern_avg_s / 1e9
    cnt = counters.cpu().numpy()

    traffic, prof = traffic_from_profile(args, B)
    res = {
        "metric": "codewords/s (N=2048 K=1024 L=32 LLR-SCL)",
        "value": value,
        "unit": "codewords/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic (Philox BPSK/AWGN trials generated on device, include/polar_synth.h)",
        "config": {
            "workload": f"LLR-SCL decode N={N} K={K} crc={args.crc} L={L} BPSK/AWGN Eb/N0={args.ebno} dB, "
                        f"batch {B} codewords/GPU/step, bit-exact fp64 path",
            "batch_per_gpu": B,
            "parallelism": f"monte-carlo trial shard x{world}, counters all-reduced (RCCL)" if world > 1 else "single GPU",
        },
        "bler": {"block_errors": int(cnt[0]), "runs": int(cnt[1])},
        "roofline": {
            "bound": "hbm",
            "kernel": "scl_decode_llr_kernel",
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic,
            "traffic_profile_lib_sha256": prof.get("lib_sha256"),
            "kernel_ms_avg": kern_avg_s * 1e3,
            "algorithmic_bytes_per_codeword": alg_bytes_per_cw,
            "algorithmic_bytes_per_launch": B * alg_bytes_per_cw,
            "node_evals_per_s": (B * L * N * code.n) / kern_avg_s,
            # the resources this kernel is really bounded by: measured HBM bytes of the
            # committed PMC profile over the LIVE kernel time, and the profile's fp64 VALU occupancy
            "traffic_rate_GBps": (traffic / kern_avg_s / 1e9) if traffic else None,
            "traffic_frac_of_peak": (traffic / kern_avg_s / 1e9 / HBM_PEAK_GBS) if traffic else None,
        },
    }
    # the two other rooflines of this path, and the LDS-vs-HBM byte split ("LDS-hit fraction").
    # LDS bytes are MODELLED from the schedule (every node of the layers of size <= 8 writes 8 B per path, every node
    # whose source layer is LDS-resident reads 2 x 8 B; register-chained visits read less: an upper bound), HBM
    # bytes and the VALU occupancy come from the committed rocprofv3 PMC profile of this very command.
    lds_layers_w, lds_layers_r = 4, 3                       # layers written to / read from LDS (S <= 8 / S <= 4)
    lds_bytes = B * L * N * 8.0 * (lds_layers_w + 2 * lds_layers_r)
    # measured: SQ_INSTS_LDS of the committed PMC pass (per-wave LDS instructions) x 64 lanes x 8 B — every LDS access of this
    # kernel's node state is a ds_read_b64 / ds_write_b64 per lane (the wider ds_read_b128 of the byte stack and the 4-byte
    # control words are a few per cent of the count and pull in opposite directions); modelled figure beside it
    lds_meas = prof.get("lds_insts_per_launch") * 64 * 8.0 if prof.get("lds_insts_per_launch") else None
    res["roofline"]["lds"] = {
        "measured_bytes_per_launch": lds_meas, "measured_from": "SQ_INSTS_LDS x 64 lanes x 8 B (profiles/traffic.json, same library)" if lds_meas else None,
        "modelled_bytes_per_launch": lds_bytes,
        "achieved": (lds_meas or lds_bytes) / kern_avg_s / 1e9, "peak": LDS_PEAK_GBS, "unit": "GB/s",
        "frac": (lds_meas or lds_bytes) / kern_avg_s / 1e9 / LDS_PEAK_GBS,
        "lds_hit_fraction": (lds_meas / (lds_meas + traffic)) if (traffic and lds_meas) else None,
        "lds_hit_fraction_modelled": (lds_bytes / (lds_bytes + traffic)) if traffic else None,
        "lds_bank_conflict_cycle_frac_in_profile": prof.get("lds_bank_conflict_frac"),
        "note": "lds_hit_fraction = LDS bytes / (LDS + measured HBM bytes) of the decoder's working state, both from the rocprofv3 PMC passes of this library; the modelled figure is the schedule's upper bound",
    }
    res["roofline"]["valu"] = {
        "busy_frac_in_profile": prof.get("valu_busy_frac_in_profile"),
        "valu_insts_per_wave_decode_in_profile": prof.get("valu_insts_per_wave_decode"),
        "note": "SQ_ACTIVE_INST_VALU x 4 / (1024 SIMDs x kernel cycles), committed profile (profiles/traffic.json)",
    }

    if args.mc_trials > 0:
        del llr, sent
        if world == 1:        # (with peers a failure on one rank cannot be contained here: the others wait in a collective)
            try:
                res["monte_carlo"] = monte_carlo_leg(args, code, dist, dev, rank, world, shared_gpu=shared)
            except Exception as ex:
                res["monte_carlo"] = {"error": f"{type(ex).__name__}: {ex}"[:500]}
                torch.cuda.synchronize()
        else:
            res["monte_carlo"] = monte_carlo_leg(args, code, dist, dev, rank, world, shared_gpu=shared)
        if not res["monte_carlo"].get("native_multi_hung"):                 # (else a thread is still inside the handle: hands off)
            llr = torch.empty((B, N), dtype=torch.float64, device=dev)      # (the CPU baseline below re-checks the batch)
            code.synth_llr_dev(args.seed, trial0, B, s, llr.data_ptr(), 0)
            torch.cuda.synchronize()
    if shared:
        res["shared_gpu_test"] = True
    if world > 1:
        # what really reduced the counters: the ranks of the process group and its backend, and whether the native leg's
        # single-process driver went through RCCL (false = host-side sum: e.g. one GPU standing in for several)
        res["rccl_ranks_seen"] = {"world_size": dist.get_world_size(), "backend": dist.get_backend(),
                                  "distinct_devices": len(set(gather_device_ids(dist, local_rank, shared))),
                                  "native_multi_used_rccl": res.get("monte_carlo", {}).get("native_multi", {}).get("used_rccl")}
    hung = bool(res.get("monte_carlo", {}).get("native_multi_hung"))
    res["section_seconds"] = {}

    def guarded(name, fn):
        """The records beside the headline must not be able to take the result line with them: a failure in one of them is
        recorded in its place."""
        t_sec = time.perf_counter()
        try:
            res[name] = fn()
            res["section_seconds"][name] = round(time.perf_counter() - t_sec, 1)
        except Exception as ex:
            import traceback
            res[name] = {"error": f"{type(ex).__name__}: {ex}"[:500], "traceback_tail": traceback.format_exc()[-800:]}
            torch.cuda.synchronize()

    if rank == 0 and world == 1 and args.cpu_sample != 0 and not hung:
        guarded("cpu_baseline", lambda: cpu_baseline(args, code, llr, out))
    if rank == 0 and world == 1 and not args.no_other_configs and not hung:
        guarded("other_configs", lambda: other_configs(args, dev))
        guarded("one_codeword_per_call", lambda: latency_record(args, code))
        guarded("host_batch", lambda: host_batch_record(args, dev))
        guarded("p1_paths", lambda: p1_record(args, code))
    if dist:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        emit_last_line(json.dumps(res))        # the ONE JSON line, last thing on stdout
        if res.get("monte_carlo", {}).get("native_multi_hung"):
            os._exit(0)                        # (a thread is still inside the library: no orderly teardown)


def gather_device_ids(dist, local_rank, shared):
    """The (host-visible) GPU index every rank of the launch really computes on."""
    t = torch.zeros(dist.get_world_size(), dtype=torch.int64)
    t[dist.get_rank()] = local_rank
    if shared or dist.get_backend() == "gloo":
        dist.all_reduce(t)
        return [int(x) for x in t]
    d = t.to(torch.device("cuda", local_rank))
    dist.all_reduce(d)
    return [int(x) for x in d.cpu()]


def emit_last_line(line):
    """Print the result line as the LAST thing this process puts on stdout: C-level buffers first (RCCL's start-up banner sits in
    one until exit), then the line, then stdout is pointed at stderr so that nothing written later (process-group teardown, exit
    handlers) can follow it."""
    import ctypes
    sys.stdout.flush()
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    print(line, flush=True)
    os.dup2(2, 1)


MC_GRID = [1.0, 1.25, 1.5, 1.75, 2.0]        # BASELINE configuration 4: Eb/N0 1:0.25:2 dB, L = 32 + CRC16
MC_PREFIX = 65536


def monte_carlo_leg(args, code, dist, dev, rank, world, engine=None, native=True, shared_gpu=False):
    """End-to-end get_bler_quick, strong scaling: args.mc_trials trials in total whatever the world size, rounds of 262144
    trials per GPU, one all-reduce of the counters per round (PolarCode.cpp:696-775; the early stop of :725 is disabled by
    max_err so that every world size does the same work). Both drivers, and the sharded counters of a 65536-trial
    prefix against ONE GPU decoding that prefix alone."""
    from polar_amd.montecarlo import get_bler_quick_sharded
    Ls = [args.L]
    total, per_round = args.mc_trials, 262144 * world
    no_stop = 10 ** 12
    use_ranks = engine is None              # the real engine: the library's own driver per rank; stand-in engines: the step-wise loop
    if engine is None:
        engine = code.mc_batch

    def sync():
        if dev is not None:
            torch.cuda.synchronize()

    red_dev = None if shared_gpu else dev       # (shared-GPU test mode: the process group is gloo, counters reduced on the host)
    host_group = None
    if dist is not None and world > 1:
        host_group = dist.new_group(backend="gloo")          # host-side barriers: no barrier kernel sits on the peers' GPUs
                                                             # while rank 0 drives all of them through the native entry point

    def hbar():
        if host_group is not None:
            dist.barrier(group=host_group)

    def tmax(dt):
        if dist is None or world == 1:
            return dt
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=host_group)
        return float(t.item())

    out = {"workload": f"get_bler_quick N={code.N} K={code.K} crc={args.crc} L={args.L}, Eb/N0 {MC_GRID[0]}:0.25:{MC_GRID[-1]} dB, "
                       f"{total} trials in total (strong scaling), rounds of 262144 trials per GPU, one counter all-reduce per round, "
                       f"generation + encoding + channel + decode + counting on the device",
           "total_trials": total, "n_gpus": world}
    # ---- multi-process driver (this launch: one rank per GPU): the library's own driver per rank (pipelined rounds), the
    # counters summed by one torch.distributed all-reduce per step; stand-in engines (dry run) keep the step-wise loop
    from polar_amd.montecarlo import get_bler_quick_ranks
    if use_ranks:
        def sharded(max_runs, global_batch, stats=None):
            return get_bler_quick_ranks(code, MC_GRID, Ls, max_runs=max_runs, max_err=no_stop, seed=args.seed, global_batch=global_batch, device=red_dev, stats=stats)
        drv = "polar_get_bler_quick_rank per rank (polar_amd/montecarlo.py get_bler_quick_ranks): rounds pipelined on the device, one torch.distributed all-reduce per step"
    else:
        def sharded(max_runs, global_batch, stats=None):
            return get_bler_quick_sharded(engine, MC_GRID, Ls, max_runs=max_runs, max_err=no_stop, seed=args.seed, global_batch=global_batch, device=red_dev, stats=stats)
        drv = "polar_amd/montecarlo.py get_bler_quick_sharded, one rank per GPU, torch.distributed all-reduce per round"
    # warm-up (allocations): THREE rounds — the merged batch of a step carries the survivors of the rounds before it, and its buffers
    # reach their steady size only once the pipeline is full (one round alone left a 4-GiB reallocation inside the timed sweep: 3 %)
    sharded(min(total, 3 * per_round), per_round)
    sync(); hbar()
    st = {}
    t0 = time.perf_counter()
    bler, err, run = sharded(total, per_round, st)
    sync(); hbar()
    dt = tmax(time.perf_counter() - t0)
    out["multiprocess"] = {"driver": drv, "seconds": dt, "mc_trials_per_s": total / dt, "rounds": st.get("rounds"), "steps": st.get("steps"),
                           "step_ms": {k: st.get("step_ms_" + k) for k in ("first", "min", "median", "max")},
                           "bler": [float(x) for x in bler[0]], "block_errors": [int(x) for x in err[0]], "runs": [int(x) for x in run[0]]}
    out["mc_trials_per_s"] = total / dt
    # weak scaling beside it: a fixed 8 rounds of 262144 trials PER GPU
    weak_total = 8 * per_round
    sync(); hbar()
    t0 = time.perf_counter()
    sharded(weak_total, per_round)
    sync(); hbar()
    dtw = tmax(time.perf_counter() - t0)
    out["mc_weak"] = {"trials": weak_total, "rounds_per_gpu": 8, "seconds": dtw, "mc_trials_per_s": weak_total / dtw,
                      "note": "the same sweep with 8 rounds of 262144 trials per GPU whatever the world size"}
    # ---- the same 65536-trial prefix: sharded over the ranks vs ONE GPU alone (rank 0)
    _, e_sh, r_sh = sharded(MC_PREFIX, MC_PREFIX)
    equal = {"multiprocess": None, "native_multi": None}
    e_one = r_one = None
    if rank == 0 and native:
        _, c1 = code.get_bler_quick(MC_GRID, Ls, max_runs=MC_PREFIX, max_err=no_stop, seed=args.seed, batch=MC_PREFIX, return_counters=True)
        e_one, r_one = c1["err"], c1["run"]
        equal["multiprocess"] = bool(np.array_equal(e_one, e_sh) and np.array_equal(r_one, r_sh))
    # ---- native driver: ONE host process (rank 0) drives all the GPUs of the launch (what the C++ / MATLAB hosts call)
    if rank == 0 and native:
        box = {}

        def native_leg():
            try:
                code.debug_set("multi_timeout_s", 180)
                devs = list(range(world))
                if shared_gpu:                                       # one GPU standing in for all of them (separate contexts, host-side sum)
                    code.debug_set("share_device", 1)
                    devs = [0] * world
                code.get_bler_quick(MC_GRID, Ls, max_runs=min(total, per_round), max_err=no_stop, seed=args.seed, batch=per_round, devices=devs)   # warm-up: contexts, communicators
                t0 = time.perf_counter()
                b2, c2 = code.get_bler_quick(MC_GRID, Ls, max_runs=total, max_err=no_stop, seed=args.seed, batch=per_round, devices=devs, return_counters=True)
                dt2 = time.perf_counter() - t0
                # (the step times of THIS call: the prefix call below overwrites the handle's record)
                step_ms = {k: code.debug_get("round_us_" + k) / 1e3 for k in ("first", "min", "median", "max")}
                steps2, used2 = code.debug_get("round_us_count"), bool(code.last_used_rccl)
                _, c3 = code.get_bler_quick(MC_GRID, Ls, max_runs=MC_PREFIX, max_err=no_stop, seed=args.seed, batch=MC_PREFIX, devices=devs, return_counters=True)
                box["equal"] = bool(np.array_equal(e_one, c3["err"]) and np.array_equal(r_one, c3["run"]))
                box["rec"] = {"driver": "polar_get_bler_quick_multi_ex from rank 0: one worker thread, stream and table clone per GPU, rounds pipelined, "
                                        "ncclAllReduce(uint64, sum) per step" + ("" if used2 else " (host-side sum: RCCL not used)"),
                              "seconds": dt2, "mc_trials_per_s": total / dt2, "rounds": c2["rounds"], "steps": steps2, "used_rccl": used2,
                              "devices": len(devs), "step_ms": step_ms, "step_ms_mean": dt2 * 1e3 / max(steps2, 1),
                              "bler": [float(x) for x in b2[0]], "block_errors": [int(x) for x in c2["err"][0]],
                              "equals_multiprocess_counters": bool(np.array_equal(c2["err"], err) and np.array_equal(c2["run"], run))}
            except Exception as ex:                              # (the headline above must be printed whatever happens here)
                box["rec"] = {"error": str(ex)[:500]}

        # in a thread of its own, with a deadline: a communicator set-up that never returns on some node must not take the
        # headline line with it (the library's own watchdog covers the rounds, not ncclCommInitAll)
        import threading
        th = threading.Thread(target=native_leg, daemon=True)
        th.start()
        th.join(timeout=float(os.environ.get("BENCH_NATIVE_DEADLINE_S", "420")))
        if th.is_alive():
            out["native_multi"] = {"error": "the native multi-device leg did not finish before its deadline; skipped"}
            out["native_multi_hung"] = True
        else:
            out["native_multi"] = box.get("rec", {"error": "no record"})
            equal["native_multi"] = box.get("equal")
    hbar()
    # true only when EVERY driver that was attempted compared equal (a native leg that raised or missed its deadline leaves
    # None there: the combined flag is then None, never a pass)
    out["counters_equal_single_gpu"] = (None if (not native or rank != 0 or equal["multiprocess"] is None or equal["native_multi"] is None)
                                        else bool(equal["multiprocess"] and equal["native_multi"]))
    out["counters_equal_single_gpu_detail"] = dict(equal, prefix_trials=MC_PREFIX)
    return out


def dry_run(args, world, rank):
    """The multi-process skeleton of main() without a GPU: rendezvous, barrier-bracketed timed region, counter
    all-reduce, max-over-ranks time, one JSON line from rank 0. Runs on CPU over gloo; measures nothing."""
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    os.environ.setdefault("RANK", "0")
    os.environ.setdefault("WORLD_SIZE", "1")
    dist.init_process_group(backend="gloo")
    counters = torch.zeros(2, dtype=torch.int64)
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        counters[1] += args.batch            # the trials this rank would have decoded (shard rank*batch ...)
    dist.all_reduce(counters)
    dist.barrier()
    tmax = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dist.barrier()
    # the end-to-end Monte-Carlo leg with a stand-in engine (no kernel): trial t is a "block error" at point i when a hash
    # of (t, i) says so — the rounds, the per-round counter all-reduce, the strided shards and the prefix comparison are real
    mc = None
    if args.mc_trials > 0:
        def engine(seed, t0, T, stride, ebno, Ls, enabled, err, run):
            t = (np.arange(T, dtype=np.uint64) * np.uint64(stride) + np.uint64(t0))
            for li in range(len(Ls)):
                for ie in range(len(ebno)):
                    if enabled[li, ie]:
                        h = (t * np.uint64(2654435761) + np.uint64(97 * ie + seed)) % np.uint64(1000)
                        err[li, ie] += np.uint64(int((h < np.uint64(160 >> ie)).sum()))
                        run[li, ie] += np.uint64(T)

        class _One:        # the "one GPU alone" side of the prefix check: the same engine, unsharded
            N, K = 1 << args.n, args.K

            def get_bler_quick(self, grid, Ls, max_runs, max_err, seed, batch, return_counters=True, devices=None):
                e = np.zeros((len(Ls), len(grid)), np.uint64); r = np.zeros_like(e)
                engine(seed, 0, max_runs, 1, grid, Ls, np.ones(e.shape, np.uint8), e, r)
                self.last_used_rccl = False
                return e / np.maximum(r, 1), {"err": e, "run": r, "rounds": 1}

            def debug_set(self, k, v):
                pass

            def debug_get(self, k):
                return 0

        a2 = argparse.Namespace(**vars(args))
        a2.mc_trials = min(args.mc_trials, 8 * 262144 * world)
        mc = monte_carlo_leg(a2, _One(), dist, None, rank, world, engine=engine, native=True)
        if rank == 0:
            mc["native_multi"] = {"skipped": "dry run: the native driver needs GPUs"}
            mc["counters_equal_single_gpu_detail"]["native_multi"] = None
    dist.destroy_process_group()
    if rank == 0:
        emit_last_line(json.dumps({"metric": "codewords/s (N=2048 K=1024 L=32 LLR-SCL)", "value": None, "unit": "codewords/s",
                                   "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "dry_run": True,
                                   "runs_all_ranks": int(counters[1]), "scaling": "weak", "monte_carlo": mc,
                                   "rccl_ranks_seen": {"world_size": world, "backend": "gloo", "distinct_devices": 0, "native_multi_used_rccl": None},
                                   "config": {"workload": "dry run (gloo, CPU): launcher and counter reduction only"}}))


def traffic_from_profile(args, B):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (profiles/traffic.json, written by tools/update_traffic.py from a tools/profile.sh run of this
    very command) — only when it was taken on the same workload; otherwise null."""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
        c = t["config"]
        # a profile counts only for the library it was taken with: a kernel change without a re-profile reports null
        if (c["n"], c["K"], c["crc"], c["L"], c["batch"]) == (args.n, args.K, args.crc, args.L, B) and t.get("lib_sha256") == lib_sha256():
            return t["traffic_bytes_per_launch"], t
    except Exception:
        pass
    return None, {}


def cpu_baseline(args, code, llr, out):
    """Time the CPU side on this box's host cores on a bounded sample of the SAME workload
    (first `sample` codewords of the batch), single thread. Uses the unmodified reference build
    (oracle/_ref, kind "reference") when it travelled with the snapshot, else our C restatement
    (kind "port"). Also cross-checks the GPU result on the sample (bit-exact)."""
    import ctypes as C
    import oracle_lib
    kind = "reference" if oracle_lib.have_reference() else "port"
    C.CDLL(None).srand(C.c_uint(1))
    if kind == "reference":
        cpu = oracle_lib.Reference(args.n, args.K, 0.32, args.crc, srand=1)
    else:
        cpu = oracle_lib.Oracle(args.n, args.K, 0.32, args.crc, srand=1)
    assert (cpu.crc_matrix() == code.crc_matrix).all() and (cpu.order() == code.channel_order_descending).all()
    # probe the speed on 16 codewords, then size the sample for ~15 s
    h = llr[:16].cpu().numpy()
    t = time.perf_counter()
    cpu.decode_scl_llr(h, args.L)
    per = (time.perf_counter() - t) / 16
    sample = args.cpu_sample if args.cpu_sample > 0 else int(max(16, min(llr.shape[0], 15.0 / per)))
    h = llr[:sample].cpu().numpy()
    t = time.perf_counter()
    ref_out = cpu.decode_scl_llr(h, args.L)
    dt = time.perf_counter() - t
    mism = int((ref_out != out[:sample].cpu().numpy()).any(axis=1).sum())
    # all usable host cores: independent per-thread decoder objects on a second bounded sample
    import threading
    cores = oracle_lib.usable_cpus()
    threads = 2 * cores
    sample2 = int(max(threads, min(llr.shape[0], 10.0 * (sample / dt) * cores * 0.8)))
    h2 = llr[:sample2].cpu().numpy()
    want2 = np.zeros((sample2, args.K), np.uint8)
    crcm = code.crc_matrix

    def work(t):
        c = (oracle_lib.Reference if kind == "reference" else oracle_lib.Oracle)(args.n, args.K, 0.32, args.crc)
        c.set_crc_matrix(crcm)
        sl = slice(t * sample2 // threads, (t + 1) * sample2 // threads)
        if sl.stop > sl.start:
            want2[sl] = c.decode_scl_llr(h2[sl], args.L)

    t = time.perf_counter()
    th = [threading.Thread(target=work, args=(i,)) for i in range(threads)]
    [x.start() for x in th]
    [x.join() for x in th]
    dt2 = time.perf_counter() - t
    mism2 = int((want2 != out[:sample2].cpu().numpy()).any(axis=1).sum())
    full = cpu_full_trial(args, kind)
    return {
        "value": sample / dt,
        "unit": "codewords/s",
        "cores": 1,
        "host_cores_available": os.cpu_count(),
        "kind": kind,
        "full_trial": full,
        "sample": f"first {sample} codewords of the benchmark batch, decode_scl_llr only, single thread",
        "gpu_vs_cpu_mismatching_codewords": mism,
        "all_cores": {
            "value": sample2 / dt2, "unit": "codewords/s", "cores": cores, "threads": threads,
            "nproc": os.cpu_count(), "note": "cores = cgroup CPU quota of the container (the box reports nproc logical CPUs)",
            "sample": f"first {sample2} codewords of the benchmark batch, one decoder object per thread",
            "gpu_vs_cpu_mismatching_codewords": mism2,
        },
    }


def cpu_full_trial(args, kind):
    """SURVEY §8(d) "single-thread full trial": the reference's OWN Monte-Carlo loop (PolarCode::get_bler_quick,
    PolarCode.cpp:696-775: info bits, encode, BPSK, AWGN, LLRs, decode, compare) timed on one host core at ONE point of the
    benchmark's Eb/N0 — its constants are compiled in (max_runs = 1000, max_err = 100, .cpp:661-662), so the number of trials
    it ran follows from the BLER it returns: 1000 when it never stopped early, else the trial at which the 101st error
    fell (errors / bler). With the restatement (kind "port") the same loop with the same constants."""
    import oracle_lib
    if kind == "reference":
        cpu = oracle_lib.Reference(args.n, args.K, 0.32, args.crc, srand=1)
        fn = lambda: cpu.get_bler_quick([args.ebno], [args.L])
    else:
        cpu = oracle_lib.Oracle(args.n, args.K, 0.32, args.crc, srand=1)
        fn = lambda: cpu.get_bler_quick_ref([args.ebno], [args.L], 1000, 100)
    t = time.perf_counter()
    bler = float(fn()[0][0])
    dt = time.perf_counter() - t
    runs = 1000 if bler * 1000 <= 100.5 else int(round(101 / bler))
    return {"value": runs / dt, "unit": "trials/s", "cores": 1, "kind": kind, "trials": runs, "seconds": dt, "bler": bler,
            "sample": f"get_bler_quick([{args.ebno}], [{args.L}]) of the CPU side: the reference's whole loop (generation, encoder, channel, "
                      f"LLRs, decode, compare), its own constants max_runs = 1000, max_err = 100"}


def lib_sha256():
    import hashlib
    import polar_amd
    path = os.environ.get("POLAR_AMD_LIB") or os.path.join(ROOT, "polar_amd", "libpolar_amd.so")
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()


def profile_entry(name):
    """Counter traffic of configuration `name` from the committed rocprofv3 PMC passes (profiles/traffic_configs.json,
    tools/profile_configs.sh) — only when they were taken with THIS library (sha256 of libpolar_amd.so); a profile of
    another build is refused (null), never reported as if it were live."""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "traffic_configs.json")))
        if t.get("lib_sha256") == lib_sha256():
            return t["configs"].get(name)
    except Exception:
        pass
    return None


OTHER_CONFIGS = {
    # name: (n, K, crc, L, batch, axis value [Eb/N0 dB, or SNR dB for 16-ASK], constellation, dominant kernel, CPU sample)
    "config1": (9, 256, 0, 1, 262144, 2.0, "bpsk", "sc8_decode_kernel", 4096),
    "config2": (11, 1024, 0, 1, 65536, 2.0, "bpsk", "sc8_decode_kernel", 4096),        # 4 096-codeword CPU prefix
    "config2_b262144": (11, 1024, 0, 1, 262144, 2.0, "bpsk", "sc8_decode_kernel", 0),
    "config3": (11, 1024, 16, 4, 65536, 2.0, "bpsk", "scl_decode_llr_kernel<4, 3, 0, true, 0, 0>", 2048),
    "config5": (10, 512, 0, 8, 65536, 13.0, "ask16-gray", "scl_decode_llr_kernel<8, 3, 0, true, 0, 0>", 2048),
    # the same at four times the batch (the tail of the persistent launch amortised; no CPU sample)
    "config3_b262144": (11, 1024, 16, 4, 262144, 2.0, "bpsk", "scl_decode_llr_kernel<4, 3, 0, true, 0, 0>", 0),
    "config5_b262144": (10, 512, 0, 8, 262144, 13.0, "ask16-gray", "scl_decode_llr_kernel<8, 3, 0, true, 0, 0>", 0),
}


def make_config(name):
    """The code and the workload description of one BASELINE.json configuration. config 5 is the reference's OWN code:
    the Monte-Carlo construction table it ships for 16-ASK Gray BICM at 13 dB (a committed data fixture,
    tests/golden/polar_golden.npz), decoded on 16-ASK BICM LLRs at the centre of its SNR grid."""
    import ctypes as C
    import polar_amd
    n, K, crc, L, B, axis, const, kern, cpu_n = OTHER_CONFIGS[name]
    C.CDLL(None).srand(C.c_uint(1))
    if const == "bpsk":
        code = polar_amd.PolarCode(n, K, 0.32, crc)
    else:
        g = np.load(os.path.join(ROOT, "tests", "golden", "polar_golden.npz"))
        code = polar_amd.PolarCode.from_counts(g["cfg5_n10_k512_ask16/counts"], K, crc)
    return code


def run_config(name, args, dev, steps=3, with_cpu=True):
    import ctypes as C
    import oracle_lib
    n, K, crc, L, B, axis, const, kern, cpu_n = OTHER_CONFIGS[name]
    code = make_config(name)
    Nn = 1 << n
    llr = torch.empty((B, Nn), dtype=torch.float64, device=dev)
    sent = torch.empty((B, K), dtype=torch.uint8, device=dev)
    out = torch.empty((B, K), dtype=torch.uint8, device=dev)
    cnt = torch.zeros(2, dtype=torch.int64, device=dev)
    if const == "bpsk":
        code.synth_llr_dev(args.seed, 0, B, code.snr_sqrt_linear(axis), llr.data_ptr(), sent.data_ptr())
    else:
        code.synth_bicm_llr_dev(const, args.seed, 0, B, axis, llr.data_ptr(), sent.data_ptr())
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    for a, b_ in ev:
        a.record(); b_.record()
    code.decode_scl_llr_dev(llr.data_ptr(), B, L, out.data_ptr())                 # warm-up (allocations)
    code.count_errors_dev(out.data_ptr(), sent.data_ptr(), B, cnt.data_ptr())
    cnt.zero_()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        code.decode_scl_llr_dev(llr.data_ptr(), B, L, out.data_ptr(), ev_start=ev[i][0].cuda_event, ev_stop=ev[i][1].cuda_event)
        code.count_errors_dev(out.data_ptr(), sent.data_ptr(), B, cnt.data_ptr())
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    kern_s = sum(a.elapsed_time(b_) for a, b_ in ev) / steps / 1e3
    alg = Nn * 8 + K
    achieved = B * alg / kern_s / 1e9
    prof = profile_entry(name)
    traffic = prof["traffic_bytes_per_launch"] if prof else None
    res = {"config": name, "workload": f"N={Nn} K={K} crc={crc} L={L} {const} " + (f"Eb/N0={axis} dB" if const == "bpsk" else f"SNR={axis} dB (BICM)") + f", batch {B}",
           "batch": B, "value": B / dt, "unit": "codewords/s", "ms_per_step": dt * 1e3,
           "block_errors_per_step": int(cnt[0].item()) // steps,
           "roofline": {"bound": "hbm", "kernel": kern, "kernel_ms_avg": kern_s * 1e3, "achieved": achieved, "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "algorithmic_bytes_per_codeword": alg,
                        "traffic": traffic,
                        "traffic_over_algorithmic": (traffic / (B * alg)) if traffic else None,
                        "traffic_rate_GBps": (traffic / kern_s / 1e9) if traffic else None,
                        "kernel_ms_in_profile": prof.get("kernel_avg_ns", 0) / 1e6 if prof else None}}
    if with_cpu and cpu_n:
        # the CPU side on a bounded prefix of the SAME batch, single thread: the unmodified reference build when it travelled
        # with the snapshot (kind "reference"), else the C restatement (kind "port"); and the GPU result checked against it
        kind = "reference" if oracle_lib.have_reference() else "port"
        C.CDLL(None).srand(C.c_uint(1))
        cpu = (oracle_lib.Reference if kind == "reference" else oracle_lib.Oracle)(n, K, 0.32, crc, srand=1)
        if const != "bpsk":
            cpu.set_tables(code.frozen_bits, code.channel_order_descending)
        h = llr[:cpu_n].cpu().numpy()
        t = time.perf_counter()
        ref_out = cpu.decode_scl_llr(h, L)
        tc = time.perf_counter() - t
        res["cpu_baseline"] = {"value": cpu_n / tc, "unit": "codewords/s", "cores": 1, "kind": kind,
                               "sample": f"first {cpu_n} codewords of the batch, decode_scl_llr only, single thread",
                               "gpu_vs_cpu_mismatching_codewords": int((ref_out != out[:cpu_n].cpu().numpy()).any(axis=1).sum())}
    del llr, sent, out
    return res


HOST_BATCH_CONFIGS = ("config2", "config3", "config5", "headline")
HOST_KNOBS = ("host_pipe_min_bytes", "host_chunk_bytes", "host_lanes", "host_threads", "host_ramp", "host_prefault")


def pcie_rates(dev, mib=256):
    """Pinned H2D / D2H rate of the link (GB/s) measured live, what a pageable hipMemcpy of the same size reaches, and one
    core's memcpy into pinned memory (what a staging loop is made of)."""
    n = mib << 20
    pin = torch.empty(n, dtype=torch.uint8, pin_memory=True)
    pin.fill_(1)
    d = torch.empty(n, dtype=torch.uint8, device=dev)
    page = np.ones(n, np.uint8)
    out = {"bytes": n}
    for name, fn in (("pinned_h2d", lambda: d.copy_(pin, non_blocking=True)), ("pinned_d2h", lambda: pin.copy_(d, non_blocking=True)),
                     ("pageable_h2d", lambda: d.copy_(torch.from_numpy(page)))):
        fn(); torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            t = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
        out[name + "_GBps"] = n / min(ts) / 1e9
    view = pin.numpy()
    t = time.perf_counter(); view[:] = page; out["memcpy_to_pinned_one_core_GBps"] = n / (time.perf_counter() - t) / 1e9
    return out


class SmallPageBuffer:
    """A fresh anonymous mapping with transparent huge pages switched OFF for it (MADV_NOHUGEPAGE): what a result array from
    glibc malloc / a std::vector / MATLAB's allocator is on a system whose THP mode is `madvise` or `never` (numpy asks for huge
    pages itself: its large arrays fault 512 times fewer pages and hide the cost of a fresh result array)."""

    def __init__(self, shape):
        import ctypes as C
        L = C.CDLL(None, use_errno=True)
        L.mmap.restype = C.c_void_p
        L.mmap.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_long]
        L.munmap.argtypes = [C.c_void_p, C.c_size_t]
        L.madvise.argtypes = [C.c_void_p, C.c_size_t, C.c_int]
        self._L, self.n = L, int(np.prod(shape))
        self.p = L.mmap(None, self.n, 3, 0x22, -1, 0)               # PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS
        assert self.p and self.p != C.c_void_p(-1).value
        L.madvise(self.p, self.n, 15)                               # MADV_NOHUGEPAGE
        self.a = np.ctypeslib.as_array((C.c_uint8 * self.n).from_address(self.p)).reshape(shape)

    def close(self):
        self.a = None
        self._L.munmap(self.p, self.n)


def host_batch_config(name, B, dev, pcie, settings=(("default", {}),), reps=3, seed=7):
    """One configuration through the HOST-POINTER batch entry points (polar_decode_scl_llr_batch / _f32 — the only path a MEX or
    PolarCode.hpp caller has: PolarCode.cpp:130-148, PolarM/PolarCode.m:312-322) from PAGEABLE numpy memory: codewords/s
    including staging, H2D, decode, D2H, next to the device-resident rate of the same batch and the bound
    min(device-resident rate, pinned PCIe rate / input bytes per codeword); bits compared with the device-resident decode."""
    import ctypes as C
    import polar_amd
    t_cfg = time.perf_counter()
    if name == "headline":
        n, K, crc, L, axis, const = 11, 1024, 16, 32, 2.0, "bpsk"
        C.CDLL(None).srand(C.c_uint(1))
        code = polar_amd.PolarCode(n, K, 0.32, crc)
    else:
        n, K, crc, L, _, axis, const, _, _ = OTHER_CONFIGS[name]
        code = make_config(name)
    N = 1 << n
    llr_d = torch.empty((B, N), dtype=torch.float64, device=dev)
    out_d = torch.empty((B, K), dtype=torch.uint8, device=dev)
    if const == "bpsk":
        code.synth_llr_dev(seed, 0, B, code.snr_sqrt_linear(axis), llr_d.data_ptr())
    else:
        code.synth_bicm_llr_dev(const, seed, 0, B, axis, llr_d.data_ptr())

    def timed(fn):
        ts = []
        for _ in range(reps):
            t = time.perf_counter(); fn(); ts.append(time.perf_counter() - t)
        return min(ts), float(np.median(ts))

    def dev_step():
        code.decode_scl_llr_dev(llr_d.data_ptr(), B, L, out_d.data_ptr()); torch.cuda.synchronize()
    dev_step()
    dmin, _ = timed(dev_step)
    want = out_d.cpu().numpy()
    llr64 = llr_d.cpu().numpy()                       # pageable host memory, as a caller's array is
    llr32 = llr64.astype(np.float32)
    l32_d = torch.from_numpy(llr32).to(dev)           # (float rows are another input: their own device-resident decode is the expectation)
    code.decode_scl_llr_dev_f32(l32_d.data_ptr(), B, L, out_d.data_ptr()); torch.cuda.synchronize()
    want32 = out_d.cpu().numpy()
    del llr_d, l32_d, out_d
    rec = {"config": name, "workload": f"N={N} K={K} crc={crc} L={L} {const}, batch {B}, LLRs in pageable host memory",
           "batch": B, "device_resident_cw_per_s": B / dmin, "rows": []}
    for label, kn in settings:
        for k in HOST_KNOBS:
            code.debug_set(k, kn.get(k, 0))
        for dt_name, a, w in (("f64", llr64, want), ("f32", llr32, want32)):
            got = code.decode_scl_llr(a, L)           # warm-up: staging slots, decode lanes
            ok = bool((got == w).all())
            for _ in range(6):                        # (a setter has just dropped the extra decode lanes: the first calls after it
                code.decode_scl_llr(a, L, out=got)    # rebuild them and run on freshly allocated device scratch — slower for ~0.2 s)
            # `value` is what a MEX gateway / std::vector caller sees: a NEW result buffer per call, small pages, never touched
            # (allocated before the clock, released after it: the library call alone). Beside it the rate with one result array
            # kept from call to call, and a new numpy array per call (huge pages by numpy's own madvise), allocation included.
            def fresh_call():
                b = SmallPageBuffer((B, K))
                t = time.perf_counter(); code.decode_scl_llr(a, L, out=b.a); dt = time.perf_counter() - t
                good = bool((b.a == w).all())
                b.close()
                return dt, good
            fr = [fresh_call() for _ in range(reps)]
            tfresh, tfresh_med = min(x[0] for x in fr), float(np.median([x[0] for x in fr]))
            ok = ok and all(x[1] for x in fr)
            us = {k: code.debug_get("host_us_" + k) for k in ("copy_in", "wait", "copy_out", "total")}
            tmin, tmed = timed(lambda: code.decode_scl_llr(a, L, out=got))
            ok = ok and bool((got == w).all())
            tnp, _ = timed(lambda: code.decode_scl_llr(a, L))
            pcie_bound = pcie["pinned_h2d_GBps"] * 1e9 / (N * a.itemsize)
            bound = min(B / dmin, pcie_bound)
            rec["rows"].append({"setting": label, "llr": dt_name, "value": B / tfresh, "unit": "codewords/s", "median": B / tfresh_med, "ms": tfresh * 1e3,
                                "value_is": "a fresh small-page result buffer per call (what a MEX gateway / C++ caller hands over), library call alone",
                                "reused_out_value": B / tmin, "reused_out_median": B / tmed, "fresh_over_reused": tmin / tfresh,
                                "fresh_numpy_out_value": B / tnp,
                                "input_GBps": B * N * a.itemsize / tfresh / 1e9, "pcie_bound_cw_per_s": pcie_bound,
                                "bound_cw_per_s": bound, "bound_by": "device" if B / dmin < pcie_bound else "pcie", "frac_of_bound": B / tfresh / bound,
                                "host_thread_us": us,
                                "bits_equal_device_resident": ok, "chunks": code.debug_get("host_chunks"),
                                "chunk_codewords": code.debug_get("host_chunk_cw"), "lanes": code.debug_get("host_lanes"),
                                "copy_threads": code.debug_get("host_threads")})
    rec["seconds_c_abi_rows"] = round(time.perf_counter() - t_cfg, 1)
    if len(settings) == 1:
        rec["mex_gateway"] = mex_gateway_rows(code, llr64, llr32, want, want32, L, B, N, K, reps)
    rec["seconds"] = round(time.perf_counter() - t_cfg, 1)
    for k in HOST_KNOBS:
        code.debug_set(k, 0)
    code.close()
    return rec


def mex_gateway_rows(code, llr64, llr32, want, want32, L, B, N, K, reps):
    """The same batch through the MEX gateway itself (polar_amd/matlab/polar_mex.cpp linked against the working mx runtime of
    tests/mex_runtime — MATLAB is not in the image): polar_mex('decode_scl_llr', h, llr, L) with llr N x B (one codeword per column:
    MATLAB's storage is the library's) — time inside mexFunction, result array created by the gateway on every call."""
    try:
        import fake_matlab
        mex = fake_matlab.polar_mex()
    except Exception as ex:
        return {"error": f"{type(ex).__name__}: {ex}"[:300]}
    args = [float(code.n), float(code.K), float(code.crc_size), code.frozen_bits.astype(np.uint8), code.channel_order_descending.astype(np.uint16)]
    if code.crc_size:
        args.append(code.crc_matrix.astype(np.uint8))
    h = mex('create_explicit', *args)
    rows = []
    try:
        for dt_name, a, w in (("f64", llr64, want), ("f32", llr32, want32)):
            cols = mex.value(a.T)                        # N x B: stored column-major = the rows of `a`; a workspace variable, converted once
            ts, ok = [], True
            for i in range(4):
                u = mex('decode_scl_llr', h, cols, float(L))
                if i >= 1:
                    ts.append(mex.last_call_seconds)
                ok = ok and u.shape == (K, B) and bool((u.T == w).all())
            cols.free()
            rows.append({"llr": dt_name, "layout": "N x B", "value": B / min(ts), "unit": "codewords/s", "median": B / float(np.median(ts)),
                         "ms_in_mexFunction": min(ts) * 1e3, "bits_equal_device_resident": ok})
    finally:
        mex('destroy', h, nlhs=0)
    return rows


def host_batch_record(args, dev):
    pcie = pcie_rates(dev)
    return {"abi": "polar_decode_scl_llr_batch / polar_decode_scl_llr_batch_f32 (host pointers: staging + H2D + decode + D2H inside the call)",
            "pcie": pcie, "configs": [host_batch_config(c, 65536, dev, pcie) for c in HOST_BATCH_CONFIGS]}


def p1_record(args, code):
    """The probability-domain members of the class surface (PolarCode::decode_scl_p1, PolarCode.cpp:110-128; PolarM decode_sc_p1,
    PolarCode.m:290-295) — no driver of the reference calls them — through their host-pointer entry points: codewords/s of one call
    of 8192 codewords of the benchmark code (two / one input arrays of doubles, staging included), next to decode_scl_llr through
    ITS host entry point at the same batch, and the first 8 codewords checked against the CPU restatement."""
    import oracle_lib
    o = oracle_lib.Oracle(args.n, args.K, 0.32, args.crc, srand=1)
    o.set_crc_matrix(code.crc_matrix)
    B = 8192
    d = torch.empty((B, code.N), dtype=torch.float64, device="cuda")
    code.synth_llr_dev(args.seed, 0, B, code.snr_sqrt_linear(args.ebno), d.data_ptr())
    llr = d.cpu().numpy()
    del d
    p1 = 1.0 / (1.0 + np.exp(llr))
    p0 = 1.0 - p1
    rows = []

    def timed(fn):
        fn()
        ts = []
        for _ in range(3):
            t = time.perf_counter(); r = fn(); ts.append(time.perf_counter() - t)
        return min(ts), r
    for name, fn, ref in (("decode_scl_p1 L=4", lambda: code.decode_scl_p1(p1, p0, 4), lambda: o.decode_scl_p1(p1[:8], p0[:8], 4)),
                          ("decode_scl_p1 L=32", lambda: code.decode_scl_p1(p1, p0, 32), lambda: o.decode_scl_p1(p1[:8], p0[:8], 32)),
                          ("decode_sc_p1", lambda: code.decode_sc_p1(p1), lambda: np.stack([o.decode_sc_p1(p1[i]) for i in range(8)])),
                          ("decode_scl_llr L=4 (for scale)", lambda: code.decode_scl_llr(llr, 4), None),
                          ("decode_scl_llr L=32 (for scale)", lambda: code.decode_scl_llr(llr, 32), None),
                          ("decode_scl_llr L=1 (for scale)", lambda: code.decode_scl_llr(llr, 1), None)):
        dt, got = timed(fn)
        rows.append({"call": name, "batch": B, "ms": dt * 1e3, "value": B / dt, "unit": "codewords/s",
                     "first_8_equal_cpu_restatement": (bool((np.asarray(got[:8]) == np.asarray(ref())).all()) if ref else None)})
    return {"abi": "polar_decode_scl_p1_batch / polar_decode_sc_p1_batch (host pointers)", "rows": rows}


def latency_record(args, code):
    """The reference's own call pattern — ONE codeword per call (PolarCode.cpp:756, PolarM/main_MC_CC_Comparison.m:96) — through the
    host-pointer ABI (staging + kernel + synchronisation + result copy), next to the CPU side on one core: median of 30 calls per
    list size on codewords of the benchmark workload, bits checked against the CPU's."""
    import ctypes as C
    import oracle_lib
    kind = "reference" if oracle_lib.have_reference() else "port"
    C.CDLL(None).srand(C.c_uint(1))
    cpu = (oracle_lib.Reference if kind == "reference" else oracle_lib.Oracle)(args.n, args.K, 0.32, args.crc, srand=1)
    o = oracle_lib.Oracle(args.n, args.K, 0.32, args.crc, srand=1)
    llr, _ = o.synth_llr(args.seed, 0, 8, o.snr_sqrt_linear(args.ebno))
    rows = []
    llr32 = llr.astype(np.float32)
    for L in (1, 2, 4, 8, 32):
        t = time.perf_counter()
        want = cpu.decode_scl_llr(llr, L)
        cpu_ms = (time.perf_counter() - t) / len(llr) * 1e3
        want32 = cpu.decode_scl_llr(llr32.astype(np.float64), L)
        ok = ok32 = True
        ts, ts32 = [], []
        for i in range(30):
            x = llr[i % 8]
            t = time.perf_counter()
            got = code.decode_scl_llr(x, L)
            ts.append(time.perf_counter() - t)
            ok = ok and bool((got == want[i % 8]).all())
            x = llr32[i % 8]
            t = time.perf_counter()
            got = code.decode_scl_llr(x, L)
            ts32.append(time.perf_counter() - t)
            ok32 = ok32 and bool((got == want32[i % 8]).all())
        rows.append({"L": L, "gpu_call_ms": float(np.median(ts)) * 1e3, "gpu_call_ms_f32": float(np.median(ts32)) * 1e3,
                     "cpu_ms_per_codeword_one_core": cpu_ms, "cpu_kind": kind, "bits_equal": ok, "bits_equal_f32": ok32})
    # PolarM's per-codeword call (main_MC_CC_Comparison.m:96): decode_sc_p1 on one vector of probabilities, next to the CPU
    # restatement of the same recursion (the reference's is MATLAB: 51 it/s on its author's laptop, README.md:65)
    p1 = 1.0 / (1.0 + np.exp(llr))
    t = time.perf_counter()
    want_p = np.stack([o.decode_sc_p1(p1[i]) for i in range(8)])
    cpu_p_ms = (time.perf_counter() - t) / 8 * 1e3
    ts, okp = [], True
    for i in range(30):
        t = time.perf_counter()
        got = code.decode_sc_p1(p1[i % 8])
        ts.append(time.perf_counter() - t)
        okp = okp and bool((got == want_p[i % 8]).all())
    rows.append({"call": "decode_sc_p1", "gpu_call_ms": float(np.median(ts)) * 1e3, "cpu_ms_per_codeword_one_core": cpu_p_ms, "cpu_kind": "port (C restatement of PolarM's recursion)",
                 "bits_equal": okp})
    # mid-size host batches (64 MiB of doubles): between one codeword per call and the pipelined batches of `host_batch` — one
    # copy in, ONE launch, one copy out since round 6 (the 32-MiB pipeline rule cost these calls a factor of 2-3)
    mid = []
    for L in (4, 32):
        want = cpu.decode_scl_llr(llr, L)
        x = np.ascontiguousarray(np.tile(llr, (512, 1)))
        got = code.decode_scl_llr(x, L)
        ok = bool((got == np.tile(want, (512, 1))).all())
        ts = []
        for _ in range(5):
            t = time.perf_counter(); code.decode_scl_llr(x, L); ts.append(time.perf_counter() - t)
        mid.append({"L": L, "batch": len(x), "llr_MiB": x.nbytes / 2 ** 20, "gpu_call_ms": float(np.median(ts)) * 1e3, "value": len(x) / float(np.median(ts)),
                    "unit": "codewords/s", "bits_equal": ok, "chunks": code.debug_get("host_chunks")})
    return {"abi": "polar_decode_scl_llr (host pointers, one codeword per call)", "rows": rows, "mid_size_batches": mid}


def other_configs(args, dev):
    """The other BASELINE.json configurations in the same run (same library, same box), each on ITS OWN workload: decode +
    count, three steps, dominant-kernel time from HIP events, roofline, counter traffic from the committed per-config
    profile (refused unless taken with this very library), a reference CPU sample and the GPU-vs-CPU mismatch count."""
    return [run_config(name, args, dev) for name in OTHER_CONFIGS]


if __name__ == "__main__":
    main()
