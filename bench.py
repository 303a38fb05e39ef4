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
# rate at which the decode kernel itself streams its state with the exact f-node compiled out
# (measurement build POLAR_EXPERIMENT_NO_EXACT_F): the traffic floor of this design
STREAM_PATTERN_GBS = 5870.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=65536, help="codewords per GPU per step (8 rounds of the 8192 codewords in flight)")
    ap.add_argument("--n", type=int, default=11)
    ap.add_argument("--K", type=int, default=1024)
    ap.add_argument("--crc", type=int, default=16)
    ap.add_argument("--L", type=int, default=32)
    ap.add_argument("--ebno", type=float, default=2.0)
    ap.add_argument("--seed", type=int, default=2024)
    ap.add_argument("--waves-per-cu", type=int, default=0)
    ap.add_argument("--lds-log", type=int, default=0)
    ap.add_argument("--cpu-sample", type=int, default=-1, help="codewords for the CPU baseline (-1 = auto, 0 = skip)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or os.environ.get("BENCH_FORCE_DIST"):
        import torch.distributed as dist_mod
        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    from polar_amd import build
    if rank == 0 and not os.environ.get("POLAR_AMD_LIB"):
        build.build()            # in-tree .so normally travels prebuilt; only one rank may compile
                                 # (an explicit POLAR_AMD_LIB — A/B runs — is used as it is)
    if dist:
        dist.barrier()
    import polar_amd

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
        dist.all_reduce(counters)        # RCCL sum of the Monte-Carlo counters (xGMI)
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if dist:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
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
            "kernel_ms_avg": kern_avg_s * 1e3,
            "algorithmic_bytes_per_codeword": alg_bytes_per_cw,
            "algorithmic_bytes_per_launch": B * alg_bytes_per_cw,
            "node_evals_per_s": (B * L * N * code.n) / kern_avg_s,
            # the resources this kernel is really bounded by: measured HBM bytes of the
            # committed PMC profile over the LIVE kernel time, and the profile's fp64 VALU occupancy
            "traffic_rate_GBps": (traffic / kern_avg_s / 1e9) if traffic else None,
            "traffic_frac_of_peak": (traffic / kern_avg_s / 1e9 / HBM_PEAK_GBS) if traffic else None,
            "traffic_frac_of_traffic_floor_rate": (traffic / kern_avg_s / 1e9 / STREAM_PATTERN_GBS) if traffic else None,
            "valu_busy_frac_in_profile": prof.get("valu_busy_frac_in_profile"),
        },
    }

    if rank == 0 and world == 1 and args.cpu_sample != 0:
        res["cpu_baseline"] = cpu_baseline(args, code, llr, out)
    if dist:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        sys.stdout.flush()
        print(json.dumps(res), flush=True)     # the ONE JSON line, last thing on stdout


def traffic_from_profile(args, B):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (profiles/traffic.json, written by tools/update_traffic.py from a tools/profile.sh run of this
    very command) — only when it was taken on the same workload; otherwise null."""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
        c = t["config"]
        if (c["n"], c["K"], c["crc"], c["L"], c["batch"]) == (args.n, args.K, args.crc, args.L, B):
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
    return {
        "value": sample / dt,
        "unit": "codewords/s",
        "cores": 1,
        "host_cores_available": os.cpu_count(),
        "kind": kind,
        "sample": f"first {sample} codewords of the benchmark batch, decode_scl_llr only, single thread",
        "gpu_vs_cpu_mismatching_codewords": mism,
    }


if __name__ == "__main__":
    main()
