#!/usr/bin/env python3
"""tools/host_path.py — throughput of the HOST-POINTER batch entry points (polar_decode_scl_llr_batch[_f32]: the only path a
MEX / PolarCode.hpp caller can use, PolarCode.cpp:130-148, PolarM/PolarCode.m:312-322) from pageable numpy memory, next to the
device-resident rate of the same batch and the measured PCIe bound (bench.py: host_batch_config), with the pipeline's knobs
swept (--sweep) to choose the defaults.

    python tools/host_path.py [--sweep] [--configs config2,config3,config5,headline] [--batch 65536] [--out gpurun_out/host_path.json]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch

import bench

SWEEP = [("off (one copy in, decode, one copy out)", {"host_pipe_min_bytes": -1}),
         ("no ramp", {"host_ramp": -1}),
         ("one lane", {"host_lanes": 1}), ("two lanes", {"host_lanes": 2}), ("three lanes", {"host_lanes": 3}), ("four lanes", {"host_lanes": 4}),
         ("five lanes", {"host_lanes": 5}), ("six lanes", {"host_lanes": 6}), ("eight lanes", {"host_lanes": 8}),
         ("six lanes, chunk 64 MiB", {"host_lanes": 6, "host_chunk_bytes": 64 << 20}),
         ("chunk 32 MiB", {"host_chunk_bytes": 32 << 20}), ("chunk 64 MiB", {"host_chunk_bytes": 64 << 20}),
         ("chunk 128 MiB", {"host_chunk_bytes": 128 << 20}), ("chunk 256 MiB", {"host_chunk_bytes": 256 << 20}),
         ("chunk 256 MiB, 2 lanes", {"host_chunk_bytes": 256 << 20, "host_lanes": 2}),
         ("4 threads", {"host_threads": 4}), ("16 threads", {"host_threads": 16})]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", default="config2,config3,config5,headline")
    ap.add_argument("--batch", type=int, default=65536)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--sweep", action="store_true")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "host_path.json"))
    a = ap.parse_args()
    from polar_amd import build
    if not os.environ.get("POLAR_AMD_LIB"):
        build.build()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    pcie = bench.pcie_rates(dev)
    print(json.dumps(pcie), flush=True)
    import oracle_lib
    res = {"pcie": pcie, "lib_sha256": bench.lib_sha256(), "usable_cpus": oracle_lib.usable_cpus(), "nproc": os.cpu_count(), "configs": []}
    settings = [("default", {})] + (SWEEP if a.sweep else [])
    for c in a.configs.split(","):
        rec = bench.host_batch_config(c, a.batch, dev, pcie, settings=settings, reps=a.reps)
        res["configs"].append(rec)
        print(f"{c}: device-resident {rec['device_resident_cw_per_s'] / 1e6:.3f} M cw/s")
        for r in rec["rows"]:
            print(f"  {r['setting']:42s} {r['llr']} {r['value'] / 1e6:8.3f} M cw/s ({r['ms']:7.2f} ms, {r['input_GBps']:5.1f} GB/s in) bound {r['bound_cw_per_s'] / 1e6:7.3f} M ({r['bound_by']}) "
                  f"-> {r['frac_of_bound']:5.2f}  chunks {r['chunks']} x {r['chunk_codewords']} lanes {r['lanes']} threads {r['copy_threads']} ok={r['bits_equal_device_resident']} reused-out {r['reused_out_value'] / 1e6:.3f} M us {r['host_thread_us']}", flush=True)
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(res, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
