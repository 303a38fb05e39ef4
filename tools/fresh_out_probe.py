#!/usr/bin/env python3
"""tools/fresh_out_probe.py — where the "fresh output array" penalty of the host-pointer batch path goes (VERDICT r05 weak 3:
a new result array per call, as a MEX gateway / std::vector returns it, was 9-25 % slower than a reused one).

Per configuration (bench.py HOST_BATCH_CONFIGS) and input type, medians of `reps` calls of polar_decode_scl_llr_batch[_f32]:
  reused            the caller keeps one output array
  fresh_call        a NEW, never-touched array per call, only the library call timed (allocation before, release after the clock)
  touched_call      a new array whose pages the harness wrote before the clock (separates page faults from everything else)
  fresh_loop        allocation + call + release inside the clock (what bench.py's fresh_out_value has always timed)
  alloc_free        allocation + release of an untouched / a touched array alone (the harness's own mmap / munmap cost)
each for the library's prefault threads off (-1), default (0) and 2 / 4 / 8, with the calling thread's breakdown
(copy in / wait for the device / copy out / total, microseconds).

    python tools/fresh_out_probe.py [--configs config2,config3,config5,headline] [--batch 65536] [--out gpurun_out/fresh_out_probe.json]
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np
import torch

import bench


SmallPageBuffer = bench.SmallPageBuffer


def med(f, reps, before=None, after=None):
    ts = []
    for _ in range(reps):
        ctx = before() if before else None
        t = time.perf_counter()
        r = f(ctx)
        ts.append(time.perf_counter() - t)
        if after:
            after(ctx, r)
        del r, ctx
    return float(np.median(ts)), float(min(ts))


def probe(name, B, dev, reps):
    import polar_amd
    if name == "headline":
        n, K, crc, L, axis, const = 11, 1024, 16, 32, 2.0, "bpsk"
        C.CDLL(None).srand(C.c_uint(1))
        code = polar_amd.PolarCode(n, K, 0.32, crc)
    else:
        n, K, crc, L, _, axis, const, _, _ = bench.OTHER_CONFIGS[name]
        code = bench.make_config(name)
    N = 1 << n
    d = torch.empty((B, N), dtype=torch.float64, device=dev)
    if const == "bpsk":
        code.synth_llr_dev(7, 0, B, code.snr_sqrt_linear(axis), d.data_ptr())
    else:
        code.synth_bicm_llr_dev(const, 7, 0, B, axis, d.data_ptr())
    llr64 = d.cpu().numpy()
    del d
    rows = []
    shape = (B, K)
    for dt_name, a in (("f64", llr64), ("f32", llr64.astype(np.float32))):
        keep = code.decode_scl_llr(a, L)
        for pf in (-1, 0, 2, 4):
            code.debug_set("host_prefault", pf)
            for _ in range(12):                      # (a setter drops the extra decode lanes: the first calls after it rebuild them
                code.decode_scl_llr(a, L, out=keep)  # and run on freshly allocated device scratch — 150-200 ms of slower calls)
            r = {"config": name, "llr": dt_name, "batch": B, "prefault_threads": pf}
            m, mn = med(lambda _: code.decode_scl_llr(a, L, out=keep), reps)
            r["reused_ms"], r["reused_min_ms"] = m * 1e3, mn * 1e3
            r["reused_us"] = {k: code.debug_get("host_us_" + k) for k in ("copy_in", "wait", "copy_out", "total")}
            m, mn = med(lambda o: code.decode_scl_llr(a, L, out=o), reps, before=lambda: np.empty(shape, np.uint8))
            r["fresh_call_ms"], r["fresh_call_min_ms"] = m * 1e3, mn * 1e3
            r["fresh_us"] = {k: code.debug_get("host_us_" + k) for k in ("copy_in", "wait", "copy_out", "total")}

            def touched():
                o = np.empty(shape, np.uint8)
                o.fill(1)
                return o
            m, mn = med(lambda o: code.decode_scl_llr(a, L, out=o), reps, before=touched)
            r["touched_call_ms"] = m * 1e3
            m, mn = med(lambda _: code.decode_scl_llr(a, L), reps)
            r["fresh_loop_ms"], r["fresh_loop_min_ms"] = m * 1e3, mn * 1e3
            # small pages (no THP): fresh per call, and the same buffer kept
            m, mn = med(lambda b: code.decode_scl_llr(a, L, out=b.a), reps, before=lambda: SmallPageBuffer(shape), after=lambda b, r_: b.close())
            r["fresh_smallpage_call_ms"], r["fresh_smallpage_call_min_ms"] = m * 1e3, mn * 1e3
            r["fresh_smallpage_us"] = {k: code.debug_get("host_us_" + k) for k in ("copy_in", "wait", "copy_out", "total")}
            sb = SmallPageBuffer(shape)
            sb.a.fill(1)
            m, mn = med(lambda _: code.decode_scl_llr(a, L, out=sb.a), reps)
            r["reused_smallpage_ms"] = m * 1e3
            assert (sb.a == keep).all()
            sb.close()
            assert (code.decode_scl_llr(a, L) == keep).all()
            r["cw_per_s"] = {k: B / (r[k + "_ms"] / 1e3) for k in ("reused", "fresh_call", "touched_call", "fresh_loop", "fresh_smallpage_call", "reused_smallpage")}
            r["fresh_smallpage_over_reused"] = r["reused_smallpage_ms"] / r["fresh_smallpage_call_ms"]
            r["fresh_call_over_reused"] = r["reused_ms"] / r["fresh_call_ms"]
            r["fresh_loop_over_reused"] = r["reused_ms"] / r["fresh_loop_ms"]
            rows.append(r)
            print(json.dumps(r), flush=True)
        code.debug_set("host_prefault", 0)
    m1, _ = med(lambda _: np.empty(shape, np.uint8), reps)

    def touch(_):
        o = np.empty(shape, np.uint8)
        o.fill(1)
        return o
    m2, _ = med(touch, reps)
    m3, _ = med(lambda _: np.zeros(shape, np.uint8), reps)
    extra = {"config": name, "alloc_free_untouched_ms": m1 * 1e3, "alloc_touch_free_ms": m2 * 1e3, "np_zeros_free_ms": m3 * 1e3, "out_bytes": B * K}
    print(json.dumps(extra), flush=True)
    code.close()
    return rows, extra


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", default="config2,config3,config5,headline")
    ap.add_argument("--batch", type=int, default=65536)
    ap.add_argument("--reps", type=int, default=7)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "fresh_out_probe.json"))
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    try:
        thp = open("/sys/kernel/mm/transparent_hugepage/enabled").read().strip()
    except OSError:
        thp = None
    res = {"transparent_hugepage": thp, "usable_cpus": len(os.sched_getaffinity(0)), "rows": [], "alloc": []}
    for c in a.configs.split(","):
        rows, extra = probe(c, a.batch, dev, a.reps)
        res["rows"] += rows
        res["alloc"].append(extra)
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(res, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
