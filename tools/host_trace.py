#!/usr/bin/env python3
"""tools/host_trace.py — one pipelined host-pointer decode under `rocprofv3 --kernel-trace`: the timeline of the chunks' kernels
(start, duration, queue) shows whether the decode lanes really overlap.

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/host_trace -- python tools/host_trace.py config3 8
    python tools/host_trace.py --report gpurun_out/host_trace
"""
import csv
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def report(d):
    f = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))[-1]
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # the last burst of scl_decode / sc8_decode kernels = the timed call
    dec = [r for r in rows if "decode" in r["Kernel_Name"]]
    t_end = int(dec[-1]["End_Timestamp"])
    burst = [r for r in rows if int(r["Start_Timestamp"]) > t_end - 200e6]
    # cut at the largest gap
    starts = [int(r["Start_Timestamp"]) for r in burst]
    gaps = [(starts[i + 1] - starts[i], i) for i in range(len(starts) - 1)]
    g, i = max(gaps)
    if g > 5e6:
        burst = burst[i + 1:]
    t0 = int(burst[0]["Start_Timestamp"])
    for r in burst:
        s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
        if e - s > 20000:
            print(f"{s / 1e6:8.3f} .. {e / 1e6:8.3f} ms ({(e - s) / 1e6:6.3f})  q{r['Queue_Id']:>3s}  grid {r['Grid_Size_X'] if 'Grid_Size_X' in r else r.get('Grid_Size', '?'):>7s}  {r['Kernel_Name'][:70]}")


if __name__ == "__main__":
    if sys.argv[1] == "--report":
        report(sys.argv[2])
        sys.exit(0)
    import numpy as np
    import torch
    import bench
    name, lanes = sys.argv[1], int(sys.argv[2])
    n, K, crc, L, _, axis, const, _, _ = bench.OTHER_CONFIGS[name]
    code = bench.make_config(name)
    B = 65536
    dev = torch.device("cuda", 0)
    llr_d = torch.empty((B, 1 << n), dtype=torch.float64, device=dev)
    code.synth_llr_dev(7, 0, B, code.snr_sqrt_linear(axis), llr_d.data_ptr())
    llr = llr_d.cpu().numpy()
    del llr_d
    code.debug_set("host_lanes", lanes)
    out = code.decode_scl_llr(llr, L)
    import time
    time.sleep(0.05)
    t = time.perf_counter()
    code.decode_scl_llr(llr, L, out=out)
    print("timed call %.2f ms" % ((time.perf_counter() - t) * 1e3))
