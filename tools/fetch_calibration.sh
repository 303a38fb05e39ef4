#!/bin/bash
# tools/fetch_calibration.sh — GPU box: calibrate rocprofv3 FETCH_SIZE / WRITE_SIZE for the decoder's access width
# (8 B per lane, 512-byte rows) on a kernel with KNOWN byte counts (tools/rowsize_microbench.hip: per launch
# 4096 waves x 4000 steps x (16 rows read + 8 rows written) x 512 B = 134.2 GB read, 67.1 GB written; the 1-KiB-row
# variant moves the same bytes as 16 B per lane). Prints counter value, unit factor and the correction.
set -u
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
hipcc --offload-arch=gfx950 -O3 -w $REPO/tools/rowsize_microbench.hip -o /tmp/rb || exit 1
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/cal_$C
  rocprofv3 --kernel-trace --pmc $C -d /tmp/cal_$C -o c --output-format csv -- /tmp/rb > /tmp/cal_$C.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections
known = {"FETCH_SIZE": 4096 * 4000 * 16 * 512, "WRITE_SIZE": 4096 * 4000 * 8 * 512}
for C in ("FETCH_SIZE", "WRITE_SIZE"):
    agg = collections.defaultdict(list)
    for f in glob.glob(f"/tmp/cal_{C}/**/c_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == C and "stream" in r["Kernel_Name"]:
                agg["1 KiB rows (16 B/lane)" if "<2>" in r["Kernel_Name"] else "512 B rows (8 B/lane)"].append(float(r["Counter_Value"]))
    for k, v in sorted(agg.items()):
        m = sum(v) / len(v)
        print(f"{C:10s} {k}: counter {m:.6g} per launch (KiB units -> {m * 1024 / 1e9:.2f} GB), known {known[C] / 1e9:.2f} GB, "
              f"correction factor known / (counter x 1024) = {known[C] / (m * 1024):.3f}")
PY
