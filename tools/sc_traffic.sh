#!/bin/bash
# tools/sc_traffic.sh <lib-tag> — GPU box: HBM bytes of the list-size-1 kernels (FETCH_SIZE x2, WRITE_SIZE x1: profiles/r02/fetch_calibration.txt)
R=$PWD; T=${1:-}; LIB=$R/polar_amd/libpolar_amd${T:+_$T}.so
cd /tmp && export TMPDIR=/tmp
for PMC in FETCH_SIZE WRITE_SIZE; do
  POLAR_AMD_LIB=$LIB rocprofv3 --kernel-trace --pmc $PMC -d $R/gpurun_out/sct_$PMC -o pmc --output-format csv -- python $R/tools/sc_time.py 262144 2 > $R/gpurun_out/sct_$PMC.log 2>&1
done
python - <<PY
import csv, collections
tot = collections.defaultdict(float); n = collections.Counter()
for c, f in (("FETCH_SIZE", 2048.0), ("WRITE_SIZE", 1024.0)):
    for r in csv.DictReader(open("$R/gpurun_out/sct_%s/pmc_counter_collection.csv" % c)):
        k = r["Kernel_Name"][:24]
        if r["Counter_Name"] == c and k.startswith("sc8") or k.startswith("void sc8"):
            tot[(k, c)] += float(r["Counter_Value"]) * f; n[(k, c)] += 1
for (k, c), v in sorted(tot.items()):
    print(f"{k:26s} {c:10s} {v / n[(k, c)] / 1e9:8.2f} GB per launch ({n[(k, c)]} launches, 262144 codewords)")
PY
