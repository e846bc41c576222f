#!/bin/bash
# HBM-side bytes of one trunk-conv launch (conv_trunk.hip), rocprofv3 PMC, separate passes (FETCH_SIZE and WRITE_SIZE do not fit one)
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/pmc
for ctr in FETCH_SIZE WRITE_SIZE; do
  SR_BENCH_TRUNK=1 timeout -s KILL 120 rocprofv3 --pmc $ctr -d gpurun_out/pmc/$ctr -o out --output-format csv -- python3 tools/conv_bench.py --only res --iters 10 > gpurun_out/pmc/$ctr.log 2>&1 || { echo "pass $ctr failed"; tail -5 gpurun_out/pmc/$ctr.log; exit 1; }
done
python3 - <<'PY'
import csv, glob, collections
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    tot = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(f"gpurun_out/pmc/{ctr}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") == ctr:
                k = row["Kernel_Name"][:60]
                tot[k][0] += float(row["Counter_Value"]); tot[k][1] += 1
    for k, (v, n) in tot.items():
        print(ctr, k, "launches", n, "per launch (KB as reported)", round(v / n, 1))
PY
