#!/bin/bash
# HBM traffic of the training step's kernels (PMC, two separate passes) -> gpurun_out/prof2/r2_train_hbm_traffic.json
R="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$R/gpurun_out/prof2"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 200 rocprofv3 --kernel-trace --pmc $ctr -d "$OUT/tpmc_$ctr" -o out --output-format csv -- python3 "$R/tools/train_bench.py" --dtype bf16x3 --steps 2 >> "$OUT/train_log.txt" 2>&1 || exit 1
done
cd "$R"
python3 tools/pmc_summary.py traffic "$OUT/tpmc_FETCH_SIZE" "$OUT/tpmc_WRITE_SIZE" "$OUT/r2_train_hbm_traffic.json" > /dev/null
rm -rf "$OUT"/tpmc_*
