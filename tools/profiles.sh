#!/bin/bash
# Profile set of a round: tools/profiles.sh <tag> (e.g. r3) (run on the GPU box through gpurun; outputs under gpurun_out/prof_<tag>, summaries copied to profiles/ by hand).
R="$(cd "$(dirname "$0")/.." && pwd)"
TAG="${1:-r3}"
OUT="$R/gpurun_out/prof_$TAG"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
step() { echo "== $1" >> "$OUT/log.txt"; }
# 1. kernel stats + trace of the bench as the driver runs it (default dtype f16mx6, two lanes, trunk sub-batches; then the one-lane per-kernel pass, the extras)
step "stats default"
timeout -s KILL 400 rocprofv3 --kernel-trace --stats -d "$OUT/stats_default" -o out --output-format csv -- python3 "$R/bench.py" --steps 30 --warmup 5 > "$OUT/bench_under_rocprof.json" 2>> "$OUT/log.txt" || exit 1
# 2. one lane and whole-batch trunk launches (every kernel alone on the device), per dtype
export SR_LANES=1 SR_TRUNK_SUB=16
for dt in f16mx6 bf16x3; do
  step "stats one lane $dt"
  timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d "$OUT/stats_1lane_$dt" -o out --output-format csv -- python3 "$R/bench.py" --dtype $dt --no-extras --steps 10 --warmup 3 >> "$OUT/log.txt" 2>&1 || exit 1
done
# 3. HBM traffic, separate passes; both dtypes under one directory per counter (tools/pmc_summary.py merges what it finds)
for ctr in FETCH_SIZE WRITE_SIZE; do
  for dt in f16mx6 bf16x3; do
    step "pmc $ctr $dt"
    timeout -s KILL 300 rocprofv3 --kernel-trace --pmc $ctr -d "$OUT/pmc_$ctr/$dt" -o out --output-format csv -- python3 "$R/bench.py" --dtype $dt --no-extras --steps 3 --warmup 1 >> "$OUT/log.txt" 2>&1 || exit 1
  done
done
# 4. SQ counters (8 SQ + 1 GRBM)
for dt in f16mx6 bf16x3; do
  step "pmc sq $dt"
  timeout -s KILL 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE -d "$OUT/pmc_sq/$dt" -o out --output-format csv -- python3 "$R/bench.py" --dtype $dt --no-extras --steps 3 --warmup 1 >> "$OUT/log.txt" 2>&1 || exit 1
done
unset SR_LANES SR_TRUNK_SUB
# 5. the bf16x3 trunk kernel alone on random data (micro-benchmark): stats that contain nothing but res launches
step "trunk alone"
export SR_BENCH_TRUNK=1
timeout -s KILL 200 rocprofv3 --kernel-trace --stats -d "$OUT/stats_trunk" -o out --output-format csv -- python3 "$R/tools/conv_bench.py" --only res --iters 40 >> "$OUT/log.txt" 2>&1 || exit 1
unset SR_BENCH_TRUNK
# 6. training step and adversarial iteration
step "train"
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d "$OUT/stats_train" -o out --output-format csv -- python3 "$R/tools/train_bench.py" --dtype bf16x3 --steps 5 >> "$OUT/log.txt" 2>&1 || exit 1
step "gan"
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d "$OUT/stats_gan" -o out --output-format csv -- python3 "$R/tools/gan_bench.py" --iters 3 >> "$OUT/log.txt" 2>&1 || exit 1
cd "$R"
python3 tools/pmc_summary.py stats "$OUT/stats_default" > "$OUT/${TAG}_bench_f16mx6_kernel_stats.csv"
python3 tools/pmc_summary.py durations "$OUT/stats_default" > "$OUT/${TAG}_bench_f16mx6_launch_durations.csv"
python3 tools/pmc_summary.py stats "$OUT/stats_1lane_f16mx6" > "$OUT/${TAG}_bench_f16mx6_kernel_stats_1lane.csv"
python3 tools/pmc_summary.py stats "$OUT/stats_1lane_bf16x3" > "$OUT/${TAG}_bench_bf16x3_kernel_stats_1lane.csv"
python3 tools/pmc_summary.py stats "$OUT/stats_trunk" > "$OUT/${TAG}_trunk_kernel_stats.csv"
python3 tools/pmc_summary.py stats "$OUT/stats_train" > "$OUT/${TAG}_train_kernel_stats.csv"
python3 tools/pmc_summary.py stats "$OUT/stats_gan" > "$OUT/${TAG}_gan_kernel_stats.csv"
python3 tools/pmc_summary.py traffic "$OUT/pmc_FETCH_SIZE" "$OUT/pmc_WRITE_SIZE" "$OUT/${TAG}_hbm_traffic.json" > /dev/null
python3 tools/pmc_summary.py counters "$OUT/pmc_sq" SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE > "$OUT/${TAG}_sq_counters.txt"
# keep only the summaries (the raw traces are large)
rm -rf "$OUT"/stats_* "$OUT"/pmc_*
ls -la "$OUT"
