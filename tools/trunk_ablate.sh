#!/bin/bash
# Ablation of the split-planar trunk kernel (conv_trunk.hip) on the 180p workload: which phase bounds a launch?
#   SR_CONV_DBG bits: 1 no tile DMA, 2 no MFMA loop, 4 no epilogue memory traffic
cd "$(dirname "$0")/.."
for dbg in 0 1 2 4 5 6 3 7; do
  echo -n "dbg=$dbg  "
  SR_BENCH_TRUNK=1 SR_CONV_DBG=$dbg timeout -k 10 60 python tools/conv_bench.py --only res --iters 40 || exit 1
done
for b in 4 8; do
  echo -n "batch=$b  "
  SR_BENCH_TRUNK=1 timeout -k 10 60 python tools/conv_bench.py --only res --iters 40 --batch $b || exit 1
done
