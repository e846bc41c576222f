#!/bin/bash
# MFMA-shape probe of the trunk kernel (conv_trunk.hip, NOMEM & 8: two 16x16x32 MFMAs in place of every 32x32x16, wrong results,
# same operands / LDS reads / MFMA cycles): whole kernel and MFMA + LDS only, against the production shape, two rounds
cd "$(dirname "$0")/.."
for round in 1 2; do
for dbg in 0 8 5 13; do
  echo -n "dbg=$dbg  "
  SR_BENCH_TRUNK=1 SR_CONV_DBG=$dbg timeout -k 10 60 python tools/conv_bench.py --only res --iters 60 || exit 1
done
done
