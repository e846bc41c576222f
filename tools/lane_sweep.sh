#!/bin/bash
# A/B of lanes x trunk sub-batch for the 180p bench on ONE box (boxes differ by ~10 %)
cd "$(dirname "$0")/.." || exit 1
run() { env "$@" timeout -k 10 100 python -u bench.py --steps 60 --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$*', d['value'], d['ms_per_step'])" || exit 1; }
run SR_SPLIT_TRUNK=0
run SR_SPLIT_TRUNK=1
run SR_LANES=2 SR_TRUNK_SUB=4
run SR_LANES=2 SR_TRUNK_SUB=2
run SR_LANES=1 SR_TRUNK_SUB=8
run SR_LANES=1 SR_TRUNK_SUB=4
run SR_LANES=3 SR_TRUNK_SUB=3
run SR_LANES=4 SR_TRUNK_SUB=4
