#!/bin/bash
# lanes x trunk sub-batch sweep of the headline (bench.py --no-extras), one box
cd "$(dirname "$0")/../.."
for dt in f16mx6; do
for lanes in 1 2 3; do
  for sub in 2 4 8; do
    v=$(SR_LANES=$lanes SR_TRUNK_SUB=$sub timeout -k 10 120 python bench.py --dtype $dt --no-extras --steps 60 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")
    echo "$dt lanes=$lanes sub=$sub: $v"
  done
done
done
v=$(timeout -k 10 120 python bench.py --no-extras --steps 60 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")
echo "default: $v"
