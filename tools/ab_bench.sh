#!/bin/bash
# Same-box A/B of bench.py: the tree in _ab_base/ (tools/ab_snapshot.sh <rev>) against the working tree, alternating, N rounds.
#   tools/ab_bench.sh [rounds] [bench.py args]
R=${1:-2}; shift
ms() { python3 -c "import json,sys; print(json.loads(sys.stdin.readline())['ms_per_step'])"; }
for i in $(seq $R); do
  a=$(cd _ab_base && timeout -k 10 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-fp32-mode --no-sampler --no-config5 "$@" 2>/dev/null | ms)
  b=$(timeout -k 10 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-fp32-mode --no-sampler --no-config5 "$@" 2>/dev/null | ms)
  echo "round $i: base $a ms/step   head $b ms/step"
done
