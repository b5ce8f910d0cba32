#!/bin/bash
# A/B two checked-out trees with their own bench.py on one box: tools/ab_trees.sh <dir> <dir> ...   (interleaved, three rounds)
for round in 1 2 3; do
  for d in "$@"; do
    out=$(cd $d && python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra 2>/dev/null | tail -1)
    echo "$d $(echo "$out" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["stages_ms"]["histogram"], d["stages_ms"]["scatter_per_pass"])')"
  done
done
