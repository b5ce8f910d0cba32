#!/bin/bash
# A/B library variants with bench.py itself (same process shape as the driver's run): tools/ab_bench.sh <tag>...
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
for round in 1 2; do
  for v in "$@"; do
    lib=$REPO/lsdradixsort_amd/liblsdsort$v.so
    out=$(LSDSORT_LIB=$lib python $REPO/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra 2>/dev/null | tail -1)
    echo "$v $(echo "$out" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["stages_ms"]["histogram"], d["stages_ms"]["scatter_per_pass"], d["config"]["tile_keys"])')"
  done
done
