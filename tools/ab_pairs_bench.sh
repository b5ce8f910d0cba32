#!/bin/bash
# A/B of library variants on the key/value configuration (2^27 pairs): tools/ab_pairs_bench.sh <tag>...
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
for round in 1 2 3; do
  for v in "$@"; do
    out=$(LSDSORT_LIB=$REPO/lsdradixsort_amd/liblsdsort$v.so python $REPO/bench.py --pairs --log2-keys 27 --steps 20 --warmup 3 --no-cpu-baseline --no-extra 2>/dev/null | tail -1)
    echo "$v $(echo "$out" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["stages_ms"]["histogram"], d["stages_ms"]["scatter_per_pass"], d["config"]["tile_keys"])')"
  done
done
