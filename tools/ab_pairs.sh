#!/bin/bash
# interleaved A/B of (library tag, tile config) pairs through bench.py: tools/ab_pairs.sh tag:cfg ...  ("" tag = product library)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
for round in 1 2 3; do
  for pair in "$@"; do
    v=${pair%%:*}; cfg=${pair##*:}
    out=$(LSDSORT_LIB=$REPO/lsdradixsort_amd/liblsdsort$v.so python $REPO/bench.py --tile-config $cfg --steps 20 --warmup 3 --no-cpu-baseline --no-extra 2>/dev/null | tail -1)
    echo "$pair $(echo "$out" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["stages_ms"]["histogram"], d["stages_ms"]["scatter_per_pass"], d["config"]["tile_keys"])')"
  done
done
