#!/bin/bash
# bench.py over tile configurations, interleaved, two rounds: tools/shape_sweep.sh "<bench args>" cfg...
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
ARGS=$1; shift
for round in 1 2; do
  for cfg in "$@"; do
    out=$(python $REPO/bench.py $ARGS --tile-config $cfg --steps 20 --warmup 3 --no-cpu-baseline --no-extra 2>/dev/null | tail -1)
    echo "cfg=$cfg $(echo "$out" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["stages_ms"]["histogram"], d["stages_ms"]["scatter_per_pass"], d["config"]["tile_keys"])')"
  done
done
