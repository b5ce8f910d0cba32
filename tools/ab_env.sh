#!/bin/bash
# A/B an environment switch with bench.py itself: tools/ab_env.sh VAR=a VAR=b ...   (interleaved, two rounds)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
for round in 1 2 3; do
  for kv in "$@"; do
    out=$(env "$kv" python $REPO/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra 2>/dev/null | tail -1)
    echo "$kv $(echo "$out" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["stages_ms"]["histogram"], d["stages_ms"]["scatter_per_pass"])')"
  done
done
