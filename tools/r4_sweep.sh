#!/bin/bash
# r = 4 sweep: tile configuration x rank method through bench.py (tools/r4_sweep.sh cfg...)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
for round in 1 2; do
  for cfg in "$@"; do
    for rk in 0 2; do
      out=$(python $REPO/bench.py --rank-method $rk --radix-bits 4 --tile-config $cfg --steps 10 --warmup 2 --no-cpu-baseline --no-extra 2>/dev/null | tail -1)
      echo "cfg=$cfg rank=$rk $(echo "$out" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["stages_ms"]["histogram"], d["stages_ms"]["scatter_per_pass"], d["config"]["tile_keys"])')"
    done
  done
done
