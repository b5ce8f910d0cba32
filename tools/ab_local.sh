#!/bin/bash
# A/B library variants on the hybrid form's stages with bench.py: tools/ab_local.sh <tag>...   ("" = the product)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
for round in 1 2; do
  for v in "$@"; do
    lib=$REPO/lsdradixsort_amd/liblsdsort$v.so
    out=$(LSDSORT_LIB=$lib timeout -k 10 100 python $REPO/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra --no-live-traffic $AB_ARGS 2>/dev/null | tail -1)
    echo "lib$v $(echo "$out" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); s=d["stages_ms"]; print(d["value"], d["ms_per_step"], s["histogram"], s["scatter_per_pass"], s.get("local_stage"))')"
  done
done
