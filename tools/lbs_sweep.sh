#!/bin/bash
# small-sort time of library variants: tools/lbs_sweep.sh <tag>...   (tools/size_sweep.py with the default shapes)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
for round in 1 2; do
  for v in "$@"; do
    echo "== '$v'"
    LSDSORT_LIB=$REPO/lsdradixsort_amd/liblsdsort$v.so python $REPO/tools/size_sweep.py --cfgs -1 --sizes 14 16 18 19 20 21 22 2>/dev/null
  done
done
