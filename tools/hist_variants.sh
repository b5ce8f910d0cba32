#!/bin/bash
# stage-1 time of library variants: tools/hist_variants.sh <tag>...   (first three lines of tools/hist_size_sweep.py each)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
for round in 1 2; do
  for v in "$@"; do
    echo "== $v"
    LSDSORT_LIB=$REPO/lsdradixsort_amd/liblsdsort$v.so python $REPO/tools/hist_size_sweep.py 2>/dev/null | head -2
  done
done
