#!/bin/bash
# tools/variant_tune.sh "<cfgs>" <tag>... : per-kernel times of library variants (tools/tune.py), interleaved
CFGS=$1; shift
for round in 1 2; do for v in "$@"; do echo "variant '$v'"; LSDSORT_LIB=$PWD/lsdradixsort_amd/liblsdsort$v.so python tools/tune.py --radix 8 --rank 2 --reps 5 --cfgs $CFGS 2>&1 | grep "r=8"; done; done
