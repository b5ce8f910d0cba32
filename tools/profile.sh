#!/bin/bash
# rocprofv3 passes over tools/prof_target.py on the GPU box; summaries land in gpurun_out/prof_<tag>/.
# Counters are collected in their own runs (never together with sys/hip tracing), one PMC
# group per run (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE do not fit one pass).
#   usage: tools/profile.sh <tag> [prof_target.py args...]
set -o pipefail
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
rm -rf $OUT   # files of an earlier run with the same tag would be averaged in
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { # name, rocprof args...
  local name=$1; shift
  rocprofv3 "$@" -d $OUT/$name --output-format csv -- python3 $REPO/tools/prof_target.py $TARGET_ARGS > $OUT/$name.log 2>&1 || { echo "pass $name failed"; tail -5 $OUT/$name.log; return 1; }
}
TARGET_ARGS="$*"
run trace --kernel-trace --stats &&
run pmc_sq1 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS &&
run pmc_sq2 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU &&
run pmc_fetch --pmc FETCH_SIZE &&
run pmc_write --pmc WRITE_SIZE TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum &&
run pmc_l2 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum &&
python3 $REPO/tools/prof_summary.py $OUT > $OUT/summary.txt && cat $OUT/summary.txt
