#!/bin/bash
# Write-path counters for the rank-and-scatter kernel (one PMC group per run).
#   usage: tools/profile_write.sh <tag> [prof_target.py args...]
set -o pipefail
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
TARGET_ARGS="$*"
run() { local name=$1; shift
  rocprofv3 "$@" -d $OUT/$name --output-format csv -- python3 $REPO/tools/prof_target.py $TARGET_ARGS > $OUT/$name.log 2>&1 || { echo "pass $name failed"; tail -5 $OUT/$name.log; return 1; }
}
run trace --kernel-trace --stats &&
run pmc_w1 --pmc TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum &&
run pmc_w2 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_WRITEBACK_sum &&
run pmc_w3 --pmc TA_BUSY_avr TA_BUSY_max SQ_INST_CYCLES_VMEM_WR SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE &&
run pmc_w4 --pmc TCC_WRITE_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum &&
run pmc_w5 --pmc TCC_EA0_WRREQ_DRAM_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_LEVEL_sum &&
python3 $REPO/tools/prof_summary.py $OUT > $OUT/summary.txt; grep -A40 "rank_scatter" $OUT/summary.txt | head -60
