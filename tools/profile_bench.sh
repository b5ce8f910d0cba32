#!/bin/bash
# rocprofv3 --kernel-trace --stats over bench.py itself (the command whose JSON line is reported), so that the
# profiler's average duration of the dominant kernel can be held against bench.py's own HIP-event figure.
#   usage: tools/profile_bench.sh <tag> [bench.py args...]   -> gpurun_out/prof_<tag>/{kernel_stats.csv,bench.json}
set -o pipefail
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace --output-format csv -- python3 $REPO/bench.py "$@" > $OUT/bench.log 2>&1 || { echo "bench under rocprofv3 failed"; tail -5 $OUT/bench.log; exit 1; }
grep '^{' $OUT/bench.log | tail -1 > $OUT/bench.json
cp "$(ls -t $(find $OUT/trace -name '*kernel_stats.csv') | head -1)" $OUT/kernel_stats.csv
head -6 $OUT/kernel_stats.csv
python3 - "$OUT" <<'PY'
import csv, json, sys
out = sys.argv[1]
b = json.loads(open(out + "/bench.json").read())
rows = list(csv.DictReader(open(out + "/kernel_stats.csv")))
rs = [r for r in rows if "rank_scatter_kernel" in r["Name"]]
top = max(rs, key=lambda r: int(r["Calls"]))
print("bench.py  scatter_per_pass (HIP events): %.1f us   roofline.achieved %.0f GB/s" % (b["stages_ms"]["scatter_per_pass"] * 1e3, b["roofline"]["achieved"]))
print("rocprofv3 %s: calls %s avg %.1f us" % (top["Name"][:60], top["Calls"], float(top["AverageNs"]) / 1e3))
PY
