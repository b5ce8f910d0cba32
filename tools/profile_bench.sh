#!/bin/bash
# rocprofv3 --kernel-trace --stats over bench.py itself (the command whose JSON line is reported), so that the
# profiler's average duration of the dominant kernel can be held against bench.py's own HIP-event figure.
#   usage: tools/profile_bench.sh <tag> [bench.py args...]   -> gpurun_out/prof_<tag>/{kernel_stats.csv,bench.json}
set -o pipefail
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
rm -rf $OUT   # files of an earlier run with the same tag would be averaged in
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace --output-format csv -- python3 $REPO/bench.py "$@" > $OUT/bench.log 2>&1 || { echo "bench under rocprofv3 failed"; tail -5 $OUT/bench.log; exit 1; }
grep '^{' $OUT/bench.log | tail -1 > $OUT/bench.json
cp "$(ls -t $(find $OUT/trace -name '*kernel_stats.csv') | head -1)" $OUT/kernel_stats.csv
head -6 $OUT/kernel_stats.csv
python3 - "$OUT" <<'PY' | tee $OUT/compare.txt
# The stats row of a kernel averages EVERY dispatch of it -- also the ones that return at once (a pass the plan skips: ~8 us) and
# the ones of the secondary configs under "extra" (other sizes).  So next to that row: the dispatches of the trace itself, the
# long ones (> 60 us) of the headline's kernels, mean and median.
import csv, glob, json, statistics, sys
out = sys.argv[1]
b = json.loads(open(out + "/bench.json").read())
rows = list(csv.DictReader(open(out + "/kernel_stats.csv")))
trace = glob.glob(out + "/trace/*/*kernel_trace.csv")[0]
per = {}
for r in csv.DictReader(open(trace)):
    per.setdefault(r["Kernel_Name"], []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
rb = b["config"]["radix_bits"]
want = [("rank_scatter_kernel<%d, " % rb, b["roofline"]["launch_ms"], "roofline.launch_ms"),
        ("hybrid_histograms_kernel", b["roofline"].get("stage1", {}).get("launch_ms"), "roofline.stage1.launch_ms"),
        ("local_sort_kernel<", b["roofline"].get("local_stage", {}).get("launch_ms"), "roofline.local_stage.launch_ms")]
print("bench.py value %.1f %s, ms_per_step %.4f (under the profiler)" % (b["value"], b["unit"], b["ms_per_step"]))
for pat, ms, label in want:
    if ms is None:
        continue
    names = [k for k in per if pat in k]
    if not names:
        continue
    name = max(names, key=lambda k: len(per[k]))
    long_ = [x for x in per[name] if x > 60.0]
    st = next(r for r in rows if r["Name"] == name)
    print("%s\n  bench.py %s: %.1f us\n  rocprofv3 stats row: calls %s avg %.1f us (min %.1f: includes launches that return at once)\n"
          "  rocprofv3 trace, dispatches > 60 us: n %d mean %.1f us median %.1f us" %
          (name[:100], label, ms * 1e3, st["Calls"], float(st["AverageNs"]) / 1e3, float(st["MinNs"]) / 1e3,
           len(long_), statistics.mean(long_), statistics.median(long_)))
PY
