#!/usr/bin/env python3
"""Idle time between the kernels of one sort, from a rocprofv3 kernel trace (the *_kernel_trace.csv of tools/profile.sh's
`trace` pass): per sort, first kernel start -> last kernel end against the sum of the kernel durations."""
import csv, glob, sys

path = sys.argv[1]
files = glob.glob(path + "/**/*kernel_trace.csv", recursive=True)
rows = []
for f in files:
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 4
# a sort = joint_histograms ... 4th rank_scatter after it
sorts, cur = [], None
for s, e, name in rows:
    if "joint_histograms" in name:
        cur = [(s, e, name)]
    elif cur is not None:
        cur.append((s, e, name))
        if sum("rank_scatter" in n for _, _, n in cur) == passes:
            sorts.append(cur)
            cur = None
if not sorts:
    sys.exit("no sorts found")
import statistics
span = [c[-1][1] - c[0][0] for c in sorts]
busy = [sum(e - s for s, e, _ in c) for c in sorts]
print(f"{len(sorts)} sorts: span median {statistics.median(span)/1e3:.1f} us, kernels {statistics.median(busy)/1e3:.1f} us, "
      f"idle between kernels {statistics.median([a - b for a, b in zip(span, busy)])/1e3:.1f} us")
c = sorts[len(sorts) // 2]
prev = None
for s, e, name in c:
    gap = "" if prev is None else f"  (+{(s - prev)/1e3:.1f} us after the previous kernel)"
    print(f"  {name[:60]:60s} {(e - s)/1e3:8.1f} us{gap}")
    prev = e
