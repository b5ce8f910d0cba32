#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (tools/profile.sh) into per-kernel means; also writes summary.json."""
import csv, glob, json, os, sys
from collections import defaultdict

out = sys.argv[1]
summary = {}

def short(name):
    name = name.split("(")[0]
    for tag in ("rank_scatter_kernel", "digit_histograms_kernel", "scan_digit_counts_kernel", "tile_histograms_kernel",
                "global_offsets_kernel", "strip_sums_kernel", "scan_strip_sums_kernel", "local_offsets_kernel"):
        if tag in name:
            return tag + name[name.find(tag) + len(tag):][:40]
    return name[:60]

# kernel trace: durations
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_trace.csv"), recursive=True):
    dur = defaultdict(list)
    for row in csv.DictReader(open(f)):
        dur[short(row["Kernel_Name"])].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
    for k, v in dur.items():
        summary.setdefault(k, {})["calls"] = len(v)
        summary[k]["avg_us"] = sum(v) / len(v) / 1e3
        summary[k]["min_us"] = min(v) / 1e3
# counters
for f in glob.glob(os.path.join(out, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    acc = defaultdict(lambda: defaultdict(list))
    for row in csv.DictReader(open(f)):
        acc[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, d in acc.items():
        for c, v in d.items():
            summary.setdefault(k, {})[c] = sum(v) / len(v)
json.dump(summary, open(os.path.join(out, "summary.json"), "w"), indent=1)
for k in sorted(summary, key=lambda k: -summary[k].get("avg_us", 0) * summary[k].get("calls", 1)):
    s = summary[k]
    print(f"== {k}  calls={s.get('calls')} avg={s.get('avg_us', 0):.1f} us min={s.get('min_us', 0):.1f} us")
    for c in sorted(s):
        if c not in ("calls", "avg_us", "min_us"):
            print(f"     {c:28s} {s[c]:.4g}")
