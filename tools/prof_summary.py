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

# kernel trace: durations.  A kernel's launches are not all alike: a pass the device-side plan skips returns at once (~8 us, same
# grid), so means over every dispatch say little.  The program is deterministic, so the i-th dispatch of a kernel is the same
# launch in every profiler pass: the trace marks the ones that did work (>= 10 % of the kernel's longest), and the counter
# passes are averaged over exactly those.
working = {}
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_trace.csv"), recursive=True):
    dur = defaultdict(list)
    for row in sorted(csv.DictReader(open(f)), key=lambda r: int(r["Dispatch_Id"])):
        dur[short(row["Kernel_Name"])].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
    for k, v in dur.items():
        mask = [x >= 0.1 * max(v) for x in v]
        working[k] = mask
        w = [x for x, m in zip(v, mask) if m]
        summary.setdefault(k, {})["calls"] = len(v)
        summary[k]["working_calls"] = len(w)
        summary[k]["avg_us"] = sum(w) / len(w) / 1e3
        summary[k]["min_us"] = min(w) / 1e3
# counters
for f in glob.glob(os.path.join(out, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    acc = defaultdict(lambda: defaultdict(dict))
    for row in csv.DictReader(open(f)):
        d = acc[short(row["Kernel_Name"])][row["Counter_Name"]]
        d[int(row["Dispatch_Id"])] = d.get(int(row["Dispatch_Id"]), 0.0) + float(row["Counter_Value"])
    for k, d in acc.items():
        for c, by_id in d.items():
            v = [by_id[i] for i in sorted(by_id)]
            mask = working.get(k)
            if mask is not None and len(mask) == len(v):
                v = [x for x, m in zip(v, mask) if m]
            else:
                summary.setdefault(k, {})["counters_over_all_calls"] = 1
            summary.setdefault(k, {})[c] = sum(v) / len(v)
json.dump(summary, open(os.path.join(out, "summary.json"), "w"), indent=1)
for k in sorted(summary, key=lambda k: -summary[k].get("avg_us", 0) * summary[k].get("calls", 1)):
    s = summary[k]
    print(f"== {k}  calls={s.get('calls')} (working: {s.get('working_calls')}) avg={s.get('avg_us', 0):.1f} us min={s.get('min_us', 0):.1f} us"
          + ("  [counters: mean over ALL calls]" if s.get("counters_over_all_calls") else ""))
    for c in sorted(s):
        if c not in ("calls", "working_calls", "avg_us", "min_us", "counters_over_all_calls"):
            print(f"     {c:28s} {s[c]:.4g}")
