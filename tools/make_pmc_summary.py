#!/usr/bin/env python3
"""profiles/<tag>/summary.json (tools/profile.sh + prof_summary.py) -> profiles/pmc_summary.json.

HBM bytes per launch of the rank-and-scatter kernel, as MI355X_MICROARCH.md prescribes for
gfx950: FETCH_SIZE and WRITE_SIZE come from SEPARATE --pmc passes, are in KiB, and FETCH_SIZE
reports exactly half of the bytes of a coalesced streaming read (128-byte requests tallied at
64 B), so it is doubled.  The doubling is calibrated here on a kernel whose read volume is known:
the upfront histogram kernel reads every key exactly once (4*n bytes) and writes next to nothing.
"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = {}
for tag, key, n, pairs in (("r3_keys_r8", "rank_scatter_r8", 1 << 28, False), ("r3_keys_r4", "rank_scatter_r4", 1 << 28, False),
                           ("r3_pairs_r8", "rank_scatter_r8_pairs", 1 << 27, True)):
    path = os.path.join(ROOT, "profiles", tag, "summary.json")
    if not os.path.exists(path):
        continue
    s = json.load(open(path))
    rs = next(v for k, v in s.items() if k.startswith("rank_scatter_kernel"))
    # the upfront read that ran (a sort that takes the hybrid form also launches the ordinary form's, which returns at once)
    hists = [v for k, v in s.items() if "histograms_kernel" in k and "FETCH_SIZE" in v]
    hist = max(hists, key=lambda v: v["FETCH_SIZE"]) if hists else None
    fetch = rs["FETCH_SIZE"] * 1024 * 2
    write = rs["WRITE_SIZE"] * 1024
    algorithmic = (16 if pairs else 8) * n
    entry = {"n": n, "kernel_avg_us": round(rs["avg_us"], 1), "fetch_bytes_corrected": int(fetch), "write_bytes": int(write),
             "hbm_bytes_per_launch": int(fetch + write), "algorithmic_bytes_per_launch": algorithmic,
             "traffic_over_algorithmic": round((fetch + write) / algorithmic, 4),
             "partial_write_requests": int(rs["TCC_EA0_WRREQ_sum"] - rs["TCC_EA0_WRREQ_64B_sum"]),
             "source": f"profiles/{tag}/summary.json"}
    if hist:
        entry["calibration"] = {"kernel": "upfront histogram (reads 4*n bytes once)", "known_read_bytes": 4 * n,
                                "fetch_size_x2_bytes": int(hist["FETCH_SIZE"] * 1024 * 2),
                                "ratio": round(hist["FETCH_SIZE"] * 1024 * 2 / (4 * n), 4)}
    out[key] = entry
json.dump(out, open(os.path.join(ROOT, "profiles", "pmc_summary.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
