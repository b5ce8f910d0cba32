#!/usr/bin/env python3
"""Per-stage device times for every compiled tile shape and both pass structures (GPU box)."""
import argparse
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import lsdradixsort_amd as lsd
from bench import mt19937_keys

ap = argparse.ArgumentParser()
ap.add_argument("--log2-keys", type=int, default=28)
ap.add_argument("--radix", type=int, nargs="*", default=[8, 4])
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--pairs", action="store_true")
ap.add_argument("--mask", type=lambda x: int(x, 0), default=0xFFFFFFFF)
ap.add_argument("--cfgs", type=int, nargs="*", default=None)
ap.add_argument("--chunk", type=int, nargs="*", default=[16])
ap.add_argument("--rank", type=int, nargs="*", default=[0, 2])
ap.add_argument("--algos", type=int, nargs="*", default=[0, 1])
args = ap.parse_args()
n = 1 << args.log2_keys
master = lsd.to_device(mt19937_keys(n, 0) & np.uint32(args.mask))
vals0 = torch.arange(n, dtype=torch.int32, device="cuda") if args.pairs else None
counts = {8: 6, 4: 4}
for r in args.radix:
    for cfg in (args.cfgs if args.cfgs is not None else range(counts[r])):
        lsd.set_tile_config(r, cfg)
        for rk, algo, ch in [(rk, al, ch) for rk in args.rank for al in args.algos for ch in args.chunk]:
            name = ("onesweep" if algo == 0 else "staged") + f"/rank{rk}/C{ch}"
            lsd.set_rank_method(rk)
            lsd.set_xcd_chunk(ch)
            ws = lsd.alloc_workspace(n, r, args.pairs, algo)
            sc, hi, scn, tot = [], [], [], []
            for i in range(args.reps + 1):
                k = master.clone()
                v = vals0.clone() if args.pairs else None
                tm = lsd.GPULSDRadixSortTimed(k, r, d_vals=v, algorithm=algo, workspace=ws)
                if i == 0:
                    continue
                sc += tm["scatter_ms"]; hi.append(tm["histogram_ms"]); scn.append(tm["scan_ms"]); tot.append(tm["total_ms"])
            u = k.to(torch.int64) & 0xFFFFFFFF
            ok = bool((u[1:] >= u[:-1]).all())
            per = (16 if args.pairs else 8) * n
            print(f"r={r} cfg={cfg} tile={tm['tile_keys']:6d} {name:18s} total={np.mean(tot):7.3f} ms  hist={np.mean(hi):6.3f}  "
                  f"scan={np.mean(scn):6.3f}  scatter/pass={np.mean(sc):6.3f} ms (min {np.min(sc):6.3f}) = {per/np.mean(sc)/1e6:7.1f} GB/s  sorted={ok}",
                  flush=True)
    lsd.set_tile_config(r, -1)
