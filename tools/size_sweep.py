#!/usr/bin/env python3
"""Device-resident sort time by size and tile configuration (GPU box): picks the small-n defaults."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import argparse
import numpy as np, torch
import lsdradixsort_amd as lsd
from bench import mt19937_keys
ap = argparse.ArgumentParser()
ap.add_argument("--radix", type=int, default=8)
ap.add_argument("--cfgs", type=int, nargs="*", default=[-1, 5, 3, 0, 4])
ap.add_argument("--sizes", type=int, nargs="*", default=list(range(14, 27, 2)))
a = ap.parse_args()
master = lsd.to_device(mt19937_keys(1 << max(a.sizes), 0))
for lg in a.sizes:
    n = 1 << lg
    row = []
    for cfg in a.cfgs:
        lsd.set_tile_config(a.radix, cfg)
        ws = lsd.alloc_workspace(n, a.radix)
        reps = 20
        bufs = [master[:n].clone() for _ in range(reps + 2)]
        for b in bufs[:2]:
            lsd.GPULSDRadixSort(b, a.radix, workspace=ws)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for b in bufs[2:]:
            lsd.GPULSDRadixSort(b, a.radix, workspace=ws)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / reps * 1e3
        u = bufs[-1].to(torch.int64) & 0xFFFFFFFF
        assert bool((u[1:] >= u[:-1]).all())
        row.append(f"cfg{cfg:>2}: {us:8.1f} us {n / us / 1e3:7.2f} Gk/s")
    print(f"n=2^{lg:<2} | " + " | ".join(row), flush=True)
lsd.set_tile_config(a.radix, -1)
