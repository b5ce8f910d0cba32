#!/usr/bin/env python3
"""Stage-1 time against n: what part of it is per-key work and what part is the fixed cost of a grid (zeroing the LDS
tables, flushing 8192 counters per workgroup with global atomics).  Run on the GPU box."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lsdradixsort_amd as lsd
from bench import mt19937_keys

full = lsd.to_device(mt19937_keys(1 << 28, 0))
for log2n in (28, 27, 26, 25, 24, 23, 22, 21, 20):
    n = 1 << log2n
    ws = lsd.alloc_workspace(n, 8, False, lsd.LSDSORT_ALGO_ONESWEEP)
    best = None
    for rep in range(6):
        k = full[:n].clone()
        tm = lsd.GPULSDRadixSortTimed(k, 8, workspace=ws)
        if rep and (best is None or tm["histogram_ms"] < best["histogram_ms"]):
            best = tm
    print(f"n=2^{log2n}: histogram {best['histogram_ms']*1e3:7.1f} us  ({n*4/best['histogram_ms']/1e9:6.2f} TB/s)  scan {best['scan_ms']*1e3:5.1f} us  "
          f"scatter/pass {np.mean(best['scatter_ms'])*1e3:7.1f} us  total {best['total_ms']*1e3:7.1f} us", flush=True)
