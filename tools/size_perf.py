#!/usr/bin/env python3
"""Sort time at arbitrary sizes (GPU box): python tools/size_perf.py [--pairs] n [n ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import argparse, torch
import lsdradixsort_amd as lsd
ap = argparse.ArgumentParser()
ap.add_argument("sizes", type=float, nargs="+")
ap.add_argument("--pairs", action="store_true")
ap.add_argument("--radix", type=int, default=8)
a = ap.parse_args()
for nf in a.sizes:
    n = int(nf)
    k = torch.randint(-(1 << 31), (1 << 31) - 1, (n,), dtype=torch.int32, device="cuda")
    v = torch.arange(n, dtype=torch.int32, device="cuda") if a.pairs else None
    ws = lsd.alloc_workspace(n, a.radix, a.pairs)
    bufs = [(k.clone(), v.clone() if a.pairs else None) for _ in range(8)]
    for kk, vv in bufs[:2]:
        lsd.GPULSDRadixSort(kk, a.radix, d_vals=vv, workspace=ws)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for kk, vv in bufs[2:]:
        lsd.GPULSDRadixSort(kk, a.radix, d_vals=vv, workspace=ws)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 6
    print(f"{os.environ.get('LSDSORT_LIB', 'product')[-22:]:>22s} n={n} pairs={a.pairs} r={a.radix}: {ms:.3f} ms  {n / ms / 1e6:.1f} G/s  form {lsd.workspace_form(ws)}", flush=True)
