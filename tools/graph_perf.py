#!/usr/bin/env python3
"""Small sorts: direct calls vs replay of a captured HIP graph (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lsdradixsort_amd as lsd
from bench import mt19937_keys

master = lsd.to_device(mt19937_keys(1 << 22, 0))
for lg in (12, 16, 18, 20, 22):
    n = 1 << lg
    ws = lsd.alloc_workspace(n, 8)
    buf = master[:n].clone()
    lsd.GPULSDRadixSort(buf, 8, workspace=ws)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        lsd.GPULSDRadixSort(buf, 8, workspace=ws)
    reps = 50
    def timed(fn):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3
    direct = timed(lambda: lsd.GPULSDRadixSort(buf, 8, workspace=ws))
    replay = timed(lambda: g.replay())
    print(f"n=2^{lg}: direct {direct:7.1f} us   graph replay {replay:7.1f} us", flush=True)
