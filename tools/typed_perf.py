#!/usr/bin/env python3
"""Sort time by key type / order at 2^28 keys (GPU box): the transforms ride on the first and last pass."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lsdradixsort_amd as lsd
from bench import mt19937_keys

n = 1 << 28
raw = lsd.to_device(mt19937_keys(n, 0))
ws = lsd.alloc_workspace(n, 8)
for key_type, desc in (("uint32", False), ("int32", False), ("float32", False), ("uint32", True), ("float32", True)):
    ts = []
    for i in range(6):
        t = raw.clone()
        if key_type == "float32":
            t = t.view(torch.float32)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        lsd.GPUSortTyped(t, key_type, desc, workspace=ws)
        e1.record()
        torch.cuda.synchronize()
        if i:
            ts.append(e0.elapsed_time(e1))
    print(f"{key_type:8s} descending={desc!s:5s} {np.median(ts):.3f} ms  {n / np.median(ts) / 1e6:.1f} Gkeys/s", flush=True)
