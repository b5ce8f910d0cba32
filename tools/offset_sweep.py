#!/usr/bin/env python3
"""Does the sort's time depend on where its buffers sit relative to each other (HBM channel / bank interleave)?
Same keys, same workspace bytes, the workspace and the key buffer shifted by various pads inside larger allocations."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lsdradixsort_amd as lsd
from bench import mt19937_keys

n = 1 << 28
master = lsd.to_device(mt19937_keys(n, 0))
need = lsd.workspace_bytes(n, 8)
big_ws = torch.empty(need + (64 << 20), dtype=torch.uint8, device="cuda")
big_keys = torch.empty(n + (16 << 20), dtype=torch.int32, device="cuda")

def timed(ws, keys_view, reps=12):
    ts = []
    for i in range(reps + 2):
        keys_view.copy_(master)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        lsd.GPULSDRadixSort(keys_view, 8, workspace=ws)
        e1.record()
        torch.cuda.synchronize()
        if i >= 2:
            ts.append(e0.elapsed_time(e1))
    return float(np.median(ts)), float(np.min(ts))

for round_ in range(3):
    for ws_pad in (0, 12288, 16 << 20, 32 << 20, (32 << 20) + 12288, (48 << 20) + 12288, 33 << 20, (24 << 20) + 8192, 60 << 20):
        for key_pad in (0,):     # in int32 elements
            ws = big_ws[ws_pad: ws_pad + need]
            kv = big_keys[key_pad: key_pad + n]
            med, mn = timed(ws, kv)
            print(f"ws_pad={ws_pad:10d} key_pad={key_pad * 4:8d} B   median {med:.4f} ms   min {mn:.4f} ms", flush=True)
