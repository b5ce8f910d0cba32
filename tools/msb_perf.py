#!/usr/bin/env python3
"""MSB partition (multi-GPU step 1) and narrow-radix sort time by rank method (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lsdradixsort_amd as lsd
from bench import mt19937_keys

n = 1 << 28
d = lsd.to_device(mt19937_keys(n, 0))


def timed(fn, reps=5):
    ts = []
    for i in range(reps + 1):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        if i:
            ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))


sp = [int(x) for x in np.linspace(0, 2**32, 9)[1:-1]]
for k in (1, 3, 7):
    print(f"splitter partition, {k} splitters: {timed(lambda: lsd.SplitterPartition(d, sp[:k] if k < 7 else sp)):.3f} ms", flush=True)
for method in (0,):
    lsd.set_rank_method(method)
    for bits in (1, 2, 3):
        print(f"rank_method={method} msb_bits={bits} partition {timed(lambda: lsd.MSBPartition(d, bits)):.3f} ms", flush=True)
    for r in (2, 1):
        m = 1 << 26
        ws = lsd.alloc_workspace(m, r)
        src = d[:m]
        def run():
            k = src.clone()
            lsd.GPULSDRadixSort(k, r, workspace=ws)
        print(f"rank_method={method} r={r} sort of 2^26 keys (incl. clone) {timed(run, 2):.3f} ms", flush=True)
lsd.set_rank_method(-1)
