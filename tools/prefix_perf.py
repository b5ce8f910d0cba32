#!/usr/bin/env python3
"""What the key-prefix hint buys a shard's local sort (GPU box): 2^log2n keys that share their top `prefix` bits, sorted with
lsdsort_u32_device (hybrid form tried and refused: buckets 2^prefix times too large) and lsdsort_u32_device_prefixed."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import argparse
import torch
import lsdradixsort_amd as lsd
ap = argparse.ArgumentParser()
ap.add_argument("--log2n", type=int, nargs="*", default=[27, 28])
ap.add_argument("--prefix", type=int, nargs="*", default=[1, 3, 4])
ap.add_argument("--radix", type=int, default=8)
a = ap.parse_args()
L = lsd.lib()
s = torch.cuda.current_stream().cuda_stream
for lg in a.log2n:
    n = 1 << lg
    base = torch.randint(-(1 << 31), (1 << 31) - 1, (n,), dtype=torch.int32, device="cuda")
    ws = lsd.alloc_workspace(n, a.radix)
    for t in a.prefix:
        u = ((base.to(torch.int64) & 0xFFFFFFFF) >> t) | ((0xA5 >> (8 - t)) << (32 - t))
        keys = ((u + (1 << 31)) % (1 << 32) - (1 << 31)).to(torch.int32)
        row = []
        for hint in (0, t):
            bufs = [keys.clone() for _ in range(12)]
            for b in bufs[:2]:
                L.lsdsort_u32_device_prefixed(b.data_ptr(), ws.data_ptr(), ws.numel(), n, a.radix, hint, s)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for b in bufs[2:]:
                L.lsdsort_u32_device_prefixed(b.data_ptr(), ws.data_ptr(), ws.numel(), n, a.radix, hint, s)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 10
            row.append(f"hint {hint}: {ms:.3f} ms ({n / ms / 1e6:.1f} Gkeys/s, form {lsd.workspace_form(ws)})")
        print(f"n=2^{lg} prefix {t} r={a.radix} | " + " | ".join(row), flush=True)
