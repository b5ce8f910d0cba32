#!/usr/bin/env python3
"""64-bit keys / payloads (lsdsort_u64_device, lsdsort_records_device) on the GPU box: Gkeys/s at 2^27 items."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lsdradixsort_amd as lsd

n = 1 << 27
g = torch.Generator(device="cuda"); g.manual_seed(1)
k64 = torch.randint(-(1 << 63), (1 << 63) - 1, (n,), dtype=torch.int64, device="cuda", generator=g)
k32 = torch.randint(-(1 << 31), (1 << 31) - 1, (n,), dtype=torch.int32, device="cuda", generator=g)
v64 = torch.arange(n, dtype=torch.int64, device="cuda")
v32 = torch.arange(n, dtype=torch.int32, device="cuda")
for name, k, v in (("uint64 keys", k64, None), ("uint64 keys + 64-bit payloads", k64, v64), ("uint64 keys + 32-bit payloads", k64, v32),
                   ("uint32 keys + 64-bit payloads", k32, v64)):
    kb = 64 if k.dtype == torch.int64 else 32
    vb = 0 if v is None else (64 if v.dtype == torch.int64 else 32)
    ws = torch.empty(int(lsd.lib().lsdsort_wide_workspace_bytes(n, 8, kb, vb)), dtype=torch.uint8, device="cuda")
    ts = []
    for i in range(4):
        kk, vv = k.clone(), (v.clone() if v is not None else None)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        lsd.GPUSortWide(kk, vv, workspace=ws)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    t = min(ts[1:])
    print(f"{name:32s} n=2^27: {t * 1e3:7.2f} ms  {n / t / 1e9:6.2f} Gitems/s  workspace {ws.numel() / n:.1f} B/item", flush=True)
