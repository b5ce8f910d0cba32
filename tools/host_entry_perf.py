#!/usr/bin/env python3
"""The literal sort(uint32_t* keys, size_t n) on a HOST array (lsdsort_u32: chunked H2D with stage 1 behind it, the
passes, D2H -- the window of the reference's .cu:966-1005), PCIe included, next to what the two transfers alone cost
(hipMemcpy of the same pageable array each way, as torch issues it).  Never bench.py's `value`."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import lsdradixsort_amd as lsd
from bench import mt19937_keys

for log2n in (20, 24, 28):
    n = 1 << log2n
    keys = mt19937_keys(n, 0)
    expect_first, expect_last = int(keys.min()), int(keys.max())
    times = []
    for i in range(4):
        k = keys.copy()
        t0 = time.perf_counter()
        lsd.sort(k)
        times.append(time.perf_counter() - t0)
    assert int(k[0]) == expect_first and int(k[-1]) == expect_last and bool(np.all(k[1:] >= k[:-1]))
    first, t = times[0], min(times[1:])
    # the transfers alone, same pageable array
    d = torch.empty(n, dtype=torch.int32, device="cuda")
    src = torch.from_numpy(keys.view(np.int32))
    h2d, d2h = [], []
    for i in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter(); d.copy_(src); torch.cuda.synchronize(); h2d.append(time.perf_counter() - t0)
        torch.cuda.synchronize(); t0 = time.perf_counter(); src.copy_(d); torch.cuda.synchronize(); d2h.append(time.perf_counter() - t0)
    ws = lsd.alloc_workspace(n, 8)
    dd = d.clone(); torch.cuda.synchronize(); t0 = time.perf_counter(); lsd.GPULSDRadixSort(dd, 8, workspace=ws); torch.cuda.synchronize()
    sort_ms = (time.perf_counter() - t0) * 1e3
    print(f"n=2^{log2n}: host entry {t * 1e3:8.2f} ms (first call, building the cache: {first * 1e3:.2f})  {n / t / 1e6:8.1f} Mkeys/s | "
          f"H2D alone {min(h2d) * 1e3:.2f} ms, D2H alone {min(d2h) * 1e3:.2f} ms, device sort {sort_ms:.2f} ms -> "
          f"{t * 1e3 / (min(h2d) * 1e3 + min(d2h) * 1e3 + sort_ms):.3f} of H2D + sort + D2H, "
          f"{t * 1e3 / (min(h2d) * 1e3 + min(d2h) * 1e3):.3f} of the two transfers", flush=True)
