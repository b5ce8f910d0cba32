#!/usr/bin/env python3
"""The literal sort(uint32_t* keys, size_t n) on a HOST array (lsdsort_u32: alloc + H2D + sort + D2H, the
window of the reference's .cu:966-1005), PCIe included.  Never bench.py's `value`."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import lsdradixsort_amd as lsd
from bench import mt19937_keys

for log2n in (20, 24, 28):
    n = 1 << log2n
    keys = mt19937_keys(n, 0)
    expect_first, expect_last = int(keys.min()), int(keys.max())
    times = []
    for i in range(3):
        k = keys.copy()
        t0 = time.perf_counter()
        lsd.sort(k)
        times.append(time.perf_counter() - t0)
    assert int(k[0]) == expect_first and int(k[-1]) == expect_last and bool(np.all(k[1:] >= k[:-1]))
    t = min(times)
    print(f"n=2^{log2n}: {t * 1e3:8.2f} ms  {n / t / 1e6:8.1f} Mkeys/s  ({8 * n / t / 1e9:.1f} GB/s of PCIe traffic both ways)", flush=True)
