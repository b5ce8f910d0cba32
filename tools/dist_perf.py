#!/usr/bin/env python3
"""Sort time by key distribution (GPU box): the reference only ever tests uniform keys."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lsdradixsort_amd as lsd
from bench import mt19937_keys

import argparse
ap = argparse.ArgumentParser()
ap.add_argument("--radix", type=int, nargs="*", default=[8, 4])
ap.add_argument("--cfg", type=int, default=-1, help="tile configuration for every radix listed (-1: library default)")
ap.add_argument("--only", nargs="*", default=None, help="case names to run")
a = ap.parse_args()
n = 1 << 28
base = mt19937_keys(n, 0)
cases = {
    "uniform": lambda: base,
    "sorted": lambda: np.sort(base),
    "reverse": lambda: np.sort(base)[::-1].copy(),
    "all_equal": lambda: np.full(n, 0x12345678, dtype=np.uint32),
    "low8_only": lambda: base & np.uint32(0xFF),
    "high8_only": lambda: base & np.uint32(0xFF000000),
    "16_values_per_digit": lambda: base & np.uint32(0x0F0F0F0F),
    "small_range_2^20": lambda: base & np.uint32(0xFFFFF),
    # heavy hitters: part of every wave row shares one digit, the rest is random (same-address LDS atomics of SOME lanes)
    "half_zero_keys": lambda: np.where((base >> np.uint32(13)) & np.uint32(1), base, np.uint32(0)).astype(np.uint32),
    "90pct_one_value": lambda: np.where((base % np.uint32(10)) != 0, np.uint32(0x80000001), base).astype(np.uint32),
    "two_values": lambda: np.where(base & np.uint32(1 << 17), np.uint32(0x11111111), np.uint32(0xEEEEEEEE)).astype(np.uint32),
    "four_values_per_digit": lambda: base & np.uint32(0x03030303),
    # shapes the hybrid form meets: a shared key prefix, dead low bits (one digit of the local stage constant), few low values
    "below_2^31": lambda: base >> np.uint32(1),
    "below_2^29": lambda: base >> np.uint32(3),
    "multiples_of_512": lambda: base & np.uint32(0xFFFFFE00),
    "low_byte_of_4_values": lambda: base & np.uint32(0xFFFFFF03),
}
if a.only:
    cases = {k: v for k, v in cases.items() if k in a.only}
for r in a.radix:
    if a.cfg >= 0:
        lsd.set_tile_config(r, a.cfg)
    ws = lsd.alloc_workspace(n, r)
    for name, make in cases.items():
        d0 = lsd.to_device(make())
        times = []
        for i in range(4):
            d = d0.clone()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            lsd.GPULSDRadixSort(d, r, workspace=ws)
            e1.record()
            torch.cuda.synchronize()
            if i:
                times.append(e0.elapsed_time(e1))
        tm = lsd.GPULSDRadixSortTimed(d0.clone(), r, workspace=ws)
        stages = f"hist {tm['histogram_ms']:.3f} passes " + " ".join(f"{x:.3f}" for x in tm["scatter_ms"])
        u = d.to(torch.int64) & 0xFFFFFFFF
        ok = bool((u[1:] >= u[:-1]).all())
        lsd.lib().lsdsort_check_device(ws.data_ptr(), torch.cuda.current_stream().cuda_stream)
        print(f"r={r} cfg={a.cfg} {name:22s} {np.median(times):8.3f} ms  {n / np.median(times) / 1e6:8.1f} Gkeys/s  sorted={ok}  | {stages}", flush=True)
        del d0, d, u
