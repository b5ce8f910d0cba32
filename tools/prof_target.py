#!/usr/bin/env python3
"""Workload for rocprofv3: a few device-resident sorts of the bench input (GPU box)."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lsdradixsort_amd as lsd
from bench import mt19937_keys

ap = argparse.ArgumentParser()
ap.add_argument("--log2-keys", type=int, default=28)
ap.add_argument("--radix-bits", type=int, default=8)
ap.add_argument("--algorithm", default="onesweep")
ap.add_argument("--pairs", action="store_true")
ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--tile-config", type=int, default=-1)
ap.add_argument("--mask", type=lambda x: int(x, 0), default=0xFFFFFFFF)
ap.add_argument("--rank", type=int, default=-1)
a = ap.parse_args()
n = 1 << a.log2_keys
algo = 0 if a.algorithm == "onesweep" else 1
if a.tile_config >= 0:
    lsd.set_tile_config(a.radix_bits, a.tile_config)
import numpy as np
lsd.set_rank_method(a.rank)
master = lsd.to_device(mt19937_keys(n, 0) & np.uint32(a.mask))
vals = torch.arange(n, dtype=torch.int32, device="cuda") if a.pairs else None
ws = lsd.alloc_workspace(n, a.radix_bits, a.pairs, algo)
for i in range(a.steps):
    k = master.clone()
    v = vals.clone() if a.pairs else None
    lsd.GPULSDRadixSort(k, a.radix_bits, d_vals=v, algorithm=algo, workspace=ws)
torch.cuda.synchronize()
print("done", n, a.radix_bits, a.algorithm)
