import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lsdradixsort_amd as lsd
from bench import mt19937_keys
L = ctypes.CDLL(lsd.LIB_PATH)
n = 1 << 28
master = lsd.to_device(mt19937_keys(n, 0))
lsd.set_rank_method(2)
for cfg in (0, 3, 8):
    lsd.set_tile_config(8, cfg)
    for C in (0, 1, 2, 4, 8, 16, 32):
        L.lsdsort_debug_set_xcd_chunk(C)
        ws = lsd.alloc_workspace(n, 8, False, 1)
        sc = []
        for i in range(4):
            k = master.clone()
            tm = lsd.GPULSDRadixSortTimed(k, 8, algorithm=1, workspace=ws)
            if i: sc += tm["scatter_ms"]
        u = k.to(torch.int64) & 0xFFFFFFFF
        print(f"cfg={cfg} tile={tm['tile_keys']} staged xcd_chunk={C:2d} scatter/pass={np.mean(sc):.3f} ms sorted={bool((u[1:]>=u[:-1]).all())}", flush=True)
