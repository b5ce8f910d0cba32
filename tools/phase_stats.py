#!/usr/bin/env python3
"""Where a rank-and-scatter tile spends its time (diagnostic build: make -C lsdradixsort_amd/csrc stats).
Run with LSDSORT_LIB=lsdradixsort_amd/liblsdsort_stats.so on the GPU box."""
import argparse, ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lsdradixsort_amd as lsd
from bench import mt19937_keys
ap = argparse.ArgumentParser()
ap.add_argument("--cfgs", type=int, nargs="*", default=[0])
ap.add_argument("--chunk", type=int, nargs="*", default=[0, 16])
ap.add_argument("--algos", type=int, nargs="*", default=[0, 1])
ap.add_argument("--mask", type=lambda x: int(x, 0), default=0xFFFFFFFF)
ap.add_argument("--radix", type=int, default=8)
ap.add_argument("--heavy", type=int, default=0, help="percent of the keys set to one value (0x80000001)")
ap.add_argument("--sorted", action="store_true", help="sorted input")
a = ap.parse_args()
L = ctypes.CDLL(lsd.LIB_PATH)
n = 1 << 28
host = mt19937_keys(n, 0) & np.uint32(a.mask)
if a.heavy:
    host = np.where((host >> np.uint32(7)) % np.uint32(100) < np.uint32(a.heavy), np.uint32(0x80000001), host).astype(np.uint32)
if a.sorted:
    host = np.sort(host)
master = lsd.to_device(host)
stats = torch.zeros((1 << 17) * 16, dtype=torch.int64, device="cuda")
L.lsdsort_debug_set_stats.argtypes = [ctypes.c_void_p]
L.lsdsort_debug_set_stats(stats.data_ptr())
names = ["ticket", "keyload", "rank", "scan+pub", "ldswrite", "lookback", "readback+store"]
for cfg in a.cfgs:
    lsd.set_tile_config(a.radix, cfg)
    for algo in a.algos:
        for C in a.chunk:
            lsd.set_xcd_chunk(C)
            ws = lsd.alloc_workspace(n, a.radix, False, algo)
            k = master.clone(); lsd.GPULSDRadixSort(k, a.radix, algorithm=algo, workspace=ws); torch.cuda.synchronize()
            stats.zero_(); torch.cuda.synchronize()
            k = master.clone()
            tm = lsd.GPULSDRadixSortTimed(k, a.radix, algorithm=algo, workspace=ws)
            assert tm["tiles"] <= (1 << 17)
            rec = stats.cpu().numpy().astype(np.float64)[: tm["tiles"] * 16].reshape(-1, 16)   # last pass's records
            s = np.zeros(16); s[14] = rec[:, 7].mean(); s[13] = rec[:, 8].mean(); tiles = 1
            per = rec[:, :7].mean(axis=0) / 100.0   # us per tile (100 MHz clock)
            rr = rec[rec[:, 9] > 0]
            st = (rr[:, 9] - rr[:, 9].min()) / 100.0; en = st + rr[:, :7].sum(axis=1) / 100.0
            print("   concurrency at 25/50/75%:", [int(np.sum((st <= q) & (en > q))) for q in (en.max() * 0.25, en.max() * 0.5, en.max() * 0.75)], flush=True)
            np.save(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out', f'rec_cfg{cfg}_algo{algo}.npy'), rec)
            print(f"cfg={cfg} tile={tm['tile_keys']} algo={algo} C={C:2d} scatter/pass={np.mean(tm['scatter_ms']):.3f} ms | " +
                  " ".join(f"{nm}={v:5.2f}" for nm, v in zip(names, per)) + f" | sum={per.sum():5.2f} us/tile refills/tile={s[14]/tiles:.2f} emptypolls/tile={s[13]/tiles:.2f}", flush=True)
