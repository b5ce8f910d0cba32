#!/usr/bin/env python3
"""Offline look at gpurun_out/rec_cfg*_algo0.npy (tools/phase_stats.py, stats build): how far each tile
walked, where the prefix frontier was, and how stale the status words it read were."""
import sys, numpy as np
rec = np.load(sys.argv[1]); regions = int(sys.argv[2]) if len(sys.argv) > 2 else 8
n = rec.shape[0]; per = n // regions
t0 = rec[:, 9][rec[:, 9] > 0].min()
for reg in (0, regions // 2):
    r = rec[reg * per:(reg + 1) * per]
    us = lambda c: (r[:, c] - t0) / 100.0
    start, walk0, met, pub = us(9), us(14), us(12), us(11)
    depth, refills, empty, metpos = r[:, 10], r[:, 7], r[:, 8], r[:, 13].astype(int)
    ks = np.arange(per // 4, 3 * per // 4)
    first = us(15)
    rf = refills[ks] > 0
    print(f"   wait for the prefetched step {np.mean(first[ks]-walk0[ks]):.2f} us; tiles needing refills {rf.mean():.2f}; per refill {np.sum((met[ks]-first[ks])[rf])/max(1,np.sum(refills[ks][rf])):.2f} us; rows from the first step {np.mean(depth[ks]-0):.1f} total")
    print(f"region {reg}: rows walked {depth[ks].mean():.1f}  refills {refills[ks].mean():.2f}  empty polls {empty[ks].mean():.2f}  walk {np.mean(met[ks]-walk0[ks]):.2f} us")
    # frontier when the walk started / ended, by the publishers' own clocks
    fs, fe, lag = [], [], []
    for k in ks:
        j = k - 1
        while j >= 0 and pub[j] > walk0[k]: j -= 1
        fs.append(k - j)
        j = k - 1
        while j >= 0 and pub[j] > met[k]: j -= 1
        fe.append(k - j)
        lag.append(met[k] - pub[metpos[k]])
    print(f"   frontier distance at walk start {np.mean(fs):.1f}, at walk end {np.mean(fe):.1f}; rows walked beyond the end frontier {np.mean(depth[ks]-np.array(fe)):.1f}")
    print(f"   age of the prefix word when it was met: mean {np.mean(lag):.2f} us, p10 {np.percentile(lag,10):.2f}, min {np.min(lag):.2f}")
