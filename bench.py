#!/usr/bin/env python3
"""bench.py -- Mkeys/s of the device-resident uint32 LSD radix sort on MI355X.

Contract (one JSON line on rank 0):  python bench.py --gpus N --steps K --warmup W
  * a "step" = one complete sort (clear + digit histograms + scan + 32/r rank-and-scatter
    passes) of 2^28 uniform uint32 keys (1 GiB) per GPU -- BASELINE.json configs[2]
    ("1 GiB uniform-random uint32 keys, 8-bit radix"), the configuration the metric and the
    70 % roofline target are quoted on.  --radix-bits 4 gives configs[1]; --pairs configs[4].
  * input: raw std::mt19937(seed=rank) outputs, resident in HBM before
    the timed region; every step sorts its own fresh copy, so no restore copy is timed.
  * timed region: barrier + torch.cuda.synchronize() on both sides of exactly K steps, MAX over
    ranks.  N > 1 (launched by torch.distributed.run): weak scaling, MSB-bucket partition +
    RCCL all-to-all + local sort per step (lsdradixsort_amd/dist.py); value = N * keys / time.
  * roofline: the rank-and-scatter kernel, algorithmic bytes per launch (8 B/key: one read, one
    write) / its mean launch duration measured with hipEvents inside the library
    (lsdsort_u32_device_timed, same stream), against 8 TB/s HBM peak.
  * cpu_baseline: the reference's CPU std::sort path (LSDRadixSort.cu:97) on one host thread,
    on a bounded sample of the same keys; rank 0, N=1 only.  The oracle is used here and only here.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, 6.29 TB/s measured copy)
POOL_LIMIT_BYTES = 96 << 30    # fresh input copies kept resident per GPU (of 288 GB)


def mt19937_keys(n: int, seed: int):
    """Raw std::mt19937(seed) outputs via numpy's MT19937 with the legacy (init_genrand) seeding;
    tests/test_bench_inputs.py pins it to the C++ generator."""
    import numpy as np

    bg = np.random.MT19937()
    bg._legacy_seeding(seed)
    out = np.empty(n, dtype=np.uint32)
    chunk = 1 << 24
    for i in range(0, n, chunk):
        m = min(chunk, n - i)
        out[i:i + m] = bg.random_raw(m).astype(np.uint32)
    return out


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--radix-bits", type=int, default=8, choices=[1, 2, 4, 8])
    ap.add_argument("--log2-keys", type=int, default=28, help="keys per GPU = 2^this (default 2^28 = 1 GiB)")
    ap.add_argument("--algorithm", choices=["onesweep", "staged"], default="onesweep")
    ap.add_argument("--pairs", action="store_true", help="key + uint32 payload (BASELINE configs[4])")
    ap.add_argument("--tile-config", type=int, default=-1)
    ap.add_argument("--exercise-exchange", action="store_true",
                    help="one GPU only: run the sharded path (partition, RCCL count exchange and all-to-all, local sort) with a "
                         "process group of one rank -- a rehearsal of the N > 1 code on a one-GPU box, not a benchmark")
    ap.add_argument("--partition", choices=["msb", "splitters"], default="msb",
                    help="N > 1: how keys are assigned to ranks (msb: top log2 N bits; splitters: sampled, for skewed keys)")
    ap.add_argument("--rank-method", type=int, default=-1, help="-1 library default, 0 peer-mask forms, 2 returning LDS add (tuning aid)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-log2", type=int, default=26)
    ap.add_argument("--no-extra", action="store_true", help="skip the secondary configs reported under 'extra'")
    return ap.parse_args()


def timed_steps(run_step, pools, steps, warmup, sync):
    """Warm up, then time exactly `steps` steps between two sync points.  `pools` yields fresh
    inputs; when steps exceed the resident pool the timed region is split into rounds and the
    restore copies between rounds are not timed."""
    for i in range(warmup):
        run_step(pools.fresh())
    elapsed = 0.0
    done = 0
    while done < steps:
        batch = min(steps - done, pools.capacity)
        inputs = [pools.fresh() for _ in range(batch)]
        sync()
        t0 = time.perf_counter()
        for x in inputs:
            run_step(x)
        sync()
        elapsed += time.perf_counter() - t0
        done += batch
    return elapsed


class InputPool:
    """K resident copies of the same input array; each is sorted once, then refilled off the clock."""

    def __init__(self, master, master_vals, capacity):
        self.master, self.master_vals = master, master_vals
        self.capacity = capacity
        self.bufs = [(master.clone(), master_vals.clone() if master_vals is not None else None)
                     for _ in range(capacity)]
        self.next = 0
        self.dirty = [False] * capacity

    def fresh(self):
        i = self.next
        self.next = (self.next + 1) % self.capacity
        k, v = self.bufs[i]
        if self.dirty[i]:
            k.copy_(self.master)
            if v is not None:
                v.copy_(self.master_vals)
        self.dirty[i] = True
        return k, v


def main():
    args = parse_args()
    # multi-process GPU work on this image needs dmabuf IPC (RCCL fails with hipIpcGetMemHandle otherwise)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import numpy as np
    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    torch.cuda.set_device(local_rank)
    import lsdradixsort_amd as lsd

    assert lsd.lib().lsdsort_device_count() >= 1, "liblsdsort.so sees no gfx950 device (no CPU fallback)"

    distributed = world > 1 or args.exercise_exchange
    if distributed:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:
            import socket

            with socket.socket() as sock:
                sock.bind(("127.0.0.1", 0))
                os.environ.setdefault("MASTER_PORT", str(sock.getsockname()[1]))
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    r = args.radix_bits
    algo = lsd.LSDSORT_ALGO_ONESWEEP if args.algorithm == "onesweep" else lsd.LSDSORT_ALGO_STAGED
    if args.tile_config >= 0:
        lsd.set_tile_config(r, args.tile_config)
    if args.rank_method >= 0:
        lsd.set_rank_method(args.rank_method)
    n = 1 << args.log2_keys
    passes = 32 // r

    host_keys = mt19937_keys(n, rank)
    master = lsd.to_device(host_keys)
    master_vals = torch.arange(n, dtype=torch.int32, device="cuda") if args.pairs else None
    bytes_per_copy = 4 * n * (2 if args.pairs else 1)
    capacity = max(1, min(args.steps + args.warmup, POOL_LIMIT_BYTES // bytes_per_copy))
    pool = InputPool(master, master_vals, capacity)
    ws = lsd.alloc_workspace(n, r, args.pairs, algo)

    def sync():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    if not distributed:
        def run_step(kv):
            lsd.GPULSDRadixSort(kv[0], r, d_vals=kv[1], algorithm=algo, workspace=ws)
    else:
        from lsdradixsort_amd.dist import HipBackend, distributed_sort

        backend = HipBackend(r)

        def run_step(kv):
            distributed_sort(kv[0], backend=backend, exchange_always=args.exercise_exchange, partition=args.partition)

    elapsed = timed_steps(run_step, pool, args.steps, args.warmup, sync)
    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # correctness guard for the sharded path (untimed, collective): every rank's slice is sorted, lies
        # in its own MSB bucket, and the slices add up to every key
        res = distributed_sort(pool.fresh()[0], backend=backend, exchange_always=args.exercise_exchange, partition=args.partition)
        u = res.keys.to(torch.int64) & 0xFFFFFFFF
        bits = world.bit_length() - 1
        ok = bool((u[1:] >= u[:-1]).all()) if u.numel() > 1 else True
        if u.numel() and bits and args.partition == "msb":
            ok = ok and int(u[0].item()) >> (32 - bits) == rank and int(u[-1].item()) >> (32 - bits) == rank
        cnt = torch.tensor([u.numel(), 0 if ok else 1], dtype=torch.int64, device="cuda")
        dist.all_reduce(cnt)
        assert int(cnt[0].item()) == n * world and int(cnt[1].item()) == 0, "sharded sort failed its check"
        del res, u
    check_status = lsd.lib().lsdsort_check_device(ws.data_ptr(), torch.cuda.current_stream().cuda_stream)
    assert check_status == 0, f"device fault word set ({check_status})"

    ms_per_step = elapsed / args.steps * 1e3
    total_keys = n * world
    mkeys = total_keys / (elapsed / args.steps) / 1e6

    # correctness guard on the last sorted buffer of this rank (single GPU): sortedness
    if not distributed:
        k = pool.bufs[(pool.next - 1) % pool.capacity][0]
        u = k.to(torch.int64) & 0xFFFFFFFF
        assert bool((u[1:] >= u[:-1]).all()), "bench output is not sorted"
        del u

    # ---- roofline of the dominant kernel (rank-and-scatter), measured live with hipEvents ----
    roofline = None
    stage_ms = None
    if rank == 0:
        scat, hist, scan, clear, totals = [], [], [], [], []
        sort_tile_keys = None
        for _ in range(5):
            kv = pool.fresh()
            tm = lsd.GPULSDRadixSortTimed(kv[0], r, d_vals=kv[1], algorithm=algo, workspace=ws)
            sort_tile_keys = tm["tile_keys"]
            scat += tm["scatter_ms"]
            hist.append(tm["histogram_ms"])
            scan.append(tm["scan_ms"])
            clear.append(tm["clear_ms"])
            totals.append(tm["total_ms"])
        per_key = 16 if args.pairs else 8          # one read + one write of the key (and payload) per pass
        scat_ms = float(np.mean(scat))
        achieved = per_key * n / (scat_ms * 1e-3) / 1e9
        traffic = None
        prof = os.path.join(ROOT, "profiles", "pmc_summary.json")
        if os.path.exists(prof):
            try:
                pj = json.load(open(prof))
                key = f"rank_scatter_r{r}{'_pairs' if args.pairs else ''}"
                if pj.get(key, {}).get("n") == n:
                    traffic = pj[key]["hbm_bytes_per_launch"]
            except Exception:
                traffic = None
        roofline = {"bound": "hbm", "kernel": "rank_scatter_kernel", "achieved": round(achieved, 1),
                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                    "traffic": traffic, "algorithmic_bytes_per_launch": per_key * n,
                    "launch_ms": round(scat_ms, 4)}
        sort_bytes = (4 + 16 * passes) * n if args.pairs else 4 * (2 * passes + 1) * n
        stage_ms = {"clear": round(float(np.mean(clear)), 4), "histogram": round(float(np.mean(hist)), 4),
                    "scan": round(float(np.mean(scan)), 4), "scatter_per_pass": round(scat_ms, 4),
                    "total_event": round(float(np.mean(totals)), 4),
                    "sort_algorithmic_gbs": round(sort_bytes / (ms_per_step * 1e-3) / 1e9 / (world if distributed else 1), 1),
                    "sort_roofline_frac": round(sort_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if not distributed else None}

    # ---- secondary configs on the same box (N=1 only): configs[1] (r=4) and configs[4] (pairs) ----
    extra = {}
    if rank == 0 and not distributed and not args.no_extra and not args.pairs and r == 8 and args.log2_keys == 28:
        def quick(rb, pairs, nn):
            kk = master[:nn].clone()
            vv = torch.arange(nn, dtype=torch.int32, device="cuda") if pairs else None
            w2 = lsd.alloc_workspace(nn, rb, pairs, algo)
            times = []
            for i in range(6):
                kk.copy_(master[:nn])
                if vv is not None:
                    vv.copy_(torch.arange(nn, dtype=torch.int32, device="cuda"))
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                lsd.GPULSDRadixSort(kk, rb, d_vals=vv, algorithm=algo, workspace=w2)
                torch.cuda.synchronize()
                if i:
                    times.append(time.perf_counter() - t0)
            return nn / float(np.median(times)) / 1e6
        extra["r4_256M_keys_mkeys_s"] = round(quick(4, False, n), 1)
        extra["pairs_r8_128M_pairs_mpairs_s"] = round(quick(8, True, n // 2), 1)

        # Stage micro-benchmarks: the counterparts of the reference's TestBuildHistogram
        # (.cu:704) and TestGPUPrefixSum (.cu:304) harnesses, through the stage-level C-ABI entries.
        def time_stage(fn, reps=5):
            ts = []
            for i in range(reps + 1):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                fn()
                e1.record()
                torch.cuda.synchronize()
                if i:
                    ts.append(e0.elapsed_time(e1))
            return float(np.median(ts))
        hist_ms = time_stage(lambda: lsd.BuildHistograms(master, 8, 0))
        h = lsd.BuildHistograms(master, 8, 0)
        offs_ms = time_stage(lambda: lsd.BuildOffsets(h, 8))
        dh_ms = time_stage(lambda: lsd.DigitHistograms(master, 8))
        extra["stage_bench"] = {
            "tile_histograms_r8": {"ms": round(hist_ms, 4), "read_gbs": round(4 * n / hist_ms / 1e6, 1),
                                   "replaces": "BuildHistogramsKernel .cu:660-702 (TestBuildHistogram .cu:704)"},
            "tile_offsets_r8": {"ms": round(offs_ms, 4), "tiles": int(h.shape[0]),
                                "table_gbs": round(3 * 4 * h.numel() / offs_ms / 1e6, 1),
                                "replaces": "offset construction .cu:862-895 (TestGPUPrefixSum .cu:304)"},
            "digit_histograms_all_passes_r8": {"ms": round(dh_ms, 4), "read_gbs": round(4 * n / dh_ms / 1e6, 1)},
        }
        del h

    # ---- CPU baseline: the reference's std::sort path on the host, one thread ----
    cpu_baseline = None
    if rank == 0 and not distributed and not args.no_cpu_baseline:
        import oracle     # test infrastructure; used only for this reported baseline

        oracle.lib()
        m = min(n, 1 << args.cpu_sample_log2)
        sample = host_keys[:m]
        t_std = oracle.time_std_sort(sample)
        t_lsd = oracle.time_lsd_sort(sample, 8)
        cpu_baseline = {"value": round(m / (t_std * 1e-3) / 1e6, 2), "unit": "Mkeys/s", "cores": 1, "kind": "port",
                        "sample": f"std::sort (LSDRadixSort.cu:97) of the first 2^{m.bit_length() - 1} keys of the "
                                  f"workload, 1 thread, {t_std / 1e3:.1f} s; O(n log n), so the full 2^{args.log2_keys} "
                                  f"would be slower per key",
                        "lsd_r8_mkeys_s": round(m / (t_lsd * 1e-3) / 1e6, 2),
                        "lsd_r8_note": "restated reference CPU LSD (LSDRadixSort.cu:25-69), r=8, same sample",
                        "host_cpus": os.cpu_count()}

    if rank == 0:
        workload = (f"2^{args.log2_keys} uniform uint32 {'key+payload pairs' if args.pairs else 'keys'} per GPU "
                    f"(mt19937 seed=rank), {r}-bit radix, {passes} passes, {args.algorithm}"
                    + (", MSB-bucket RCCL all-to-all + local sort" if distributed else ", device-resident"))
        line = {
            "metric": "Mkeys/s sorting uniform uint32, 1 GiB, 1/2/4/8 MI355X; % HBM roofline",
            "value": round(mkeys, 1), "unit": "Mkeys/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u32", "data": "synthetic",
            "config": {"workload": workload, "keys_per_gpu": n, "radix_bits": r, "algorithm": args.algorithm,
                       "pairs": bool(args.pairs), "tile_keys": sort_tile_keys},
            "roofline": roofline, "cpu_baseline": cpu_baseline, "stages_ms": stage_ms, "extra": extra,
        }
        print(json.dumps(line), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
