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
    ranks.  N > 1: one rank per GPU -- either launched by torch.distributed.run (RANK/WORLD_SIZE in
    the environment) or, when started plainly as `python bench.py --gpus N`, by this script itself: a
    parent that never touches a GPU spawns N fresh children and relays rank 0's JSON line.  Workload
    for N > 1 = BASELINE.json configs[3]: 2^30 keys in total (2^(30 - log2 N) per GPU, "strong"
    scaling), MSB-bucket partition + RCCL all-to-all + local sort per step (lsdradixsort_amd/dist.py);
    --total-log2-keys 28 gives the metric's "1 GiB at 1/2/4/8"; value = total keys / time.
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


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--radix-bits", type=int, default=8, choices=[1, 2, 4, 8])
    ap.add_argument("--log2-keys", type=int, default=None,
                    help="keys per GPU = 2^this (weak scaling when given with N > 1); default: 28 at N = 1, "
                         "--total-log2-keys minus log2 N otherwise")
    ap.add_argument("--total-log2-keys", type=int, default=None,
                    help="N > 1: keys over all GPUs = 2^this (default 30 = BASELINE configs[3], 4 GiB; 28 = the metric's 1 GiB)")
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
    ap.add_argument("--cpu-sample-log2", type=int, default=28,
                    help="keys of the workload the CPU baseline sorts (default: all 2^28 of config 3, about 16 s of std::sort on one core)")
    ap.add_argument("--no-extra", action="store_true", help="skip the secondary configs reported under 'extra'")
    ap.add_argument("--no-hybrid", action="store_true",
                    help="always every digit through global memory (lsdsort_set_hybrid(0)); default: the library decides on the device "
                         "whether the global passes on bits 16-31 + the LDS-resident local stage can run")
    ap.add_argument("--sub-buckets", type=int, default=1, choices=[1, 2, 4],
                    help="N > 1: sub-bucket pipelining of the sharded step (lsdsort_comm_set_sub_buckets)")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="do not run the two rocprofv3 --pmc passes that measure roofline.traffic in this very run (N = 1 only; "
                         "the committed profiles/pmc_summary.json figure is reported instead, labelled as such)")
    args = ap.parse_args(argv)
    if args.gpus not in (1, 2, 4, 8):
        ap.error("--gpus must be 1, 2, 4 or 8 (MSB buckets: one rank per power-of-two share of the key space)")
    return args


def keys_per_gpu_log2(args, world):
    """(log2 keys per GPU, scaling label).  N = 1: 2^28 (configs[2]).  N > 1: 2^30 in total unless told otherwise."""
    if args.log2_keys is not None:
        return args.log2_keys, "weak"
    if world == 1:
        return (args.total_log2_keys if args.total_log2_keys is not None else 28), "weak"
    total = args.total_log2_keys if args.total_log2_keys is not None else 30
    return total - (world.bit_length() - 1), "strong"


def visible_gpus():
    """GPUs this process could use, without initialising any (torch.cuda.device_count() does not, on this image)."""
    import torch

    return int(torch.cuda.device_count())


def spawn_ranks(n_ranks, argv, device_count=None, child_cmd=None, timeout=None, out=sys.stdout):
    """`python bench.py --gpus N` started plainly: be the launcher.  This parent process never touches a GPU; it
    starts N fresh children (never an exec of a process that has initialised one) with RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_ADDR / MASTER_PORT set, relays rank 0's stdout (the JSON line), and returns non-zero
    as soon as any rank fails, stopping the others by PID."""
    import socket
    import subprocess

    have = visible_gpus() if device_count is None else device_count
    if have < n_ranks:
        print(f"bench.py: --gpus {n_ranks} needs {n_ranks} visible GPUs, this machine shows {have}", file=sys.stderr)
        return 2
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = list(child_cmd) if child_cmd is not None else [sys.executable, os.path.abspath(__file__)] + list(argv)
    import threading

    procs = []
    for r in range(n_ranks):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_ranks), LOCAL_WORLD_SIZE=str(n_ranks),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    # rank 0's stdout is drained WHILE the ranks run: read only at the end, a rank 0 that writes more than the pipe holds
    # (64 KiB: NCCL_DEBUG=INFO does) would block in write() and this loop would poll for ever
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    deadline = None if timeout is None else time.time() + timeout
    rc = 0
    pending = set(range(n_ranks))
    while pending and rc == 0:
        for r in sorted(pending):
            code = procs[r].poll()
            if code is not None:
                pending.discard(r)
                if code != 0:
                    print(f"bench.py: rank {r} exited with code {code}", file=sys.stderr)
                    rc = code if code > 0 else 1
        if deadline is not None and time.time() > deadline:
            print("bench.py: ranks timed out", file=sys.stderr)
            rc = 124
        if pending and rc == 0:
            time.sleep(0.05)
    for r in pending:                      # a rank failed or timed out: stop the others (our own children, by PID)
        procs[r].terminate()
    for r in pending:
        try:
            procs[r].wait(timeout=10)
        except subprocess.TimeoutExpired:
            procs[r].kill()
    reader.join(timeout=10)
    text = "".join(chunks)
    if rc == 0:
        out.write(text)
        out.flush()
    else:
        sys.stderr.write(text)
    return rc


def live_pmc_traffic(radix_bits, pairs, log2n, timeout=240):
    """HBM bytes per launch of the sort's kernels measured NOW, as MI355X_MICROARCH.md prescribes for gfx950: FETCH_SIZE and
    WRITE_SIZE from SEPARATE `rocprofv3 --pmc` passes (counters only, no tracing), each over a child process that runs the same sort
    on the same input (tools/prof_target.py; the program after `--` is python3 itself); both counters are in KiB; FETCH_SIZE reports
    half of a coalesced streaming read and is doubled, the doubling calibrated on the upfront histogram kernel's known 4*n read.
    Launches that returned at once (the kernels of the form of the sort that did NOT run) are left out.  Returns a dict by kernel
    class -- "rank_scatter", "stage1", "local" -- or None (with a reason on stderr) if rocprofv3 is missing or a pass fails."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile

    if shutil.which("rocprofv3") is None:
        print("bench.py: rocprofv3 not found; roofline.traffic falls back to profiles/pmc_summary.json", file=sys.stderr)
        return None
    n = 1 << log2n
    out = tempfile.mkdtemp(prefix="lsd_pmc_", dir="/tmp")
    target = [sys.executable, os.path.join(ROOT, "tools", "prof_target.py"), "--log2-keys", str(log2n), "--radix-bits", str(radix_bits),
              "--steps", "2"] + (["--pairs"] if pairs else [])
    env = dict(os.environ, TMPDIR="/tmp")
    classes = {"rank_scatter_kernel": "rank_scatter", "histograms_kernel": "stage1", "local_sort_kernel": "local"}
    got = {}
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(out, counter)
            p = subprocess.run(["rocprofv3", "--pmc", counter, "-d", d, "--output-format", "csv", "--"] + target, cwd="/tmp", env=env,
                               capture_output=True, text=True, timeout=timeout)
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if p.returncode != 0 or not files:
                print(f"bench.py: rocprofv3 --pmc {counter} failed (rc {p.returncode}); roofline.traffic falls back", file=sys.stderr)
                return None
            per = {}
            for row in csv.DictReader(open(files[0])):
                if row["Counter_Name"] != counter:
                    continue
                dur_us = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3
                for needle, cls in classes.items():
                    if needle in row["Kernel_Name"] and dur_us > 60.0:      # a launch that really ran (n >= 2^24)
                        per.setdefault(cls, []).append((float(row["Counter_Value"]) * 1024.0, dur_us))
            got[counter] = per
    except Exception as e:      # a profiler problem must not take the benchmark down
        print(f"bench.py: live PMC pass failed ({e}); roofline.traffic falls back", file=sys.stderr)
        return None
    finally:
        shutil.rmtree(out, ignore_errors=True)
    result = {}
    for cls in ("rank_scatter", "stage1", "local"):
        f, w = got["FETCH_SIZE"].get(cls), got["WRITE_SIZE"].get(cls)
        if not f or not w:
            continue
        fetch = sum(x for x, _ in f) / len(f)
        write = sum(x for x, _ in w) / len(w)
        us = (sum(t for _, t in f) / len(f) + sum(t for _, t in w) / len(w)) / 2.0
        result[cls] = {"bytes": int(2.0 * fetch + write), "fetch_bytes_corrected": int(2.0 * fetch), "write_bytes": int(write),
                       "launches": len(f), "kernel_us_under_profiler": round(us, 1)}
    if "rank_scatter" not in result:
        return None
    if "stage1" in result:
        result["calibration_ratio"] = round(result["stage1"]["fetch_bytes_corrected"] / (4.0 * n), 4)
    return result


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


def main(argv=None):
    argv = sys.argv[1:] if argv is None else list(argv)
    args = parse_args(argv)
    # multi-process GPU work on this image needs dmabuf IPC (RCCL fails with hipIpcGetMemHandle otherwise)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_ranks(args.gpus, argv, timeout=3000.0)   # plain `python bench.py --gpus N`: this process only launches
    import numpy as np
    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        args.gpus = world                            # under torch.distributed.run the environment decides
    if world not in (1, 2, 4, 8):
        raise SystemExit(f"bench.py: world size {world} is not 1, 2, 4 or 8")
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
    if args.no_hybrid:
        lsd.set_hybrid(False)
    log2_keys, scaling = keys_per_gpu_log2(args, world)
    n = 1 << log2_keys
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
        from lsdradixsort_amd.dist import HipBackend, ShardedSorter, distributed_sort

        # The product path: the C++ step behind the C-ABI (lsdsort_sharded_u32_device_ex), RCCL called from C++, with
        # either ownership rule (--partition msb | splitters).  It is built and proved once on a small array before
        # anything is timed; if ANY rank fails there (its own librccl cannot be loaded or refuses the communicator),
        # every rank switches to the torch.distributed driver over the same kernels and the JSON line says so -- a
        # number from the other path, not no number.
        exchange_path = "c++"
        backend, why = None, ""
        try:
            backend = ShardedSorter(r, partition=args.partition, sub_buckets=args.sub_buckets)
            probe = backend.sort(master[: min(n, 1 << 16)].clone())
            torch.cuda.synchronize()
            if backend.check_fault() != 0:
                raise RuntimeError("fault word set by the probe sort")
            del probe
        except Exception as exc:                      # noqa: BLE001 -- reported below, on every rank
            why = f"{type(exc).__name__}: {exc}"
        bad = torch.tensor([1 if why else 0], dtype=torch.int32, device="cuda")
        dist.all_reduce(bad, op=dist.ReduceOp.MAX)
        if int(bad.item()):
            print(f"bench.py[rank {rank}]: C++ RCCL step unavailable ({why or 'failed on another rank'}); "
                  f"using the torch.distributed driver", file=sys.stderr, flush=True)
            if backend is not None:
                backend.close()
            exchange_path = "torch"
            backend = HipBackend(r)

            def sharded(keys):
                return distributed_sort(keys, backend=backend, exchange_always=args.exercise_exchange, partition=args.partition)
        else:
            def sharded(keys):
                return backend.sort(keys)

        def run_step(kv):
            sharded(kv[0])

    elapsed = timed_steps(run_step, pool, args.steps, args.warmup, sync)
    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # correctness guard for the sharded path (untimed, collective): every rank's slice is sorted, lies
        # in its own MSB bucket, and the slices add up to every key
        res = sharded(pool.fresh()[0])
        u = res.keys.to(torch.int64) & 0xFFFFFFFF
        bits = world.bit_length() - 1
        ok = bool((u[1:] >= u[:-1]).all()) if u.numel() > 1 else True
        if u.numel() and bits and args.partition == "msb":
            ok = ok and int(u[0].item()) >> (32 - bits) == rank and int(u[-1].item()) >> (32 - bits) == rank
        ok = ok and backend.check_fault() == 0      # the workspace the sharded path's sorts actually ran in
        cnt = torch.tensor([u.numel(), 0 if ok else 1], dtype=torch.int64, device="cuda")
        dist.all_reduce(cnt)
        assert int(cnt[0].item()) == n * world and int(cnt[1].item()) == 0, "sharded sort failed its check"
        del res, u
    else:
        # `ws` is the workspace every timed sort ran in
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

    def pmc_traffic(rb, pairs, nn):
        """HBM bytes per launch from the committed rocprofv3 --pmc passes (profiles/pmc_summary.json), with where it
        came from: a builder's profiling run of the same kernel at the same n, NOT this run."""
        prof = os.path.join(ROOT, "profiles", "pmc_summary.json")
        try:
            e = json.load(open(prof)).get(f"rank_scatter_r{rb}{'_pairs' if pairs else ''}", {})
            if e.get("n") == nn:
                return e["hbm_bytes_per_launch"], (f"{e.get('source', 'profiles/pmc_summary.json')} (separate rocprofv3 --pmc passes, "
                                                    f"kernel averaged {e.get('kernel_avg_us')} us in that run; not measured in this run)")
        except Exception:
            pass
        return None, None

    def measure_roofline(keys_t, vals_t, rb, pairs, nn, wsx, reps=5):
        """The dominant kernel's own begin/end events (hipExtLaunchKernelGGL on the launch stream) over `reps` sorts.  Where the
        hybrid form runs (lsdsort_timing.hybrid) the global passes are its two, and the local stage is timed by marker events."""
        scat, hist, scan, clear, totals, local = [], [], [], [], [], []
        tile, hybrid, global_passes = None, 0, 32 // rb
        for _ in range(reps):
            # an untimed sort is queued right in front of the timed one (no synchronise between them), so that the timed
            # kernels run as they do inside the timed region -- back to back behind another sort, clocks and caches in
            # their steady state -- and not as the first work after a host wait
            wk, wv = keys_t(), (vals_t() if pairs else None)
            lsd.GPULSDRadixSort(wk, rb, d_vals=wv, algorithm=algo, workspace=wsx)
            kk, vv = keys_t(), (vals_t() if pairs else None)
            tm = lsd.GPULSDRadixSortTimed(kk, rb, d_vals=vv, algorithm=algo, workspace=wsx)
            tile = tm["tile_keys"]
            hybrid = tm["hybrid"]
            global_passes = tm["passes"]
            scat += tm["scatter_ms"]
            hist.append(tm["histogram_ms"])
            scan.append(tm["scan_ms"])
            clear.append(tm["clear_ms"])
            totals.append(tm["total_ms"])
            local.append(tm["local_ms"])
        per_key = 16 if pairs else 8               # one read + one write of the key (and payload) per pass
        scat_ms = float(np.mean(scat))
        achieved = per_key * nn / (scat_ms * 1e-3) / 1e9
        traffic, source = pmc_traffic(rb, pairs, nn)
        roof = {"bound": "hbm", "kernel": "rank_scatter_kernel", "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": traffic, "traffic_source": source, "algorithmic_bytes_per_launch": per_key * nn,
                "launch_ms": round(scat_ms, 4), "launches_per_sort": global_passes}
        stages = {"clear": round(float(np.mean(clear)), 4), "histogram": round(float(np.mean(hist)), 4),
                  "scan": round(float(np.mean(scan)), 4), "scatter_per_pass": round(scat_ms, 4),
                  "global_passes": global_passes, "hybrid_form": bool(hybrid),
                  "local_stage": round(float(np.mean(local)), 4) if hybrid else None,
                  "total_event": round(float(np.mean(totals)), 4)}
        return roof, stages, tile

    # ---- roofline of the dominant kernel (rank-and-scatter), measured live with hipEvents ----
    roofline = None
    stage_ms = None
    sort_tile_keys = None
    if rank == 0:
        roofline, stage_ms, sort_tile_keys = measure_roofline(lambda: pool.fresh()[0], lambda: pool.bufs[(pool.next - 1) % pool.capacity][1],
                                                              r, args.pairs, n, ws)
        sort_bytes = (4 + 16 * passes) * n if args.pairs else 4 * (2 * passes + 1) * n
        stage_ms["sort_algorithmic_gbs"] = round(sort_bytes / (ms_per_step * 1e-3) / 1e9, 1) if not distributed else None
        stage_ms["sort_roofline_frac"] = round(sort_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if not distributed else None
        if stage_ms["hybrid_form"] and not distributed:
            # what the sort that ran really moves: one counting read, 16 / r global passes, one local stage = 4 + (16 / r + 1) x 8 B/key
            per_item = 2 if args.pairs else 1
            moved_per_key = 4 + (16 // r + 1) * 8 * per_item
            moved = moved_per_key * n
            stage_ms["sort_moved_bytes_per_key"] = moved_per_key
            stage_ms["sort_moved_gbs"] = round(moved / (ms_per_step * 1e-3) / 1e9, 1)
            stage_ms["sort_moved_frac_of_peak"] = round(moved / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
            stage_ms["accounting"] = (f"sort_roofline_frac prices the sort at SURVEY 8d's algorithmic {sort_bytes // n} B/key (an LSD sort of {passes} "
                                      f"passes through HBM); the hybrid form moves {moved_per_key} B/key (sort_moved_*): the low 16 bits are sorted "
                                      "inside the CUs' LDS")
        roofline["rank_method"] = lsd.rank_method(r)
        if not distributed and not args.no_live_traffic and args.algorithm == "onesweep" and args.tile_config < 0:
            live = live_pmc_traffic(r, args.pairs, log2_keys)
            if live is not None:
                rs = live["rank_scatter"]
                roofline["traffic"] = rs["bytes"]
                roofline["traffic_source"] = ("measured in this run: two separate rocprofv3 --pmc passes (FETCH_SIZE x 2, WRITE_SIZE; KiB) over "
                                              f"child processes running the same sort on the same input (tools/prof_target.py), {rs['launches']} launches, "
                                              f"kernel {rs['kernel_us_under_profiler']} us under the profiler; FETCH_SIZE x 2 on the upfront read's "
                                              f"known 4n bytes: ratio {live.get('calibration_ratio')}")
                roofline["traffic_over_algorithmic"] = round(rs["bytes"] / roofline["algorithmic_bytes_per_launch"], 4)
                # the other two kernels of the sort, priced the same way (VERDICT r2 task 3: a roofline object for stage 1).  Their
                # durations come from the profiler's own dispatch timestamps in these passes (no launch-stream events inside the
                # library for them), so they read a little long against an unprofiled run.
                per_item = 8 if args.pairs else 4
                if "stage1" in live:
                    st = live["stage1"]
                    ach = per_item * n / (st["kernel_us_under_profiler"] * 1e-6) / 1e9
                    roofline["stage1"] = {"bound": "hbm", "kernel": "hybrid_histograms_kernel" if stage_ms["hybrid_form"] else "joint_histograms_kernel",
                                          "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                                          "traffic": st["bytes"], "algorithmic_bytes_per_launch": per_item * n,
                                          "launch_ms": round(st["kernel_us_under_profiler"] / 1e3, 4), "launches_per_sort": 1,
                                          "timing": "rocprofv3 dispatch timestamps of the PMC passes"}
                if "local" in live:
                    lc = live["local"]
                    ach = 8 * n / (lc["kernel_us_under_profiler"] * 1e-6) / 1e9
                    roofline["local_stage"] = {"bound": "hbm", "kernel": "local_sort_kernel", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS,
                                               "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": lc["bytes"],
                                               "algorithmic_bytes_per_launch": 8 * n, "launch_ms": round(lc["kernel_us_under_profiler"] / 1e3, 4),
                                               "launches_per_sort": 1, "timing": "rocprofv3 dispatch timestamps of the PMC passes",
                                               "note": "two digit passes from LDS to LDS per bucket: bound by LDS work, not by HBM"}

    # ---- secondary configs on the same box (N=1 only): configs[1] (r=4) and configs[4] (pairs), stage rows ----
    extra = {}
    if rank == 0 and not distributed and not args.no_extra and not args.pairs and r == 8 and log2_keys == 28:
        def config_line(rb, pairs, nn, steps=8):
            """The same measurement as the headline, on another BASELINE config: `steps` back-to-back sorts of fresh
            copies between two synchronisations, then the kernel's own events for its roofline object."""
            vals = torch.arange(nn, dtype=torch.int32, device="cuda") if pairs else None
            w2 = lsd.alloc_workspace(nn, rb, pairs, algo)
            copies = [(master[:nn].clone(), vals.clone() if pairs else None) for _ in range(steps + 1)]
            lsd.GPULSDRadixSort(copies[0][0], rb, d_vals=copies[0][1], algorithm=algo, workspace=w2)   # warm-up
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for kk, vv in copies[1:]:
                lsd.GPULSDRadixSort(kk, rb, d_vals=vv, algorithm=algo, workspace=w2)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / steps * 1e3
            it = iter(copies * 2)

            def fresh_pair():
                kk, vv = next(it)
                kk.copy_(master[:nn])
                if vv is not None:
                    vv.copy_(vals)
                fresh_pair.last = vv
                return kk
            roof, stages, tile = measure_roofline(fresh_pair, lambda: fresh_pair.last, rb, pairs, nn, w2, reps=3)
            pp = 32 // rb
            sort_bytes = (4 + 16 * pp) * nn if pairs else 4 * (2 * pp + 1) * nn
            unit = "Mpairs/s" if pairs else "Mkeys/s"
            return {"value": round(nn / (ms * 1e-3) / 1e6, 1), "unit": unit, "ms_per_step": round(ms, 4), "steps": steps,
                    "n": nn, "radix_bits": rb, "pairs": pairs, "tile_keys": tile, "roofline": roof, "stages_ms": stages,
                    "sort_algorithmic_bytes_per_item": sort_bytes // nn,
                    "sort_roofline_frac": round(sort_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
        extra["config2_r4_256M_keys"] = config_line(4, False, n)
        extra["config5_pairs_r8_128M_pairs"] = config_line(8, True, n // 2)
        extra["r4_256M_keys_mkeys_s"] = extra["config2_r4_256M_keys"]["value"]
        extra["pairs_r8_128M_pairs_mpairs_s"] = extra["config5_pairs_r8_128M_pairs"]["value"]

        # The four-pass form (every digit through global memory, the reference's structure) on the same input, when the headline
        # ran the hybrid form: lsdsort_set_hybrid(0).
        if stage_ms["hybrid_form"]:
            lsd.set_hybrid(False)
            try:
                copies = [master.clone() for _ in range(9)]
                lsd.GPULSDRadixSort(copies[0], r, algorithm=algo, workspace=ws)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for kk in copies[1:]:
                    lsd.GPULSDRadixSort(kk, r, algorithm=algo, workspace=ws)
                torch.cuda.synchronize()
                ms4 = (time.perf_counter() - t0) / 8 * 1e3
                roof4, stages4, _ = measure_roofline(lambda: pool.fresh()[0], lambda: None, r, False, n, ws, reps=3)
                del copies
            finally:
                lsd.set_hybrid(True)
            extra["four_pass_form"] = {"value": round(n / (ms4 * 1e-3) / 1e6, 1), "unit": "Mkeys/s", "ms_per_step": round(ms4, 4),
                                       "sort_roofline_frac": round(36 * n / (ms4 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                       "roofline": roof4, "stages_ms": stages4,
                                       "note": "lsdsort_set_hybrid(0): four global passes, 36 B/key moved (the structure of GPULSDRadixSort, .cu:844-905)"}

        # What the default rank form (one returning LDS add per key, probed on the device) buys over the
        # architecture-guaranteed peer-mask form: the same sort with lsdsort_set_rank_method(0) and (2).
        def method_ms(m, steps=4):
            lsd.set_rank_method(m)
            lsd.set_hybrid(False)           # like with like: the hybrid form needs rank form 2, so both run the four-pass form
            copies = [master.clone() for _ in range(steps + 1)]
            lsd.GPULSDRadixSort(copies[0], r, algorithm=algo, workspace=ws)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for kk in copies[1:]:
                lsd.GPULSDRadixSort(kk, r, algorithm=algo, workspace=ws)
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / steps * 1e3
        ab0, ab2 = method_ms(0), method_ms(2)
        lsd.set_rank_method(args.rank_method if args.rank_method >= 0 else -1)
        lsd.set_hybrid(not args.no_hybrid)
        extra["rank_method_ab"] = {"method_0_peer_mask_ms": round(ab0, 4), "method_2_lds_add_ms": round(ab2, 4),
                                   "in_use": lsd.rank_method(r), "form": "four global passes (hybrid form off for both)",
                                   "opt_out": "lsdsort_set_rank_method(0) (include/lsdsort.h); method 2 is only used when the device probe passes"}

        # Key distributions the reference never tests: made on the device from the workload's keys, sorted
        # with the same call, each checked for sortedness.  ms per sort, best of 3 behind a warm-up.
        def dist_ms(make):
            src = make()
            best = None
            for i in range(4):
                k = src.clone()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                lsd.GPULSDRadixSort(k, r, algorithm=algo, workspace=ws)
                e1.record()
                torch.cuda.synchronize()
                if i:
                    t = e0.elapsed_time(e1)
                    best = t if best is None or t < best else best
            u = k.to(torch.int64) & 0xFFFFFFFF
            assert bool((u[1:] >= u[:-1]).all()), "distribution case not sorted"
            assert lsd.lib().lsdsort_check_device(ws.data_ptr(), torch.cuda.current_stream().cuda_stream) == 0
            del src, k, u
            return round(best, 4)

        def sorted_keys():
            k = master.clone()
            lsd.GPULSDRadixSort(k, r, algorithm=algo, workspace=ws)
            return k
        low16 = master & 0xFFFF
        extra["key_distributions_ms"] = {
            "uniform": dist_ms(lambda: master),
            "sorted": dist_ms(sorted_keys),
            "constant": dist_ms(lambda: torch.full_like(master, 0x12345678)),
            "one_live_byte": dist_ms(lambda: master & 0xFF),
            "values_below_2p20": dist_ms(lambda: master & 0xFFFFF),
            "half_zero": dist_ms(lambda: torch.where((master & 0x2000) != 0, master, torch.zeros_like(master))),
            "ninety_pct_one_value": dist_ms(lambda: torch.where(low16 % 10 != 0, torch.full_like(master, -0x7FFFFFFF), master)),
            "note": "2^%d keys each; dead passes skipped on the device, heavy values counted from scalar registers (DESIGN.md section 4.7)" % log2_keys,
        }
        del low16

        # Stage micro-benchmarks: the counterparts of the reference's TestBuildHistogram
        # (.cu:704, sweep .cu:1123-1136) and TestGPUPrefixSum (.cu:304, sweep .cu:1083-1092) harnesses, through the
        # stage-level C-ABI entries, for the reference's radix widths rs = {1, 2, 4, 8} (.cu:1055-1062).
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
        stage = {}
        for rb in (1, 2, 4, 8):
            hist_ms = time_stage(lambda: lsd.BuildHistograms(master, rb, 0))
            h = lsd.BuildHistograms(master, rb, 0)
            offs_ms = time_stage(lambda: lsd.BuildOffsets(h, rb))
            stage[f"tile_histograms_r{rb}"] = {"ms": round(hist_ms, 4), "read_gbs": round(4 * n / hist_ms / 1e6, 1),
                                               "tile_keys": lsd.tile_keys(rb)}
            stage[f"tile_offsets_r{rb}"] = {"ms": round(offs_ms, 4), "tiles": int(h.shape[0]),
                                            "table_gbs": round(3 * 4 * h.numel() / offs_ms / 1e6, 1)}
            del h
        stage["replaces"] = ("tile_histograms: BuildHistogramsKernel .cu:660-702 (BenchmarkBuildHistogram.md rows); "
                             "tile_offsets: offset construction .cu:862-895 (BenchmarkPrefixSum.md rows)")
        dh_ms = time_stage(lambda: lsd.DigitHistograms(master, 8))
        stage["digit_histograms_all_passes_r8"] = {"ms": round(dh_ms, 4), "read_gbs": round(4 * n / dh_ms / 1e6, 1)}
        extra["stage_bench"] = stage

    # ---- CPU baseline: the reference's std::sort path on the host, one thread ----
    cpu_baseline = None
    if rank == 0 and not distributed and not args.no_cpu_baseline:
        import oracle     # test infrastructure; used only for this reported baseline

        oracle.lib()
        m = min(n, 1 << args.cpu_sample_log2)
        sample = host_keys[:m]
        t_std = oracle.time_std_sort(sample)
        t_lsd = oracle.time_lsd_sort(sample, 8)
        small = host_keys[:1 << 20]                  # BASELINE configs[0]: 1M keys, the reference's own CPU-runnable case
        t_small = min(oracle.time_std_sort(small) for _ in range(3))
        t_small_lsd = min(oracle.time_lsd_sort(small, 8) for _ in range(3))
        cpu_baseline = {"value": round(m / (t_std * 1e-3) / 1e6, 2), "unit": "Mkeys/s", "cores": 1, "kind": "port",
                        "sample": f"std::sort (LSDRadixSort.cu:97) of the first 2^{m.bit_length() - 1} keys of the "
                                  f"workload ({'all of it' if m == n else 'O(n log n): the full 2^' + str(log2_keys) + ' would be slower per key'}), "
                                  f"1 thread, {t_std / 1e3:.1f} s",
                        "lsd_r8_mkeys_s": round(m / (t_lsd * 1e-3) / 1e6, 2),
                        "lsd_r8_note": "restated reference CPU LSD (LSDRadixSort.cu:25-69), r=8, same sample",
                        "config1_2p20_std_sort_mkeys_s": round(small.size / (t_small * 1e-3) / 1e6, 2),
                        "config1_2p20_lsd_r8_mkeys_s": round(small.size / (t_small_lsd * 1e-3) / 1e6, 2),
                        "config1_note": "BASELINE configs[0]: 2^20 keys, std::sort and the restated CPU LSD, 1 thread, best of 3",
                        "host_cpus": os.cpu_count()}
        if oracle.ref_available():
            # the reference's own LSDRadixSort, compiled where it lies by oracle/Makefile (`make ref`) and carried along as
            # oracle/_ref/libref_lsd.so: beside the restatement it checks, on the same sample
            t_ref = oracle.time_ref_lsd_sort(sample, 8)
            cpu_baseline["reference_lsd_r8_mkeys_s"] = round(m / (t_ref * 1e-3) / 1e6, 2)
            cpu_baseline["reference_lsd_r8_note"] = ("kind 'reference': LSDRadixSort (LSDRadixSort.cu:62-69) itself, compiled from the "
                                                     "reference tree into oracle/_ref, r=8, same sample, 1 thread")

    if rank == 0:
        if distributed:
            total_txt = f"2^{(n * world).bit_length() - 1} keys over {world} GPU{'s' if world > 1 else ''}" if (n * world) & (n * world - 1) == 0 else f"{n * world} keys"
            workload = (f"{total_txt} = 2^{log2_keys} uniform uint32 keys per GPU (mt19937 seed=rank), "
                        f"{'MSB-bucket' if args.partition == 'msb' else 'sampled-splitter'} partition + {'grouped ncclSend/ncclRecv (C++ step, lsdsort_sharded_u32_device_ex)' if exchange_path == 'c++' else 'torch.distributed all-to-all'} over xGMI + "
                        f"local {r}-bit LSD sort per step (a shard of this size alone: "
                        + (f"hybrid form, {16 // r} global passes + LDS-resident local stage" if stage_ms and stage_ms.get("hybrid_form") else f"{passes} global passes")
                        + "; the step's sort plans below the shard's key prefix)"
                        + (" [BASELINE configs[3]]" if n * world == 1 << 30 and world == 8 else ""))
        else:
            form = (f", hybrid form (bits 16-31 by {16 // r} global passes, the low bits inside each CU's LDS; decided on the device)"
                    if stage_ms and stage_ms.get("hybrid_form") else f", {passes} global passes")
            workload = (f"2^{log2_keys} uniform uint32 {'key+payload pairs' if args.pairs else 'keys'} per GPU "
                        f"(mt19937 seed=rank), {r}-bit radix{form}, {args.algorithm}, device-resident")
        line = {
            "metric": "Mkeys/s sorting uniform uint32, 1 GiB, 1/2/4/8 MI355X; % HBM roofline",
            "value": round(mkeys, 1), "unit": "Mkeys/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            "dtype": "u32", "data": "synthetic",
            "config": {"workload": workload, "keys_per_gpu": n, "total_keys": n * world, "radix_bits": r,
                       "algorithm": args.algorithm, "pairs": bool(args.pairs), "tile_keys": sort_tile_keys,
                       "hybrid_form": bool(stage_ms and stage_ms.get("hybrid_form"))},
            "roofline": roofline, "cpu_baseline": cpu_baseline, "stages_ms": stage_ms, "extra": extra,
        }
        print(json.dumps(line), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main() or 0)
