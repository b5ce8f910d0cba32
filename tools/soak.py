#!/usr/bin/env python3
"""Bounded soak (GPU box): a few hundred sorts of random size / radix / kind back to back on two streams, each
checked against torch.sort.  Looks for rare protocol problems (chain spins, fault word), not for speed."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lsdradixsort_amd as lsd

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 7)
budget_s = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
lo_lg = int(sys.argv[3]) if len(sys.argv) > 3 else 0
hi_lg = int(sys.argv[4]) if len(sys.argv) > 4 else 24
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
gen = torch.Generator(device="cuda"); gen.manual_seed(99)
t0 = time.time(); done = 0; last_note = t0
while time.time() - t0 < budget_s and done < 200000:
    lg = rng.integers(lo_lg, hi_lg + 1)
    n = int(rng.integers(1 << max(lg - 1, 0), (1 << lg) + 1))
    r = int(rng.choice([8, 8, 8, 4, 4, 2, 1])) if n <= (1 << 20) else int(rng.choice([8, 8, 4]))
    kind = int(rng.integers(0, 9))
    s = streams[done % 2]
    with torch.cuda.stream(s):
        k = torch.randint(-(1 << 31), (1 << 31) - 1, (n,), dtype=torch.int32, device="cuda", generator=gen)
        if kind == 1:
            k &= int(rng.integers(0, 1 << 31))
        if kind == 2:
            k = torch.sort(k).values
        if kind == 5:   # heavy values: a random share of the keys on one value, another share on a second one
            a, b = int(rng.integers(1, 100)), int(rng.integers(0, 50))
            sel = torch.randint(0, 100, (n,), dtype=torch.int32, device="cuda", generator=gen)
            k = torch.where(sel < a, torch.full_like(k, int(rng.integers(-(1 << 31), 1 << 31))), k)
            k = torch.where(sel >= 100 - b, torch.full_like(k, int(rng.integers(-(1 << 31), 1 << 31))), k)
        if kind == 6:   # dead bytes: whole digits the same for every key (skipped passes), odd and even numbers of them
            keep = [0xFFFFFFFF, 0x0000FFFF, 0x00FFFFFF, 0xFF, 0x7F000000, 0x00FF0000, 0x7F0000FF, 0][int(rng.integers(0, 8))]
            k = (k & keep) | (int(rng.integers(0, 1 << 31)) & ~keep & 0x7FFFFFFF)
        ws = lsd.alloc_workspace(n, r, kind == 3)
        if kind == 3 and r >= 4:
            v = torch.arange(n, dtype=torch.int32, device="cuda")
            ref = torch.sort((k.to(torch.int64) & 0xFFFFFFFF), stable=True)
            lsd.GPULSDRadixSort(k, r, d_vals=v, workspace=ws, stream=s)
            assert torch.equal(v.to(torch.int64), ref.indices), (done, n, r, "pairs")
            assert torch.equal(k.to(torch.int64) & 0xFFFFFFFF, ref.values), (done, n, r, "pairs keys")
        elif kind == 7 and r >= 4:   # a shard: keys that share their top t bits, sorted with and (every other time) against the hint
            t = int(rng.integers(1, 9))
            k = (((k.to(torch.int64) & 0xFFFFFFFF) >> t) | (int(rng.integers(0, 1 << t)) << (32 - t)))
            k = ((k + (1 << 31)) % (1 << 32) - (1 << 31)).to(torch.int32)
            if done % 4 == 3 and n > 2:
                k[n // 2] ^= -(1 << 31)                      # one key outside the prefix: the device must notice
            ref = torch.sort(k.to(torch.int64) & 0xFFFFFFFF).values
            st = lsd.lib().lsdsort_u32_device_prefixed(k.data_ptr(), ws.data_ptr(), ws.numel(), n, r, t, s.cuda_stream)
            assert st == 0, (done, n, r, t, st)
            assert torch.equal(k.to(torch.int64) & 0xFFFFFFFF, ref), (done, n, r, "prefixed", t)
        elif kind == 8 and r >= 4:   # two or three payload arrays
            np_ = int(rng.integers(2, 4))
            ws = torch.empty(max(int(lsd.lib().lsdsort_workspace_bytes(n, r, np_)), 256), dtype=torch.uint8, device="cuda")
            idx = torch.arange(n, dtype=torch.int32, device="cuda")
            pay = [(idx * (2 * e + 3) + e).to(torch.int32) for e in range(np_)]
            k &= int(rng.integers(0, 1 << 31)) | 0xFF          # duplicates: stability matters
            ref = torch.sort((k.to(torch.int64) & 0xFFFFFFFF), stable=True)
            want = [p_[ref.indices] for p_ in pay]
            lsd.GPUSortMulti(k, pay, r=r, workspace=ws, stream=s)
            assert torch.equal(k.to(torch.int64) & 0xFFFFFFFF, ref.values), (done, n, r, "multi keys")
            for e in range(np_):
                assert torch.equal(pay[e], want[e]), (done, n, r, "multi payload", e)
        elif kind == 4 and r >= 4:
            ref = torch.sort(k, descending=bool(done & 1)).values
            lsd.GPUSortTyped(k, "int32", bool(done & 1), r=r, workspace=ws, stream=s)
            assert torch.equal(k, ref), (done, n, r, "int32")
        else:
            ref = torch.sort(k.to(torch.int64) & 0xFFFFFFFF).values
            lsd.GPULSDRadixSort(k, r, workspace=ws, stream=s)
            assert torch.equal(k.to(torch.int64) & 0xFFFFFFFF, ref), (done, n, r, kind)
        assert lsd.lib().lsdsort_check_device(ws.data_ptr(), s.cuda_stream) == 0, (done, "fault word")
    done += 1
    if time.time() - last_note > 60.0:          # a line a minute: the GPU box takes ten silent minutes for a hang
        last_note = time.time()
        print(f"... {done} sorts, {last_note - t0:.0f} s", flush=True)
torch.cuda.synchronize()
print(f"soak ok: {done} sorts in {time.time() - t0:.1f} s")
