"""Non-uniform keys on the DEFAULT tile of large sorts (1024 threads x 32 keys, one workgroup per CU: every sort of 2^23 keys
and more), keys and stable pairs, 4- and 8-bit digits -- the cell the round-2 memory fault happened in (gpurun_out/dist8.log:
r = 4, sorted keys, 2^28) and that the smaller parity cases (2^20 .. 2^21 keys: the 512-thread tiles)
never reached.  The reference checks every run bit-exactly (LSDRadixSort.cu:1018) but only ever on uniform keys (.cu:1146).

Checker: torch.sort on the same device array (rocPRIM, test-only) -- bit-exact; for pairs the STABLE argsort, so the order
among equal keys (= the rank the heavy-value paths compute from scalar counts) is pinned too.
"""
import pytest

pytestmark = pytest.mark.gpu

N_TILE = (1 << 23) + 4321          # ragged, above the 2^23 switch to the 32768-key tile


def _u64(t):
    import torch

    return t.to(torch.int64) & 0xFFFFFFFF


def _shapes(n, seed, device="cuda"):
    """name -> int32 device tensor holding n uint32 bit patterns."""
    import torch

    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    base = torch.randint(-(1 << 31), (1 << 31) - 1, (n,), dtype=torch.int32, device=device, generator=gen)
    u = _u64(base)

    def i32(x):   # uint32 values held in int64 -> int32 bit pattern
        return ((x + (1 << 31)) % (1 << 32) - (1 << 31)).to(torch.int32)

    srt = torch.sort(u).values
    out = {
        "sorted": lambda: i32(srt),
        "reverse": lambda: i32(torch.flip(srt, dims=[0])),
        "half_zero": lambda: i32(torch.where(((u >> 13) & 1) != 0, u, torch.zeros_like(u))),
        "ninety_pct_one_value": lambda: i32(torch.where((u % 10) != 0, torch.full_like(u, 0x80000001), u)),
        "two_values": lambda: i32(torch.where((u & (1 << 17)) != 0, torch.full_like(u, 0x11111111), torch.full_like(u, 0xEEEEEEEE))),
        "four_values_per_digit": lambda: i32(u & 0x03030303),
        "runs": lambda: i32(torch.repeat_interleave(u[: n // 300 + 1], 300)[:n]),
        "three_heavy_values": lambda: i32(torch.where(u % 7 < 2, torch.full_like(u, 5),
                                          torch.where(u % 7 < 4, torch.full_like(u, 0x05000005),
                                                      torch.where(u % 7 < 6, torch.full_like(u, 0xFF0000FF), u)))),
        "sorted_low16_only": lambda: i32(torch.sort(u & 0xFFFF).values),
    }
    return out


@pytest.mark.parametrize("r", [8, 4])
def test_default_tile_nonuniform_keys(gpu, r):
    import torch

    rank_forms = (-1, 0)   # the default (returning LDS add where probed) and the peer-mask fallback
    for name, make in _shapes(N_TILE, 500 + r).items():
        keys = make()
        expect = torch.sort(_u64(keys)).values
        for form in rank_forms:
            gpu.set_rank_method(form)
            try:
                d = keys.clone()
                gpu.GPULSDRadixSort(d, r, check_fault=True)
                assert torch.equal(_u64(d), expect), (name, r, form)
            finally:
                gpu.set_rank_method(-1)
        del keys, expect


@pytest.mark.parametrize("r", [8, 4])
def test_default_tile_nonuniform_pairs_are_stable(gpu, r):
    import torch

    vals0 = torch.arange(N_TILE, dtype=torch.int32, device="cuda")
    for name, make in _shapes(N_TILE, 600 + r).items():
        keys = make()
        expect = torch.sort(_u64(keys), stable=True)
        d, v = keys.clone(), vals0.clone()
        gpu.GPULSDRadixSort(d, r, d_vals=v, check_fault=True)
        assert torch.equal(_u64(d), expect.values), (name, r)
        assert torch.equal(v.to(torch.int64), expect.indices), (name, r, "order among equal keys")
        del keys, expect, d, v


@pytest.mark.parametrize("r", [4, 8])
def test_full_size_nonuniform_keys(gpu, r):
    """The same shapes at BASELINE's full size, 2^28 keys, once per radix: the exact cell of the dist8.log fault (r = 4,
    sorted) among them.  Bit-exact against torch.sort of the same array; the workspace is reused across the shapes, as
    tools/dist_perf.py did when the fault happened."""
    import torch

    n = 1 << 28
    ws = gpu.alloc_workspace(n, r)
    shapes = _shapes(n, 700 + r)
    for name in ("sorted", "reverse", "half_zero", "ninety_pct_one_value", "two_values", "four_values_per_digit", "runs"):
        keys = shapes[name]()
        expect = torch.sort(_u64(keys)).values
        gpu.GPULSDRadixSort(keys, r, workspace=ws, check_fault=True)
        assert torch.equal(_u64(keys), expect), (name, r)
        del keys, expect
        torch.cuda.empty_cache()


def test_full_size_nonuniform_pairs(gpu):
    """2^27 pairs (BASELINE configs[4]) with heavy values and sorted keys: stable against torch's stable sort."""
    import torch

    n = 1 << 27
    shapes = _shapes(n, 801)
    vals0 = torch.arange(n, dtype=torch.int32, device="cuda")
    for name in ("sorted", "half_zero", "ninety_pct_one_value", "four_values_per_digit"):
        keys = shapes[name]()
        expect = torch.sort(_u64(keys), stable=True)
        v = vals0.clone()
        gpu.GPULSDRadixSort(keys, 8, d_vals=v, check_fault=True)
        assert torch.equal(_u64(keys), expect.values), name
        assert torch.equal(v.to(torch.int64), expect.indices), (name, "order among equal keys")
        del keys, expect, v
        torch.cuda.empty_cache()


@pytest.mark.parametrize("r", [8, 4])
def test_multi_payload_arrays_follow_their_keys(gpu, r):
    """lsdsort_multi_u32_device: keys with two and three payload arrays (the building block of the record sorts): every array
    comes out in the order of torch's STABLE argsort of the keys.  Sizes in every tile class (4096 .. 32768 keys per tile),
    duplicate-heavy keys, dead digits (pass skipping: the copy back covers every array), a constant key."""
    import torch

    gen = torch.Generator(device="cuda")
    for case, (n, mask) in enumerate([(4097, 0xFFFFFFFF), ((1 << 19) + 5, 0x00FF00FF), ((1 << 21) + 77, 0xFFFFFFFF),
                                      ((1 << 23) + 4321, 0x0000FFFF), ((1 << 23) + 4321, 0xFFFFFFFF), (70001, 0), (1, 0xFFFFFFFF)]):
        gen.manual_seed(900 + case + r)
        base = torch.randint(-(1 << 31), (1 << 31) - 1, (n,), dtype=torch.int32, device="cuda", generator=gen)
        keys = (base.to(torch.int64) & mask & 0xFFFFFFFF)
        keys32 = ((keys + (1 << 31)) % (1 << 32) - (1 << 31)).to(torch.int32)
        expect = torch.sort(keys, stable=True)
        for num in (2, 3):
            pay = [torch.arange(n, dtype=torch.int32, device="cuda") * (e + 1) + e for e in range(num)]
            want = [p[expect.indices] for p in pay]
            d = keys32.clone()
            gpu.GPUSortMulti(d, pay, r, check_fault=True)
            assert torch.equal(_u64(d), expect.values), (n, hex(mask), num, r)
            for e in range(num):
                assert torch.equal(pay[e], want[e]), (n, hex(mask), num, r, e)
