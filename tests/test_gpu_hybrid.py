"""The hybrid form of large keys-only 8-bit sorts (hybrid.hip, local_sort.hip): two global passes on the high bytes, then every
top-15-bit bucket finished in LDS.  No reference counterpart (every reference pass goes through global memory,
LSDRadixSort.cu:844-905); the RESULT has one: the sorted array (.cu:1018, .cu:120).  Checked bit-exact against torch.sort of
the same device array, with the form switched on and off, on keys for which the device takes it and on keys for which it
must refuse it -- and the local stage on its own against numpy.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _u64(t):
    import torch

    return t.to(torch.int64) & 0xFFFFFFFF


def _i32(x):
    import torch

    return ((x + (1 << 31)) % (1 << 32) - (1 << 31)).to(torch.int32)


def test_local_stage_alone(gpu):
    """lsdsort_local_sort_u32_device: buckets of every shape a workgroup meets -- empty, one key, around a row of 512, around
    the 16384-key capacity (the larger one must be left untouched) -- sorted in place by their low 9 / 17 / 24 / 27 bits."""
    import torch

    rng = np.random.default_rng(11)
    sizes = [0, 1, 2, 63, 64, 65, 511, 512, 513, 1000, 4096, 8191, 8192, 16383, 16384, 16385, 0, 7, 12345]
    bases = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint32)
    n = int(bases[-1])
    for low_bits in (9, 17, 24, 27, 1):
        keys = rng.integers(0, 1 << 32, size=n, dtype=np.uint64).astype(np.uint32)
        if low_bits == 24:
            keys[: n // 2] &= np.uint32(0xFF0000FF)          # few distinct low values: heavy digits in every local pass
        db = gpu.to_device(bases)
        mask = np.uint32((1 << low_bits) - 1)
        for with_vals in (False, True):
            d = gpu.to_device(keys)
            vals = np.arange(n, dtype=np.uint32) * np.uint32(7) + np.uint32(3)
            dv = gpu.to_device(vals) if with_vals else None
            st = gpu.lib().lsdsort_local_sort_u32_device(d.data_ptr(), dv.data_ptr() if with_vals else None, db.data_ptr(), len(sizes), low_bits,
                                                         torch.cuda.current_stream().cuda_stream)
            assert st == 0
            got = gpu.to_host(d)
            got_v = gpu.to_host(dv) if with_vals else None
            for b, size in enumerate(sizes):
                lo, hi = int(bases[b]), int(bases[b + 1])
                part = keys[lo:hi]
                if size > 16384:
                    assert np.array_equal(got[lo:hi], part), "a bucket above the capacity was touched"
                    continue
                order = np.argsort(part & mask, kind="stable")
                assert np.array_equal(got[lo:hi], part[order]), (low_bits, b, size, with_vals)
                if with_vals:
                    assert np.array_equal(got_v[lo:hi], vals[lo:hi][order]), (low_bits, b, size, "payload order = stable order")


@pytest.mark.parametrize("log2n,extra,radix", [(26, 999, 8), (27, 0, 8), (27, 12345, 8), (28, 777, 8), (29, 4242, 8), (25, 6445568, 8), (25, 77, 4), (24, 99, 4), (26, 4097, 4), (27, 31, 4), (28, 5, 4),
                                               (29, 11, 4)])
def test_hybrid_equals_torch_sort_and_the_four_pass_form(gpu, log2n, extra, radix):
    """radix 8: two global passes; radix 4 (BASELINE configs[1]'s digit width): four, their count fields derived by the planner."""
    import torch

    n = (1 << log2n) + extra
    bb = {24: 14, 25: 14, 26: 14, 27: 14, 28: 15, 29: 16}[log2n]   # bucket = the top bb bits (lsd_kernels.hpp hybrid_bucket_bits, keys)
    sh, low = 32 - bb, (1 << (32 - bb)) - 1
    gen = torch.Generator(device="cuda")
    gen.manual_seed(4000 + log2n + extra)
    base = torch.randint(-(1 << 31), (1 << 31) - 1, (n,), dtype=torch.int32, device="cuda", generator=gen)
    ws = gpu.alloc_workspace(n, radix)
    shapes = {
        "uniform": lambda: base.clone(),
        "sorted": lambda: _i32(torch.sort(_u64(base)).values),
        "low_bits_dead": lambda: _i32(_u64(base) & 0xFFFFFE00),                       # the local stage's first digit constant
        "half_zero": lambda: _i32(torch.where(((_u64(base) >> 13) & 1) != 0, _u64(base), torch.zeros_like(_u64(base)))),   # bucket 0 too large
        "small_range": lambda: _i32(_u64(base) & 0x000FFFFF),                         # one bucket holds everything
        "one_bucket_just_too_large": lambda: _i32(torch.cat([_u64(base[: n - 16385]), (_u64(base[:16385]) & low) | (5 << sh)])),
        # fifty buckets between the two capacities of the local stage (10240 < size <= 16384): the planner's list, walked by
        # the large variant, while the small variant takes the others
        "fifty_buckets_above_the_small_capacity": lambda: _i32(torch.cat([
            _u64(base[: n - 50 * hot]), (_u64(base[: 50 * hot]) & low) | ((torch.arange(50 * hot, device="cuda") // hot * 301 + 77) << sh)])),
    }
    hot = 16000 - (n >> bb) - 700          # a hot bucket: its share of the uniform keys plus this many, under 16384 in all
    taken = {}
    for name, make in shapes.items():
        keys = make()
        expect = torch.sort(_u64(keys)).values
        d = keys.clone()
        tm = gpu.GPULSDRadixSortTimed(d, radix, workspace=ws)
        taken[name] = tm["hybrid"]
        assert torch.equal(_u64(d), expect), (name, "hybrid on", tm["hybrid"])
        assert gpu.lib().lsdsort_check_device(ws.data_ptr(), torch.cuda.current_stream().cuda_stream) == 0, name
        gpu.set_hybrid(False)
        try:
            d2 = keys.clone()
            tm2 = gpu.GPULSDRadixSortTimed(d2, radix, workspace=ws)
            assert tm2["hybrid"] == 0
            assert torch.equal(d2, d), (name, "the two forms differ")
        finally:
            gpu.set_hybrid(True)
        del keys, expect, d, d2
    assert taken["uniform"] == 1 and taken["sorted"] == 1 and taken["low_bits_dead"] == 1, taken
    assert taken["fifty_buckets_above_the_small_capacity"] == (1 if hot > 10240 - (n >> bb) else taken["fifty_buckets_above_the_small_capacity"]), taken
    assert taken["half_zero"] == 0 and taken["small_range"] == 0 and taken["one_bucket_just_too_large"] == 0, taken


@pytest.mark.parametrize("n", [24_000_000, (1 << 25) + 11, (1 << 26) + 77, (1 << 27) + 4321])
def test_hybrid_pairs_are_stable(gpu, n):
    """Key/value pairs through the hybrid form (BASELINE configs[4]'s size): payload = input position, so the output must be
    torch's STABLE sort -- on keys the device takes (uniform; duplicates inside the buckets: 20 live bits below the bucket's) and on
    keys it refuses (half zeros); the same with the form off."""
    import torch

    gen = torch.Generator(device="cuda")
    gen.manual_seed(77)
    base = torch.randint(-(1 << 31), (1 << 31) - 1, (n,), dtype=torch.int32, device="cuda", generator=gen)
    vals0 = torch.arange(n, dtype=torch.int32, device="cuda")
    ws = gpu.alloc_workspace(n, 8, True)
    shapes = {
        "uniform": (lambda: base.clone(), 1),
        "duplicates_in_buckets": (lambda: _i32(_u64(base) & 0xFFFE0F0F), 1),
        # a local digit that is the same for every key of a bucket is skipped (local_sort.hip): the payloads must stay in input
        # order among equal keys all the same
        "first_local_digit_dead": (lambda: _i32(_u64(base) & 0xFFFFFE00), 1),
        "second_local_digit_dead": (lambda: _i32(_u64(base) & 0xFFFC01FF), 1),
        "half_zero": (lambda: _i32(torch.where(((_u64(base) >> 13) & 1) != 0, _u64(base), torch.zeros_like(_u64(base)))), 0),
    }
    for name, (make, expect_hybrid) in shapes.items():
        keys = make()
        expect = torch.sort(_u64(keys), stable=True)
        for on in (True, False):
            gpu.set_hybrid(on)
            try:
                d, v = keys.clone(), vals0.clone()
                tm = gpu.GPULSDRadixSortTimed(d, 8, d_vals=v, workspace=ws)
                assert tm["hybrid"] == (expect_hybrid if on else 0), (name, on, tm["hybrid"])
                assert torch.equal(_u64(d), expect.values), (name, on)
                assert torch.equal(v.to(torch.int64), expect.indices), (name, on, "order among equal keys")
                assert gpu.lib().lsdsort_check_device(ws.data_ptr(), torch.cuda.current_stream().cuda_stream) == 0
            finally:
                gpu.set_hybrid(True)
        del keys, expect


def test_hybrid_is_not_tried_outside_its_range(gpu):
    """Below 3.8e7 keys at 8-bit digits (2.2e7 pairs, 2^24 items at 4-bit digits) and at 1- and 2-bit digits the ordinary form runs."""
    import torch

    n = (1 << 25) - 5
    d = torch.randint(-(1 << 31), (1 << 31) - 1, (n,), dtype=torch.int32, device="cuda")
    expect = torch.sort(_u64(d)).values
    tm = gpu.GPULSDRadixSortTimed(d, 8)
    assert tm["hybrid"] == 0 and torch.equal(_u64(d), expect)
    n = (1 << 24) - 5
    d = torch.randint(-(1 << 31), (1 << 31) - 1, (n,), dtype=torch.int32, device="cuda")
    expect = torch.sort(_u64(d)).values
    tm = gpu.GPULSDRadixSortTimed(d, 4)
    assert tm["hybrid"] == 0 and torch.equal(_u64(d), expect)
    n = (1 << 27) + 3
    d = torch.randint(-(1 << 31), (1 << 31) - 1, (n,), dtype=torch.int32, device="cuda")
    tm = gpu.GPULSDRadixSortTimed(d.clone(), 2)
    assert tm["hybrid"] == 0


@pytest.mark.parametrize("key_type,descending,radix,log2n", [("int32", False, 8, 27), ("float32", False, 8, 26), ("float32", True, 8, 27),
                                                             ("uint32", True, 4, 26), ("int32", True, 4, 27)])
def test_typed_keys_take_the_hybrid_form(gpu, key_type, descending, radix, log2n):
    """int32 / float32 / descending sorts of hybrid sizes (lsdsort_keys_device): the upfront read counts the sortable form of the
    keys, the first global pass stores it, the local stage's store turns it back.  Against torch.sort of the typed tensor
    (float32: random bit patterns with the NaNs replaced; -0 < +0 by IEEE total order is checked on the bits); the form that ran
    is read from the workspace; with pairs, the payload order is torch's stable order."""
    import torch

    n = (1 << log2n) + 1234
    gen = torch.Generator(device="cuda")
    gen.manual_seed(99 + log2n + radix)
    bits = torch.randint(-(1 << 31), (1 << 31) - 1, (n,), dtype=torch.int32, device="cuda", generator=gen)
    if key_type == "float32":
        # NaN / inf patterns (exponent 0xFF, 0.4 % of random bits) become finite values spread over 128 exponents (taken from their
        # own low mantissa bits): one replacement value, or one replacement exponent, would be buckets of twice and more the
        # average, which the planner rightly refuses
        bad = ((bits >> 23) & 0xFF) == 0xFF
        bits = torch.where(bad, (bits & ~(0xFF << 23)) | ((bits & 0x7F) << 23), bits)
        keys = bits.view(torch.float32).clone()
        keys[:4] = torch.tensor([0.0, -0.0, float("inf"), float("-inf")], device="cuda")
    else:
        keys = bits.clone()
    for pairs in (False, True):
        d = keys.clone()
        v = torch.arange(n, dtype=torch.int32, device="cuda") if pairs else None
        ws = gpu.alloc_workspace(n, radix, pairs)
        gpu.GPUSortTyped(d, key_type, descending, d_vals=v, r=radix, workspace=ws, check_fault=True)
        assert gpu.workspace_form(ws) == 1, (key_type, descending, radix, pairs)
        if key_type == "uint32":
            ref = _u64(keys)
            got = _u64(d)
        else:
            ref, got = keys, d
        expect = torch.sort(ref, descending=descending, stable=True)
        if key_type == "float32":
            # torch's sort treats -0 == +0; compare values, then the bit order of the zeros separately
            assert torch.equal(got, expect.values)
            zeros = got[got == 0].view(torch.int32)
            if zeros.numel() > 1:
                neg_first = bool((zeros[:-1] <= zeros[1:]).all()) if not descending else bool((zeros[:-1] >= zeros[1:]).all())
                # ascending: -0 (0x80000000 = int32 min) before +0 (0); descending: +0 before -0
                assert neg_first
        else:
            assert torch.equal(got, expect.values), (key_type, descending, radix, pairs)
        if pairs and key_type != "float32":
            assert torch.equal(v.to(torch.int64), expect.indices), "payloads keep the input order among equal keys"
        # the ordinary passes give the same
        gpu.set_hybrid(False)
        try:
            d2 = keys.clone()
            v2 = torch.arange(n, dtype=torch.int32, device="cuda") if pairs else None
            gpu.GPUSortTyped(d2, key_type, descending, d_vals=v2, r=radix, workspace=ws, check_fault=True)
            assert gpu.workspace_form(ws) == 0
            assert torch.equal(d2.view(torch.int32), d.view(torch.int32))
            if pairs:
                assert torch.equal(v2, v)
        finally:
            gpu.set_hybrid(True)


@pytest.mark.parametrize("prefix,radix,log2n", [(1, 8, 27), (1, 8, 28), (3, 8, 27), (3, 8, 26), (4, 8, 28), (7, 8, 27), (3, 4, 27), (8, 8, 26)])
def test_keys_that_share_a_prefix_take_the_hybrid_form(gpu, prefix, radix, log2n):
    """Keys that share their top bits -- values below 2^31, a shard of a range-partitioned array (what a rank holds after the
    MSB-bucket exchange of the multi-GPU sort).  With the prefix inside them such keys fill 2^(15 - prefix) buckets 2^prefix times
    too large; the device finds the prefix (65536-key sample), plans its buckets below it, checks it against every key in the
    upfront read -- up to seven bits; a constant top byte is the ordinary form's business (it skips a pass).  Through the plain
    entry and through lsdsort_u32_device_prefixed (whose argument is only validated); bit-exact against torch.sort.  One key
    outside the prefix, where the sample does not look: the upfront read notices, ordinary passes, same result."""
    import torch

    n = (1 << log2n) + 777
    gen = torch.Generator(device="cuda")
    gen.manual_seed(5 + prefix + log2n)
    base = torch.randint(-(1 << 31), (1 << 31) - 1, (n,), dtype=torch.int32, device="cuda", generator=gen)
    top = (0xA5 >> (8 - prefix)) << (32 - prefix)                       # some prefix value with its top bit set
    keys = _i32((_u64(base) >> prefix) | top)
    expect = torch.sort(_u64(keys)).values
    ws = gpu.alloc_workspace(n, radix)
    stream = torch.cuda.current_stream().cuda_stream
    L = gpu.lib()
    form = 1 if prefix <= 7 else 0

    for hint in (prefix, 0):
        d = keys.clone()
        assert L.lsdsort_u32_device_prefixed(d.data_ptr(), ws.data_ptr(), ws.numel(), n, radix, hint, stream) == 0
        assert gpu.workspace_form(ws) == form, (hint, gpu.workspace_form(ws))
        assert L.lsdsort_check_device(ws.data_ptr(), stream) == 0
        assert torch.equal(_u64(d), expect)
    d = keys.clone()
    gpu.GPULSDRadixSort(d, radix, workspace=ws, check_fault=True)
    assert gpu.workspace_form(ws) == form and torch.equal(_u64(d), expect)

    # one key outside the prefix, at the far end (not a sample position): the exact check refuses, the result is still right
    wrong = torch.cat([keys[:-1], _i32(torch.tensor([5], device="cuda", dtype=torch.int64))])
    d = wrong.clone()
    gpu.GPULSDRadixSort(d, radix, workspace=ws, check_fault=True)
    assert gpu.workspace_form(ws) == 0, "a key outside the sampled prefix: the ordinary passes"
    assert torch.equal(_u64(d), torch.sort(_u64(wrong)).values)
    assert L.lsdsort_u32_device_prefixed(d.data_ptr(), ws.data_ptr(), ws.numel(), n, radix, 9, stream) != 0


def test_non_negative_int32_keys_take_the_hybrid_form(gpu):
    """The commonest prefix there is: int32 keys that are all >= 0 (sortable form: top bit set in every key)."""
    import torch

    n = (1 << 27) + 99
    gen = torch.Generator(device="cuda")
    gen.manual_seed(17)
    keys = torch.randint(0, (1 << 31) - 1, (n,), dtype=torch.int32, device="cuda", generator=gen)
    ws = gpu.alloc_workspace(n, 8)
    d = keys.clone()
    gpu.GPUSortTyped(d, "int32", False, r=8, workspace=ws, check_fault=True)
    assert gpu.workspace_form(ws) == 1
    assert torch.equal(d, torch.sort(keys).values)
    d = keys.clone()
    gpu.GPUSortTyped(d, "int32", True, r=8, workspace=ws, check_fault=True)
    assert gpu.workspace_form(ws) == 1
    assert torch.equal(d, torch.sort(keys, descending=True).values)


@pytest.mark.parametrize("payloads,radix,log2n", [(2, 8, 26), (3, 8, 27), (3, 4, 26)])
def test_several_payload_arrays_take_the_hybrid_form(gpu, payloads, radix, log2n):
    """lsdsort_multi_u32_device at hybrid sizes (what a 64-bit-key or 64-bit-payload record sort is made of): the global passes
    carry every payload array, the local stage sorts the keys, composes its two digit passes' slots and sends each payload array
    through LDS once.  Keys with many duplicates inside the buckets (stability), payload e = a function of the input position:
    torch's stable sort says where every element must end up.  On keys the form refuses (half zeros): the ordinary passes."""
    import torch

    n = (1 << log2n) + 4321
    gen = torch.Generator(device="cuda")
    gen.manual_seed(31 + payloads + log2n)
    base = torch.randint(-(1 << 31), (1 << 31) - 1, (n,), dtype=torch.int32, device="cuda", generator=gen)
    idx = torch.arange(n, dtype=torch.int32, device="cuda")
    need = int(gpu.lib().lsdsort_workspace_bytes(n, radix, payloads))
    ws = torch.empty(need, dtype=torch.uint8, device="cuda")
    shapes = {
        "duplicates_in_buckets": (_i32(_u64(base) & 0xFFFC0F0F), 1),
        "half_zero": (torch.where(((_u64(base) >> 13) & 1) != 0, base, torch.zeros_like(base)), 0),
    }
    for name, (keys, form) in shapes.items():
        expect = torch.sort(_u64(keys), stable=True)
        d = keys.clone()
        pay = [(idx * (2 * e + 3) + e).to(torch.int32) for e in range(payloads)]          # int32 wrap-around is part of the function
        want = [p[expect.indices] for p in pay]
        gpu.GPUSortMulti(d, pay, r=radix, workspace=ws, check_fault=True)
        assert gpu.workspace_form(ws) == form, (name, gpu.workspace_form(ws))
        assert torch.equal(_u64(d), expect.values), name
        for e in range(payloads):
            assert torch.equal(pay[e], want[e]), (name, "payload", e)
        del expect, want, pay, d


def test_records_at_hybrid_size(gpu):
    """lsdsort_records_device, 64-bit keys with 64-bit payloads, 2^26 records: two multi-payload sorts, each in the hybrid form.
    Against torch's stable sort of the keys as unsigned 64-bit numbers (sign bit flipped); payload = input position."""
    import torch

    n = (1 << 26) + 999
    gen = torch.Generator(device="cuda")
    gen.manual_seed(64)
    k = torch.randint(-(1 << 63), (1 << 63) - 1, (n,), dtype=torch.int64, device="cuda", generator=gen)
    k[: n // 8] &= 0x0000FFFFFFFF0000                     # duplicates of high and low parts: stability across the two sorts
    v = torch.arange(n, dtype=torch.int64, device="cuda") * 3 + 1
    expect = torch.sort(k ^ (-(1 << 63)), stable=True)      # unsigned order
    kk, vv = k.clone(), v.clone()
    gpu.GPUSortWide(kk, vv, check_fault=True)
    assert torch.equal(kk, k[expect.indices])
    assert torch.equal(vv, v[expect.indices])
