"""GPU suite: the HIP path, through the C-ABI, against the oracle and the golden vectors.

Shaped like the reference's own self-checks: whole-sort output vs the CPU LSD result
(TestGPULSDRadixSort, LSDRadixSort.cu:1018), histograms vs BuildHistogramsCPU (.cu:785), scans
vs PrefixSum (.cu:364) -- plus everything the reference never tries (section 4 of SURVEY.md):
ragged sizes, duplicates, constant, sorted and reversed inputs, key/value pairs.
Bit-exact everywhere: integer work, tolerance zero.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ALGOS = {"onesweep": 0, "staged": 1}


def _sort_dev(lsd, keys, r, algo=0, vals=None):
    d = lsd.to_device(keys)
    dv = lsd.to_device(vals) if vals is not None else None
    lsd.GPULSDRadixSort(d, r, d_vals=dv, algorithm=algo, check_fault=True)
    if vals is None:
        return lsd.to_host(d)
    return lsd.to_host(d), lsd.to_host(dv)


@pytest.fixture(params=[-1, 0, 2], ids=["rank_auto", "rank_peer_mask", "rank_lds_add"])
def rank_form(request, gpu):
    """Every rank form the library can run (lsdsort_set_rank_method): the default picks the returning LDS add where
    the device probe passed; 0 forces the architecture-guaranteed peer-mask forms (LDS OR at 8 bits, ballots at
    4) that the library falls back to if the probe ever fails -- they must stay parity-green on their own.
    Under the forced returning add ("rank_lds_add") the one-launch sort of up to 16384 items is switched OFF, so that the chained
    kernels keep their coverage of tiny inputs in that rank form too; "rank_auto" runs the library as shipped (one launch there)."""
    gpu.set_rank_method(request.param)
    gpu.set_small_sort(request.param != 2)
    yield request.param
    gpu.set_rank_method(-1)
    gpu.set_small_sort(True)


# ----------------------------------------------------------------------------- golden vectors
@pytest.mark.parametrize("algo", list(ALGOS))
@pytest.mark.parametrize("r", [1, 2, 4, 8])
def test_golden_cases(gpu, golden, oracle_mod, algo, r, rank_form):
    for name in golden["case_names"]:
        keys, expect = golden[f"in__{name}"], golden[f"sorted__{name}"]
        got = _sort_dev(gpu, keys, r, ALGOS[algo])
        bad = oracle_mod.first_mismatch(got, expect)
        assert bad == expect.size, f"{name} r={r} {algo}: first mismatch at {bad}"


def test_golden_pairs(gpu, golden):
    for r in (4, 8):
        for algo in ALGOS.values():
            k, v = _sort_dev(gpu, golden["pairs_keys"], r, algo, golden["pairs_vals"])
            assert np.array_equal(k, golden["pairs_sorted_keys"]), (r, algo)
            assert np.array_equal(v, golden["pairs_sorted_vals"]), (r, algo)


@pytest.mark.parametrize("r", [4, 8])
def test_pass_states_match_reference_passes(gpu, golden, r):
    """After pass g the array equals what the reference's LSDRadixSortPass leaves (.cu:25-54):
    the stage entries chained by hand, one pass at a time."""
    cur = gpu.to_device(golden["passes_in"])
    states = golden[f"passes_r{r}"]
    for g in range(32 // r):
        h = gpu.BuildHistograms(cur, r, g)
        _, glob = gpu.BuildOffsets(h, r)
        cur = gpu.RankScatter(cur, glob, r, g)
        assert np.array_equal(gpu.to_host(cur), states[g]), (r, g)


# ----------------------------------------------------------------------------- host-pointer entries
def test_host_pointer_sort(gpu, golden, oracle_mod):
    for name in ("uniform_16384_seed0", "uniform_12345_seed1", "n1", "n2", "allmax_4099"):
        keys = golden[f"in__{name}"].copy()
        assert gpu.sort(keys) is keys
        assert np.array_equal(keys, golden[f"sorted__{name}"]), name
    keys = oracle_mod.mt19937_keys((1 << 20) + 3, 21)       # BASELINE configs[0] size, through the GPU
    expect = oracle_mod.std_sort(keys)
    assert np.array_equal(gpu.sort(keys.copy(), radix_bits=4), expect)
    empty = np.zeros(0, dtype=np.uint32)
    assert gpu.sort(empty).size == 0


def test_host_pointer_sort_spans_chunks(gpu, oracle_mod):
    """The host entry feeds the device in 64 MiB chunks (stage 1 runs behind each): a ragged size spanning three chunks,
    keys and pairs, against std::sort / std::stable_sort, twice (the second call reuses the cached device buffers),
    then a smaller and a larger call on the same cache."""
    n = 2 * (1 << 24) + (1 << 22) + 4097 + 3
    keys = oracle_mod.mt19937_keys(n, 77)
    expect = oracle_mod.std_sort(keys)
    for _ in range(2):
        got = keys.copy()
        gpu.sort(got)
        assert np.array_equal(got, expect)
    for r in (4, 2):
        small = keys[: (1 << 20) + 5].copy()
        assert np.array_equal(gpu.sort(small, radix_bits=r), np.sort(keys[: (1 << 20) + 5])), r
    dup = (keys % 100003).astype(np.uint32)
    vals = np.arange(n, dtype=np.uint32)
    ek, ev = oracle_mod.std_stable_sort_pairs(dup, vals)
    k, v = dup.copy(), vals.copy()
    gpu.sort_pairs(k, v)
    assert np.array_equal(k, ek) and np.array_equal(v, ev)
    assert gpu.lib().lsdsort_release_host_cache() == 0
    got = keys[:12345].copy()
    assert np.array_equal(gpu.sort(got), np.sort(keys[:12345]))          # after a release the cache is rebuilt


def test_host_pointer_pairs(gpu, golden):
    k, v = golden["pairs_keys"].copy(), golden["pairs_vals"].copy()
    gpu.sort_pairs(k, v)
    assert np.array_equal(k, golden["pairs_sorted_keys"]) and np.array_equal(v, golden["pairs_sorted_vals"])


# ----------------------------------------------------------------------------- against the oracle, seeded
@pytest.mark.parametrize("algo", list(ALGOS))
@pytest.mark.parametrize("r", [4, 8])
@pytest.mark.parametrize("n,seed", [(8192, 1), (8193, 2), (3 * 8192 - 1, 3), ((1 << 20) + 17, 0), ((1 << 22) + 4099, 5)])
def test_uniform_vs_oracle(gpu, oracle_mod, algo, r, n, seed):
    keys = oracle_mod.mt19937_keys(n, seed)
    got = _sort_dev(gpu, keys, r, ALGOS[algo])
    expect = oracle_mod.lsd_sort(keys, r)        # == std::sort, as the reference asserts (.cu:120)
    bad = oracle_mod.first_mismatch(got, expect)
    assert bad == n, f"first mismatch at {bad}: got {got[bad]} expect {expect[bad]}"


@pytest.mark.parametrize("r", [1, 2])
def test_narrow_radix_vs_oracle(gpu, oracle_mod, r):
    keys = oracle_mod.mt19937_keys((1 << 18) + 77, 12)
    for algo in ALGOS.values():
        assert np.array_equal(_sort_dev(gpu, keys, r, algo), oracle_mod.std_sort(keys))
    for n in ((1 << 21) + 5, (1 << 23) + 9):               # the larger tiles narrow digits use from 2^21 / 2^23 keys
        keys = oracle_mod.mt19937_keys(n, 13)
        assert np.array_equal(_sort_dev(gpu, keys, r), np.sort(keys)), (r, n)


@pytest.mark.parametrize("r", [4, 8])
def test_distributions(gpu, oracle_mod, r, rank_form):
    n = (1 << 21) + 1234
    base = oracle_mod.mt19937_keys(n, 33)
    cases = {
        "allequal": np.full(n, 0x12345678, dtype=np.uint32),
        "allzero": np.zeros(n, dtype=np.uint32),
        "allmax": np.full(n, 0xFFFFFFFF, dtype=np.uint32),          # indistinguishable from tail padding
        "sorted": np.sort(base),
        "reverse": np.sort(base)[::-1].copy(),
        "dup3": (base % 3).astype(np.uint32) * np.uint32(0x55555555),
        "low8": (base & 0xFF).astype(np.uint32),
        "high8": (base & 0xFF000000).astype(np.uint32),
        "mid": (base & 0x00FFFF00).astype(np.uint32),
        "two_values_far": np.where(base & 1, np.uint32(0xFFFFFFFF), np.uint32(0)).astype(np.uint32),
        # heavy values: part of every wave row shares a digit (hand-counted in stage 1, ranked from scalar counts in the passes)
        "half_zero": np.where((base >> np.uint32(13)) & np.uint32(1), base, np.uint32(0)).astype(np.uint32),
        "ninety_pct_one_value": np.where((base % np.uint32(10)) != 0, np.uint32(0x80000001), base).astype(np.uint32),
        "quarter_max": np.where((base & np.uint32(3)) == 0, np.uint32(0xFFFFFFFF), base).astype(np.uint32),
        "four_values_per_digit": (base & np.uint32(0x03030303)).astype(np.uint32),
        "three_heavy_values": np.where(base % np.uint32(7) < 2, np.uint32(5), np.where(base % np.uint32(7) < 4, np.uint32(0x05000005),
                                       np.where(base % np.uint32(7) < 6, np.uint32(0xFF0000FF), base))).astype(np.uint32),
        "runs_then_random": np.concatenate([np.repeat(base[: n // 600], 300)[: n // 2], base[n // 2:]]).astype(np.uint32)[:n],
    }
    for name, keys in cases.items():
        got = _sort_dev(gpu, keys, r)
        assert np.array_equal(got, np.sort(keys)), (name, r)


@pytest.mark.parametrize("r", [8, 4, 2])
def test_dead_passes_are_skipped_with_identical_results(gpu, oracle_mod, r):
    """Passes whose digit is the same for every key are skipped on the device (plan written by stage 2): every number of
    skipped passes (0 .. all), odd numbers (the copy back), dead digits below, between and above live ones, keys and
    stable pairs, sizes around a tile; the same input with skipping switched off gives the same output."""
    rng = np.random.default_rng(77 + r)
    masks = [0xFFFFFFFF, 0x0000FFFF, 0x00FFFFFF, 0x000000FF, 0xFF000000, 0x00FF0000, 0xFF0000FF, 0x0000FF00, 0x00000000,
             0x000FFFFF, 0x0F0F0F0F, 0xFFFF0000, 0x00FFFF00]
    for case, mask in enumerate(masks):
        n = int(rng.choice([1, 5, 4097, 32768, 32769, (1 << 20) + 3, (1 << 23) + 77])) if case % 3 else (1 << 21) + 11
        keys = (oracle_mod.mt19937_keys(n, 100 + case) & np.uint32(mask)) | np.uint32(0x12345678 & ~mask & 0xFFFFFFFF)
        keys = keys.astype(np.uint32)
        vals = np.arange(n, dtype=np.uint32)
        ek, ev = oracle_mod.std_stable_sort_pairs(keys, vals)
        for skipping in (True, False):
            gpu.set_pass_skipping(skipping)
            try:
                assert np.array_equal(_sort_dev(gpu, keys, r), ek), (hex(mask), n, r, skipping)
                k, v = _sort_dev(gpu, keys, r, 0, vals)
                assert np.array_equal(k, ek) and np.array_equal(v, ev), (hex(mask), n, r, skipping, "pairs")
            finally:
                gpu.set_pass_skipping(True)


def test_randomised_shapes_of_input(gpu, oracle_mod):
    """Seeded sweep over sizes, radix widths and key shapes that steer the low-entropy paths of stage 1
    (wave-uniform fields, edge groups, few-valued digits) and of the rank phase: random bit masks, sorted /
    nearly sorted / run-structured keys, ragged sizes around tile and chunk boundaries.  Against numpy."""
    rng = np.random.default_rng(20260)
    sizes = [1, 2, 63, 64, 65, 255, 1023, 1025, 2047, 2049, 4095, 4097, 8191, 16385, 32769, 65537, (1 << 19) - 3, (1 << 19) + 3,
             (1 << 21) + 11, (1 << 22) - 1]
    for case in range(64):
        n = int(sizes[case % len(sizes)] if case < 40 else rng.integers(1, 1 << 21))
        r = (8, 4, 8, 2, 8, 4, 8, 1)[case % 8]
        if r <= 2 and n > (1 << 18):
            r = 4
        base = rng.integers(0, 1 << 32, size=n, dtype=np.uint64).astype(np.uint32)
        kind = case % 7
        if kind == 1:
            base = np.sort(base)
        elif kind == 2:
            base &= np.uint32(rng.integers(0, 1 << 32))                       # random dead bits
        elif kind == 3:
            base = np.sort(base)
            swap = rng.integers(0, n, size=max(1, n // 100))
            base[swap] = rng.integers(0, 1 << 32, size=swap.size, dtype=np.uint64).astype(np.uint32)   # nearly sorted
        elif kind == 4:
            base = np.repeat(base[: max(1, n // 300)], 300)[:n] if n >= 300 else base                 # runs of equal keys
        elif kind == 5:
            base = (base % np.uint32(rng.integers(1, 40))) * np.uint32(0x01010101)                    # few values per byte
        elif kind == 6:
            base = np.sort(base)[::-1].copy()
        got = _sort_dev(gpu, base, r)
        assert np.array_equal(got, np.sort(base)), (case, n, r, kind)


@pytest.mark.parametrize("r", [4, 8])
def test_pairs_vs_stable_sort(gpu, oracle_mod, r, rank_form):
    n = (1 << 20) + 9
    keys = (oracle_mod.mt19937_keys(n, 41) % 1021).astype(np.uint32) * np.uint32(0x00400801)   # heavy duplicates
    vals = np.arange(n, dtype=np.uint32)
    ek, ev = oracle_mod.std_stable_sort_pairs(keys, vals)
    for algo in ALGOS.values():
        k, v = _sort_dev(gpu, keys, r, algo, vals)
        assert np.array_equal(k, ek), (r, algo)
        assert np.array_equal(v, ev), (r, algo, "payload order among equal keys = stability")
    # all keys equal: the payload must come back untouched
    k, v = _sort_dev(gpu, np.full(n, 7, dtype=np.uint32), r, 0, vals)
    assert np.array_equal(v, vals)
    # heavy values (ranked from running scalar counts, not by LDS atomics): the order among equal keys is the rank itself
    base = oracle_mod.mt19937_keys(n, 43)
    for name, hk in (("half_zero", np.where((base >> np.uint32(9)) & np.uint32(1), base & np.uint32(0x00FF00FF), np.uint32(0))),
                     ("ninety_pct", np.where((base % np.uint32(10)) != 0, np.uint32(0x80000001), base & np.uint32(0xFFFF))),
                     ("two_values", np.where(base & np.uint32(1 << 17), np.uint32(0x11111111), np.uint32(0xEEEEEEEE))),
                     ("runs", np.repeat(base[: n // 200 + 1] & np.uint32(0x0F0F0F0F), 200)[:n])):
        hk = hk.astype(np.uint32)
        ek, ev = oracle_mod.std_stable_sort_pairs(hk, vals)
        k, v = _sort_dev(gpu, hk, r, 0, vals)
        assert np.array_equal(k, ek) and np.array_equal(v, ev), (name, r)


@pytest.mark.parametrize("r", [8, 4, 1])
@pytest.mark.parametrize("skip", [1, 2, 3])
def test_keys_not_16_byte_aligned(gpu, oracle_mod, r, skip):
    """A slice of a larger buffer (keys[skip:]): the upfront histogram's 16-byte loads do not apply, everything goes
    through its scalar loop (grid-wide); the passes load 4 bytes per lane anyway."""
    import torch

    n = (1 << 20) + 4099
    keys = oracle_mod.mt19937_keys(n + skip, 77 + skip)
    whole = gpu.to_device(keys)
    part = whole[skip:]
    assert part.data_ptr() % 16 == 4 * skip and part.is_contiguous()
    gpu.GPULSDRadixSort(part, r, check_fault=True)
    got = gpu.to_host(whole)
    assert np.array_equal(got[:skip], keys[:skip]), "keys in front of the slice were touched"
    assert np.array_equal(got[skip:], np.sort(keys[skip:]))
    # typed keys and the multi-GPU partition take the same histogram kernels
    f = torch.from_numpy(keys.view(np.float32).copy()).cuda()[skip:]
    f[torch.isnan(f)] = 0.0
    ref = np.sort(f.cpu().numpy())
    gpu.GPUSortTyped(f, "float32", r=8, check_fault=True)
    assert np.array_equal(f.cpu().numpy().view(np.uint32), ref.view(np.uint32))
    out, counts = gpu.MSBPartition(gpu.to_device(keys)[skip:], 2)
    assert [int(c) for c in counts.cpu()] == [int(np.sum((keys[skip:] >> 30) == b)) for b in range(4)]
    out, counts = gpu.SplitterPartition(gpu.to_device(keys)[skip:], [1 << 30, 1 << 31, 3 << 30])
    assert [int(c) for c in counts.cpu()] == [int(np.sum((keys[skip:] >> 30) == b)) for b in range(4)]


def test_temporary_workspace_belongs_to_the_sorts_stream(gpu, oracle_mod):
    """No workspace given, explicit side stream: the temporary workspace is allocated under that stream,
    so work queued on the current stream right after the call cannot be handed its block while the passes run."""
    import torch

    n = (1 << 22) + 17
    keys = oracle_mod.mt19937_keys(n, 91)
    side = torch.cuda.Stream()
    for rep in range(4):
        d = gpu.to_device(keys)
        torch.cuda.synchronize()
        gpu.GPULSDRadixSort(d, 8, stream=side)                 # returns while the passes are still running
        junk = [torch.full((1 << 22,), 0x5A5A5A5A, dtype=torch.int32, device="cuda") for _ in range(8)]   # current stream
        side.synchronize()
        torch.cuda.synchronize()
        assert np.array_equal(gpu.to_host(d), np.sort(keys)), rep
        del junk


def test_tile_configs(gpu, oracle_mod):
    keys = oracle_mod.mt19937_keys((1 << 20) + 5, 8)
    expect = np.sort(keys)
    vals = np.arange(keys.size, dtype=np.uint32)
    ek, ev = oracle_mod.std_stable_sort_pairs(keys, vals)
    try:
        for r, count in ((8, 6), (4, 6)):
            for cfg in range(count):
                gpu.set_tile_config(r, cfg)
                for algo in ALGOS.values():
                    assert np.array_equal(_sort_dev(gpu, keys, r, algo), expect), (r, cfg, algo)
                k, v = _sort_dev(gpu, keys, r, 0, vals)
                assert np.array_equal(k, ek) and np.array_equal(v, ev), (r, cfg)
    finally:
        gpu.set_tile_config(8, -1)
        gpu.set_tile_config(4, -1)


def test_workspace_reuse_and_idempotence(gpu, oracle_mod):
    """One workspace across sorts of different data (the tile-status words are recycled), and
    sorting a sorted array changes nothing."""
    import torch

    n = (1 << 20) + 1
    ws = gpu.alloc_workspace(n, 8)
    ws.fill_(0xA5)                                     # garbage workspace must not matter
    for seed in (1, 2, 3):
        keys = oracle_mod.mt19937_keys(n, seed)
        d = gpu.to_device(keys)
        gpu.GPULSDRadixSort(d, 8, workspace=ws, check_fault=True)
        assert np.array_equal(gpu.to_host(d), np.sort(keys)), seed
        before = d.clone()
        gpu.GPULSDRadixSort(d, 8, workspace=ws, check_fault=True)
        assert torch.equal(d, before)


@pytest.mark.parametrize("n", [(1 << 19) - 1, 1 << 19, (1 << 21) - 1, 1 << 21, (1 << 23) - 1, 1 << 23, (1 << 23) + 12345])
def test_default_shape_threshold(gpu, oracle_mod, n):
    """The library picks its tile by size class (4096 / 8192 / 16384 / 32768 keys, switching at 2^19,
    2^21 and 2^23 keys): both sides of every switch, keys (r = 8 and 4) and pairs, against std::sort /
    std::stable_sort."""
    keys = oracle_mod.mt19937_keys(n, 11)
    assert np.array_equal(_sort_dev(gpu, keys, 8), np.sort(keys))
    assert np.array_equal(_sort_dev(gpu, keys, 4), np.sort(keys))
    vals = np.arange(n, dtype=np.uint32)
    ek, ev = oracle_mod.std_stable_sort_pairs(keys >> np.uint32(12), vals)
    k, v = _sort_dev(gpu, keys >> np.uint32(12), 8, 0, vals)
    assert np.array_equal(k, ek) and np.array_equal(v, ev)


def test_workspace_serves_smaller_sorts(gpu, oracle_mod):
    """A workspace sized for n keys must do for every n' <= n, whichever tile the library picks."""
    n = (1 << 23) + 5
    ws = gpu.alloc_workspace(n, 8)
    for m in (n, (1 << 23) - 7, (1 << 21) + 3, (1 << 21) - 1, (1 << 19) + 1, (1 << 19) - 1, 1000, 1):
        keys = oracle_mod.mt19937_keys(m, m & 0xFF)
        d = gpu.to_device(keys)
        gpu.GPULSDRadixSort(d, 8, workspace=ws, check_fault=True)
        assert np.array_equal(gpu.to_host(d), np.sort(keys)), m
    points = [1, 4095, 4097, (1 << 19) - 1, 1 << 19, (1 << 19) + 1, (1 << 21) - 1, 1 << 21, (1 << 22), (1 << 23) - 1, 1 << 23,
              (1 << 23) + 1, 1 << 24]
    for r, pairs in ((8, 0), (8, 1), (4, 0), (2, 0)):
        sizes = [gpu.lib().lsdsort_workspace_bytes(m, r, pairs) for m in points]
        assert sizes == sorted(sizes), (r, pairs)


@pytest.mark.parametrize("r,pairs", [(8, False), (4, False), (8, True)])
def test_hip_graph_capture_and_replay(gpu, oracle_mod, r, pairs):
    """The device entry allocates nothing and never synchronises, so a sort can be captured into a HIP
    graph once and replayed on new data in the same buffers (include/lsdsort.h)."""
    import torch

    n = (1 << 18) + 77
    gpu.lib().lsdsort_prepare_device()
    ws = gpu.alloc_workspace(n, r, pairs)
    static_k = gpu.to_device(oracle_mod.mt19937_keys(n, 1))
    static_v = torch.arange(n, dtype=torch.int32, device="cuda") if pairs else None
    gpu.GPULSDRadixSort(static_k, r, d_vals=static_v, workspace=ws)       # warm-up outside the capture
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        gpu.GPULSDRadixSort(static_k, r, d_vals=static_v, workspace=ws)
    for seed in (2, 3):
        keys = oracle_mod.mt19937_keys(n, seed) >> np.uint32(8 if pairs else 0)
        static_k.copy_(gpu.to_device(keys))
        if pairs:
            static_v.copy_(torch.arange(n, dtype=torch.int32, device="cuda"))
        graph.replay()
        torch.cuda.synchronize()
        if pairs:
            ek, ev = oracle_mod.std_stable_sort_pairs(keys, np.arange(n, dtype=np.uint32))
            assert np.array_equal(gpu.to_host(static_k), ek) and np.array_equal(gpu.to_host(static_v), ev), seed
        else:
            assert np.array_equal(gpu.to_host(static_k), np.sort(keys)), seed
    assert gpu.lib().lsdsort_check_device(ws.data_ptr(), None) == 0


def test_two_host_threads_two_streams(gpu, oracle_mod):
    """Two host threads sorting at once, each on its own stream with its own workspace (ctypes drops the
    GIL inside the library): the library keeps no per-call state outside the workspace."""
    import threading
    import torch

    failures = []

    def worker(seed):
        try:
            torch.cuda.set_device(0)
            stream = torch.cuda.Stream()
            with torch.cuda.stream(stream):
                for i in range(12):
                    n = (1 << 16) * (1 + (seed + i) % 5) + 13 * i
                    keys = oracle_mod.mt19937_keys(n, 100 * seed + i)
                    r = 8 if i % 3 else 4
                    d = gpu.to_device(keys)
                    ws = gpu.alloc_workspace(n, r)
                    gpu.GPULSDRadixSort(d, r, workspace=ws, stream=stream, check_fault=True)
                    stream.synchronize()
                    if not np.array_equal(gpu.to_host(d), np.sort(keys)):
                        failures.append((seed, i, n, r))
        except Exception as exc:       # noqa: BLE001 - reported below
            failures.append((seed, repr(exc)))

    threads = [threading.Thread(target=worker, args=(s,)) for s in (1, 2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not failures, failures


def test_device_entry_errors(gpu):
    import torch

    from lsdradixsort_amd import errors

    L = gpu.lib()
    d = gpu.to_device(np.arange(5000, dtype=np.uint32))
    ws = gpu.alloc_workspace(5000, 8)
    stream = torch.cuda.current_stream().cuda_stream
    assert L.lsdsort_u32_device(d.data_ptr(), ws.data_ptr(), 16, 5000, 8, stream) == errors.LSDSORT_ERR_WORKSPACE
    assert L.lsdsort_u32_device(d.data_ptr(), ws.data_ptr() + 4, ws.numel() - 4, 5000, 8, stream) == errors.LSDSORT_ERR_WORKSPACE
    assert L.lsdsort_u32_device(d.data_ptr(), None, 0, 5000, 8, stream) == errors.LSDSORT_ERR_WORKSPACE
    assert L.lsdsort_u32_device(None, ws.data_ptr(), ws.numel(), 5000, 8, stream) == errors.LSDSORT_ERR_INVALID_ARG
    assert L.lsdsort_u32_device(d.data_ptr(), ws.data_ptr(), ws.numel(), 5000, 3, stream) == errors.LSDSORT_ERR_INVALID_ARG
    assert L.lsdsort_u32_device(None, None, 0, 0, 8, stream) == errors.LSDSORT_OK
    torch.cuda.synchronize()
    assert np.array_equal(gpu.to_host(d), np.arange(5000, dtype=np.uint32))


# ----------------------------------------------------------------------------- stage-level parity
@pytest.mark.parametrize("r,bg", [(8, 0), (8, 3), (4, 5), (2, 9), (1, 18)])
def test_stage_histograms(gpu, oracle_mod, r, bg):
    """a2: h[tile][digit] vs the restated BuildHistogramsCPU (.cu:643-658)."""
    keys = oracle_mod.mt19937_keys(5 * gpu.tile_keys(r) + 321, 50 + r)
    h = gpu.to_host(gpu.BuildHistograms(gpu.to_device(keys), r, bg).reshape(-1)).reshape(-1, 1 << r)
    assert np.array_equal(h, oracle_mod.tile_histograms(keys, gpu.tile_keys(r), r, bg))


def test_stage_histograms_reference_vector(gpu, golden):
    """The reference's own BuildHistogramsCPU output, regrouped to our tile size."""
    keys = golden["hist_in"]                                   # 8192 keys = 8 reference blocks of 1024
    tile = gpu.tile_keys(8)
    assert tile % 1024 == 0
    per_ref_tile = tile // 1024
    blocks = golden["hist_block1024_r8_bg1"].astype(np.uint64)  # [8][256]
    tiles = (keys.size + tile - 1) // tile
    ref = np.stack([blocks[t * per_ref_tile:(t + 1) * per_ref_tile].sum(axis=0) for t in range(tiles)])
    h = gpu.to_host(gpu.BuildHistograms(gpu.to_device(keys), 8, 1).reshape(-1)).reshape(-1, 256)
    assert np.array_equal(h.astype(np.uint64), ref)


@pytest.mark.parametrize("r,tiles", [(8, 1), (8, 63), (8, 64), (8, 1000), (4, 129), (2, 4097), (1, 70000)])
def test_stage_offsets(gpu, oracle_mod, r, tiles):
    """a3-a6: local (per-tile exclusive scan, .cu:869) and global (digit-major exclusive scan,
    .cu:877-895) tables vs the restatement; counts are arbitrary, not tied to a key array."""
    rng = np.random.default_rng(r * 1000 + tiles)
    hist = rng.integers(0, 9000, size=(tiles, 1 << r), dtype=np.uint32)
    d_local, d_global = gpu.BuildOffsets(gpu.to_device(hist.reshape(-1)).reshape(tiles, 1 << r), r)
    assert np.array_equal(gpu.to_host(d_local.reshape(-1)).reshape(hist.shape), oracle_mod.local_offsets(hist, r))
    assert np.array_equal(gpu.to_host(d_global.reshape(-1)).reshape(hist.shape), oracle_mod.global_offsets(hist, r))


def test_stage_scan_reference_vector(gpu, golden):
    """PrefixSum known answer {3,1,4,1,5} -> {0,3,4,8,9} (.cu:128-139) through the offset stage:
    one tile of 2^3... the scan entry works on 2^r columns, so pad to r=8 with zeros."""
    row = np.zeros((1, 256), dtype=np.uint32)
    row[0, :5] = golden["scan_kat_in"]
    d_local, _ = gpu.BuildOffsets(gpu.to_device(row.reshape(-1)).reshape(1, 256), 8)
    assert list(gpu.to_host(d_local.reshape(-1))[:5]) == list(golden["scan_kat_out"])


@pytest.mark.parametrize("r,bg", [(8, 1), (4, 2), (2, 0), (1, 31)])
def test_stage_rank_scatter(gpu, oracle_mod, r, bg):
    """a7: dst = rank - local[d] + global[d] (.cu:833) vs the restatement, and vs one CPU pass."""
    tile = gpu.tile_keys(r)
    keys = oracle_mod.mt19937_keys(7 * tile + 11, 60 + r)
    h = oracle_mod.tile_histograms(keys, tile, r, bg)
    local, glob = oracle_mod.local_offsets(h, r), oracle_mod.global_offsets(h, r)
    expect = oracle_mod.rank_scatter(keys, local, glob, tile, r, bg)
    assert np.array_equal(expect, oracle_mod.lsd_pass(keys, r, bg))
    d_glob = gpu.to_device(glob.reshape(-1)).reshape(glob.shape)
    got = gpu.to_host(gpu.RankScatter(gpu.to_device(keys), d_glob, r, bg))
    assert np.array_equal(got, expect)


@pytest.mark.parametrize("r", [1, 2, 4, 8])
def test_stage_digit_histograms(gpu, oracle_mod, r):
    for n, seed in ((1, 1), (1023, 2), ((1 << 20) + 13, 3)):
        keys = oracle_mod.mt19937_keys(n, seed)
        got = gpu.to_host(gpu.DigitHistograms(gpu.to_device(keys), r).reshape(-1)).reshape(32 // r, 1 << r)
        assert np.array_equal(got.astype(np.uint64), oracle_mod.digit_histograms(keys, r)), (r, n)
    const = np.full(300001, 0xA5A5A5A5, dtype=np.uint32)      # wave-uniform digits take the aggregated path
    got = gpu.to_host(gpu.DigitHistograms(gpu.to_device(const), r).reshape(-1)).reshape(32 // r, 1 << r)
    assert np.array_equal(got.astype(np.uint64), oracle_mod.digit_histograms(const, r))


@pytest.mark.parametrize("msb_bits", [0, 1, 2, 3])
def test_msb_partition(gpu, oracle_mod, msb_bits):
    for n, seed in ((1, 1), (4097, 2), ((1 << 20) + 7, 3)):
        keys = oracle_mod.mt19937_keys(n, seed)
        out, counts = gpu.MSBPartition(gpu.to_device(keys), msb_bits)
        eo, ec = oracle_mod.msb_partition(keys, msb_bits)
        assert np.array_equal(counts.cpu().numpy().astype(np.uint64), ec), (msb_bits, n)
        assert np.array_equal(gpu.to_host(out), eo), (msb_bits, n)


def test_rccl_exchange_path_world_of_one(gpu, oracle_mod):
    """The multi-GPU driver's collective calls (all_gather_into_tensor of the count matrix,
    all_to_all_single with split sizes, RCCL = backend "nccl") on the one GPU a test box has:
    a process group of one rank, exchange forced.  The N > 1 logic is covered by the gloo tests."""
    import os
    import socket
    import torch
    import torch.distributed as dist
    from lsdradixsort_amd.dist import HipBackend, distributed_sort

    if dist.is_initialized():
        pytest.skip("a process group already exists in this process")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    try:
        keys = oracle_mod.mt19937_keys((1 << 20) + 3, 21)
        res = distributed_sort(gpu.to_device(keys), backend=HipBackend(8), exchange_always=True)
        torch.cuda.synchronize()
        assert res.global_offset == 0 and int(res.counts.sum()) == keys.size
        assert np.array_equal(gpu.to_host(res.keys), np.sort(keys))
    finally:
        dist.destroy_process_group()


def test_cpp_sharded_step_world_of_one(gpu, oracle_mod):
    """The product's multi-GPU step (ShardedSorter -> lsdsort_sharded_u32_device, C++ over RCCL) as a world of one:
    ncclGetUniqueId / ncclCommInitRank / ncclAllGather on the side stream / the (empty) grouped exchange / own-bucket
    copy / local sort, several steps on one communicator, ragged and empty shards, and the capacity retry."""
    import torch
    from lsdradixsort_amd.dist import ShardedSorter

    sorter = ShardedSorter(8)
    try:
        assert sorter.world == 1 and sorter.rank == 0
        for n, seed in (((1 << 20) + 3, 5), ((1 << 22) + 12345, 6), (1, 7), (0, 8), ((1 << 23) + 1, 9)):
            keys = oracle_mod.mt19937_keys(n, seed)
            d = gpu.to_device(keys)
            res = sorter.sort(d)
            torch.cuda.synchronize()
            assert res.global_offset == 0 and int(res.counts.sum()) == n and res.keys.numel() == n
            assert np.array_equal(gpu.to_host(res.keys), np.sort(keys)), n
            assert np.array_equal(gpu.to_host(d), keys), "the input shard is left untouched"
            assert sorter.check_fault() == 0
        # a capacity that is too small: every rank hears LSDSORT_ERR_CAPACITY before the exchange; the face retries once
        keys = oracle_mod.mt19937_keys((1 << 20) + 9, 10)
        res = sorter.sort(gpu.to_device(keys), capacity=1000)
        assert np.array_equal(gpu.to_host(res.keys), np.sort(keys))
    finally:
        sorter.close()


@pytest.mark.parametrize("nsplit", [0, 1, 3, 7])
def test_splitter_partition(gpu, oracle_mod, nsplit):
    """Partition by value (multi-GPU step 1 for skewed keys): bucket = number of splitters <= key, stable,
    against numpy's searchsorted + stable argsort; ragged sizes, keys equal to splitters, empty buckets."""
    rng = np.random.default_rng(5 + nsplit)
    for n, seed in ((1, 1), (4097, 2), ((1 << 20) + 7, 3), ((1 << 23) + 1001, 4)):
        keys = oracle_mod.mt19937_keys(n, seed)
        if seed == 3:
            keys = (keys >> np.uint32(5)).astype(np.uint32)          # skewed: top five bits clear
        splitters = np.sort(rng.choice(keys, size=nsplit, replace=True)) if nsplit else np.zeros(0, dtype=np.uint32)
        if nsplit >= 3 and seed == 2:
            splitters[1] = splitters[0]                               # an empty bucket
        out, counts = gpu.SplitterPartition(gpu.to_device(keys), [int(x) for x in splitters])
        bucket = np.searchsorted(splitters, keys, side="right")
        order = np.argsort(bucket, kind="stable")
        assert np.array_equal(counts.cpu().numpy(), np.bincount(bucket, minlength=nsplit + 1)), (nsplit, n)
        assert np.array_equal(gpu.to_host(out), keys[order]), (nsplit, n)
    with pytest.raises(Exception):
        gpu.SplitterPartition(gpu.to_device(keys), [5, 4, 9])        # not ascending


@pytest.mark.parametrize("nthr", [1, 3, 7])
def test_threshold_partition(gpu, oracle_mod, nthr):
    """The form of the partition the C++ step's splitter rule uses: 64-bit thresholds, 2^32 = above every key (a 32-bit
    splitter cannot say that), 0 = below every key; keys equal to 0xFFFFFFFF and to thresholds; against numpy."""
    rng = np.random.default_rng(50 + nthr)
    for case, (n, seed) in enumerate(((3, 1), (4097, 2), ((1 << 20) + 7, 3), ((1 << 22) + 1001, 4))):
        keys = oracle_mod.mt19937_keys(n, seed)
        keys[rng.integers(0, n, size=max(1, n // 8))] = 0xFFFFFFFF
        th = sorted(int(x) for x in rng.choice(keys, size=nthr, replace=True))
        if case == 1:
            th[-1] = 1 << 32                                          # the last bucket stays empty, 0xFFFFFFFF keys included
            if nthr >= 3:
                th[-2] = 1 << 32
                th[0] = 0                                             # and the first one too
        if case == 2:
            th = [1 << 32] * nthr                                     # one bucket holds everything
        if case == 3:
            th[-1] = 0xFFFFFFFF                                       # the all-ones keys alone in the last bucket
        out, counts = gpu.ThresholdPartition(gpu.to_device(keys), th)
        bucket = np.zeros(n, dtype=np.int64)
        for t in th:
            if t < (1 << 32):
                bucket += keys >= np.uint32(t)
        order = np.argsort(bucket, kind="stable")
        assert np.array_equal(counts.cpu().numpy(), np.bincount(bucket, minlength=nthr + 1)), (nthr, n, th)
        assert np.array_equal(gpu.to_host(out), keys[order]), (nthr, n, th)
    if nthr >= 3:
        with pytest.raises(Exception):
            gpu.ThresholdPartition(gpu.to_device(keys), [9, 4, 9] + [9] * (nthr - 3))                             # not ascending
    with pytest.raises(Exception):
        gpu.ThresholdPartition(gpu.to_device(keys), [(1 << 32) + 1] * nthr)                                       # out of range


def test_cpp_sharded_step_splitter_rule_world_of_one(gpu, oracle_mod):
    """lsdsort_sharded_u32_device_ex with LSDSORT_PARTITION_SPLITTERS as a world of one: the sample kernel, its
    ncclAllGather and host wait, the threshold arithmetic (no thresholds for one rank) and the rest of the step, on
    skewed, constant, tiny and empty shards."""
    import torch
    from lsdradixsort_amd.dist import ShardedSorter

    sorter = ShardedSorter(8, partition="splitters")
    try:
        for n, seed in (((1 << 20) + 3, 5), (1, 7), (0, 8), (300, 9), ((1 << 22) + 77, 10)):
            keys = oracle_mod.mt19937_keys(n, seed)
            if seed == 5:
                keys = (keys >> np.uint32(12)).astype(np.uint32)      # top bits clear
            if seed == 10:
                keys[:] = 0xFFFFFFFF
            d = gpu.to_device(keys)
            res = sorter.sort(d)
            torch.cuda.synchronize()
            assert res.global_offset == 0 and res.keys.numel() == n
            assert np.array_equal(gpu.to_host(res.keys), np.sort(keys)), n
            assert sorter.check_fault() == 0
    finally:
        sorter.close()


# ----------------------------------------------------------------------------- other key types and orders
@pytest.mark.parametrize("r", [8, 4])
@pytest.mark.parametrize("descending", [False, True])
@pytest.mark.parametrize("key_type", ["uint32", "int32", "float32"])
def test_typed_keys(gpu, oracle_mod, key_type, descending, r):
    """int32 / float32 / descending orders, fused into the first pass's load and the last pass's store:
    against numpy's sort of the typed view; float32 includes +-0, +-inf, denormals (no NaN: numpy and IEEE
    total order place them differently)."""
    import torch

    for n, seed in ((0, 1), (1, 1), (4097, 2), ((1 << 20) + 13, 3), ((1 << 23) + 5, 4)):
        raw = oracle_mod.mt19937_keys(n, seed)
        if key_type == "float32":
            f = raw.view(np.float32).copy()
            f[np.isnan(f)] = np.float32(1.5)
            if n > 16:
                f[:8] = np.array([0.0, -0.0, np.inf, -np.inf, 1e-45, -1e-45, 3.4e38, -3.4e38], dtype=np.float32)
            host = f
            t = torch.from_numpy(host.copy()).cuda()
        elif key_type == "int32":
            host = raw.view(np.int32)
            t = torch.from_numpy(host.copy()).cuda()
        else:
            host = raw
            t = gpu.to_device(raw)
        gpu.GPUSortTyped(t, key_type, descending, r=r, check_fault=True)
        got = t.cpu().numpy() if key_type != "uint32" else gpu.to_host(t)
        expect = np.sort(host)
        if descending:
            expect = expect[::-1]
        if key_type == "float32":       # bit-exact, not just ==: -0 must come before +0 ascending
            expect_bits = np.sort(oracle_float_key(host))
            if descending:
                expect_bits = expect_bits[::-1]
            assert np.array_equal(oracle_float_key(got), expect_bits), (key_type, descending, n)
        assert np.array_equal(got, expect), (key_type, descending, n)


def oracle_float_key(f):
    """IEEE total-order key of float32 values as uint32 (numpy restatement of the transform under test)."""
    u = np.asarray(f, dtype=np.float32).view(np.uint32)
    neg = (u >> np.uint32(31)).astype(bool)
    return np.where(neg, ~u, u | np.uint32(0x80000000)).astype(np.uint32)


@pytest.mark.parametrize("key_type,descending", [("int32", True), ("float32", False), ("uint32", True)])
def test_typed_pairs_are_stable(gpu, oracle_mod, key_type, descending):
    import torch

    n = (1 << 20) + 3
    raw = (oracle_mod.mt19937_keys(n, 9) >> np.uint32(22)).astype(np.uint32)     # 1024 distinct values: many ties
    if key_type == "float32":
        host = (raw.astype(np.float32) - np.float32(500.0))
        t = torch.from_numpy(host.copy()).cuda()
    elif key_type == "int32":
        host = raw.astype(np.int32) - np.int32(500)
        t = torch.from_numpy(host.copy()).cuda()
    else:
        host = raw
        t = gpu.to_device(raw)
    vals = torch.arange(n, dtype=torch.int32, device="cuda")
    gpu.GPUSortTyped(t, key_type, descending, d_vals=vals, check_fault=True)
    order = np.argsort(-host.astype(np.float64) if descending else host, kind="stable")
    assert np.array_equal(vals.cpu().numpy(), order.astype(np.int32))
    got = t.cpu().numpy() if key_type != "uint32" else gpu.to_host(t)
    assert np.array_equal(got, host[order])


# ----------------------------------------------------------------------------- 64-bit keys and payloads
def _i64(a):
    import torch

    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.uint64).view(np.int64)).cuda()


def _u64(t):
    return t.cpu().numpy().view(np.uint64)


@pytest.mark.parametrize("r", [8, 4])
def test_u64_golden(gpu, golden, r):
    """uint64 keys, and records with 64-bit payloads, against the committed std::sort / std::stable_sort vectors: equal
    high words, exact duplicates, all-ones keys (the tail padding's value in BOTH words)."""
    d = _i64(golden["u64_keys"])
    gpu.GPUSortWide(d, r=r, check_fault=True)
    assert np.array_equal(_u64(d), golden["u64_sorted"])
    dk, dv = _i64(golden["u64_keys"]), _i64(golden["u64_vals"])
    gpu.GPUSortWide(dk, dv, r=r, check_fault=True)
    assert np.array_equal(_u64(dk), golden["u64_records_keys"]) and np.array_equal(_u64(dv), golden["u64_records_vals"])
    k32, v64 = gpu.to_device(golden["u32v64_keys"]), _i64(golden["u64_vals"])
    gpu.GPUSortWide(k32, v64, r=r, check_fault=True)
    assert np.array_equal(gpu.to_host(k32), golden["u32v64_sorted_keys"]) and np.array_equal(_u64(v64), golden["u32v64_sorted_vals"])


@pytest.mark.parametrize("n", [0, 1, 2, 4097, (1 << 20) + 5, (1 << 22) + 12345])
def test_u64_keys_vs_oracle(gpu, oracle_mod, n):
    rng = np.random.default_rng(n + 3)
    shapes = {
        "uniform": rng.integers(0, 1 << 64, size=n, dtype=np.uint64),
        "low_word_only": rng.integers(0, 1 << 32, size=n, dtype=np.uint64),
        "high_word_only": rng.integers(0, 1 << 32, size=n, dtype=np.uint64) << np.uint64(32),
        "all_ones": np.full(n, 0xFFFFFFFFFFFFFFFF, dtype=np.uint64),
        "few_values": rng.integers(0, 5, size=n, dtype=np.uint64) * np.uint64(0x0101010101010101),
    }
    for name, keys in shapes.items():
        d = _i64(keys)
        gpu.GPUSortWide(d, check_fault=True)
        assert np.array_equal(_u64(d), oracle_mod.std_sort_u64(keys)), (name, n)


@pytest.mark.parametrize("kb,vb", [(64, 64), (64, 32), (32, 64)])
def test_wide_records_are_stable(gpu, oracle_mod, kb, vb):
    """Records = key + payload: equal keys keep their input order (payload = input index), ragged size."""
    import torch

    n = (1 << 21) + 777
    rng = np.random.default_rng(kb + vb)
    if kb == 64:
        keys = (rng.integers(0, 50, size=n, dtype=np.uint64) << np.uint64(32)) | rng.integers(0, 40, size=n, dtype=np.uint64)
        dk = _i64(keys)
    else:
        keys = rng.integers(0, 1000, size=n, dtype=np.uint64)
        dk = gpu.to_device(keys.astype(np.uint32))
    vals = np.arange(n, dtype=np.uint64)
    dv = _i64(vals) if vb == 64 else gpu.to_device(vals.astype(np.uint32))
    gpu.GPUSortWide(dk, dv, check_fault=True)
    ek, ev = oracle_mod.std_stable_sort_records(keys, vals)
    got_k = _u64(dk) if kb == 64 else gpu.to_host(dk).astype(np.uint64)
    got_v = _u64(dv) if vb == 64 else gpu.to_host(dv).astype(np.uint64)
    assert np.array_equal(got_k, ek) and np.array_equal(got_v, ev)
    with pytest.raises(ValueError):
        gpu.GPUSortWide(gpu.to_device(np.zeros(4, dtype=np.uint32)))              # 32/0 is the ordinary entry
    assert gpu.lib().lsdsort_wide_workspace_bytes(100, 8, 32, 32) == 0
    assert gpu.lib().lsdsort_u64_device(dk.data_ptr(), None, 0, 10, 8, None) == -4   # LSDSORT_ERR_WORKSPACE


def test_typed_sort_limits(gpu):
    import torch
    from lsdradixsort_amd import errors

    t = torch.zeros(100, dtype=torch.int32, device="cuda")
    ws = gpu.alloc_workspace(100, 8)
    assert gpu.lib().lsdsort_keys_device(t.data_ptr(), None, ws.data_ptr(), ws.numel(), 100, 8, 7, 0, None) == errors.LSDSORT_ERR_INVALID_ARG
    assert gpu.lib().lsdsort_keys_device(t.data_ptr(), None, ws.data_ptr(), ws.numel(), 100, 2, 1, 0, None) == errors.LSDSORT_ERR_UNSUPPORTED


# ----------------------------------------------------------------------------- full size, by property
def _as_u64(t):
    import torch

    return t.to(torch.int64) & 0xFFFFFFFF


@pytest.mark.parametrize("r", [8, 4])
def test_full_size_properties(gpu, oracle_mod, r):
    """BASELINE configs[1]/[2]: 2^28 keys.  Too big for the CPU oracle in test time, so:
    sortedness, permutation (all digit histograms unchanged + checksums), idempotence, and a
    test-only cross-check against torch.sort (rocPRIM) of the same array."""
    import torch

    n = 1 << 28
    gen = torch.Generator(device="cuda")
    gen.manual_seed(1234 + r)
    d = torch.randint(-(1 << 31), (1 << 31) - 1, (n,), dtype=torch.int32, device="cuda", generator=gen)
    hist_before = gpu.DigitHistograms(d, 8).clone()
    sum_before = int(_as_u64(d).sum().item())
    xor_before = int(torch.bitwise_xor(d[: n // 2], d[n // 2:]).sum().item())
    expect = torch.sort(_as_u64(d)).values
    gpu.GPULSDRadixSort(d, r, check_fault=True)
    u = _as_u64(d)
    assert bool((u[1:] >= u[:-1]).all()), "not sorted"
    assert torch.equal(gpu.DigitHistograms(d, 8), hist_before), "not a permutation of the input"
    assert int(u.sum().item()) == sum_before
    assert torch.equal(u, expect), "differs from torch.sort of the same keys"
    del expect, u
    again = d.clone()
    gpu.GPULSDRadixSort(again, r, check_fault=True)
    assert torch.equal(again, d), "sorting the sorted array changed it"
    _ = xor_before


def test_maximum_size(gpu):
    """The largest n the library accepts (LSDSORT_MAX_KEYS = 2^30 - 1: 30 value bits per status word),
    4 GiB of keys (the size of BASELINE configs[3] on ONE GPU): sorted, a permutation (digit
    histograms and sum unchanged), equal to torch.sort; one key more is refused, not mis-sorted."""
    import torch
    from lsdradixsort_amd import errors

    n = (1 << 30) - 1
    gen = torch.Generator(device="cuda")
    gen.manual_seed(99)
    d = torch.randint(-(1 << 31), (1 << 31) - 1, (n,), dtype=torch.int32, device="cuda", generator=gen)
    hist_before = gpu.DigitHistograms(d, 8).clone()
    sum_before = int(_as_u64(d).sum().item())
    ws = gpu.alloc_workspace(n, 8)
    gpu.GPULSDRadixSort(d, 8, workspace=ws, check_fault=True)
    assert torch.equal(gpu.DigitHistograms(d, 8), hist_before), "not a permutation of the input"
    step = 1 << 28                                   # compare in slices: int64 views of 2^30 keys are 8 GiB each
    for lo in range(0, n, step):
        u = _as_u64(d[lo:min(n, lo + step + 1)])
        assert bool((u[1:] >= u[:-1]).all()), "not sorted"
        del u
    assert sum(int(_as_u64(d[lo:lo + step]).sum().item()) for lo in range(0, n, step)) == sum_before
    assert gpu.lib().lsdsort_workspace_bytes(n + 1, 8, 0) == 0
    st = gpu.lib().lsdsort_u32_device(d.data_ptr(), ws.data_ptr(), ws.numel(), n + 1, 8, None)   # refused before any access
    assert st == errors.LSDSORT_ERR_TOO_LARGE


def test_full_size_pairs_properties(gpu):
    """BASELINE configs[4]: 2^27 (key, payload) pairs = 1 GiB.  Payload = original index, so
    stability is checkable: among equal keys payloads ascend, and keys[payload] == sorted keys."""
    import torch

    n = 1 << 27
    gen = torch.Generator(device="cuda")
    gen.manual_seed(99)
    keys = torch.randint(0, 1 << 20, (n,), dtype=torch.int32, device="cuda", generator=gen) * 2048 + 5   # ~128 copies of each key
    orig = keys.clone()
    vals = torch.arange(n, dtype=torch.int32, device="cuda")
    gpu.GPULSDRadixSort(keys, 8, d_vals=vals, check_fault=True)
    k, v = _as_u64(keys), vals.to(torch.int64)
    assert bool((k[1:] >= k[:-1]).all())
    assert torch.equal(orig[v], keys), "payload does not point at its key"
    same = k[1:] == k[:-1]
    assert bool((v[1:][same] > v[:-1][same]).all()), "equal keys out of input order: not stable"
    assert int(v.sum().item()) == n * (n - 1) // 2


@pytest.mark.parametrize("pairs", [False, True])
def test_small_sorts_are_one_launch_and_equal_the_chained_form(gpu, oracle_mod, pairs):
    """Up to 16384 items (lsdsort_set_small_sort, on by default): one workgroup, four 8-bit digit passes in LDS.  Every size around
    the workgroup's rows and the capacity, keys with dead digits, duplicates and the padding value; against std::sort /
    std::stable_sort (the oracle) and against the chained form's output; 16385 items take the chained form again."""
    import torch

    rng = np.random.default_rng(77)
    for n in (1, 2, 63, 64, 65, 511, 512, 513, 1000, 4096, 8191, 12345, 16383, 16384, 16385):
        for kind in ("uniform", "few_values", "all_ones", "low_byte_only"):
            keys = oracle_mod.mt19937_keys(n, n + len(kind))
            if kind == "few_values":
                keys = (keys % np.uint32(5)) * np.uint32(0x01010101)
            elif kind == "all_ones":
                keys = np.full(n, 0xFFFFFFFF, dtype=np.uint32)
            elif kind == "low_byte_only":
                keys = keys & np.uint32(0xFF)
            vals = rng.permutation(n).astype(np.uint32) if pairs else None
            got = {}
            for on in (True, False):
                gpu.set_small_sort(on)
                try:
                    got[on] = _sort_dev(gpu, keys, 8, 0, vals)
                finally:
                    gpu.set_small_sort(True)
            if pairs:
                ek, ev = oracle_mod.std_stable_sort_pairs(keys, vals)
                for on in (True, False):
                    assert np.array_equal(got[on][0], ek) and np.array_equal(got[on][1], ev), (n, kind, on)
            else:
                expect = oracle_mod.std_sort(keys)
                for on in (True, False):
                    assert np.array_equal(got[on], expect), (n, kind, on)
    # a captured small sort replays like any other
    n = 10000
    static_k = gpu.to_device(oracle_mod.mt19937_keys(n, 5))
    ws = gpu.alloc_workspace(n, 8)
    gpu.GPULSDRadixSort(static_k, 8, workspace=ws)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        gpu.GPULSDRadixSort(static_k, 8, workspace=ws)
    keys = oracle_mod.mt19937_keys(n, 6)
    static_k.copy_(gpu.to_device(keys))
    graph.replay()
    torch.cuda.synchronize()
    assert np.array_equal(gpu.to_host(static_k), np.sort(keys))
    assert gpu.lib().lsdsort_check_device(ws.data_ptr(), None) == 0
