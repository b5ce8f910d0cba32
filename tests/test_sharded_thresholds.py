"""CPU suite: the splitter rule of the C++ multi-GPU step (lsdsort_sharded_thresholds, sharded.hip) without a GPU.

Every rank samples its shard at a regular stride, the samples are gathered, and each rank derives ITS thresholds from
the sorted (key, source rank) sample.  Simulated here for worlds of 1, 2, 4 and 8 in numpy: the sample is taken exactly
as the device kernel takes it (keys[i * n / m]), the thresholds come from the library, buckets are formed with
bucket(key) = number of thresholds <= key, "exchanged" in source-rank order and sorted per rank.  Checked: the
concatenation of the ranks' slices is the sorted union (any distribution: the rule only has to be a consistent cut),
thresholds ascend, and the slices are balanced where fixed MSB buckets are not -- including runs of one value longer
than a bucket (cut between source ranks) and keys equal to 0xFFFFFFFF (a threshold of 2^32).
"""
import numpy as np
import pytest

S = 512   # LSDSORT_SPLITTER_SAMPLES


def _gather(shards):
    g = np.zeros((len(shards), 1 + S), dtype=np.uint32)
    for r, keys in enumerate(shards):
        n = keys.size
        m = min(S, n)
        g[r, 0] = m
        if m:
            idx = (np.arange(m, dtype=np.uint64) * np.uint64(n)) // np.uint64(m)     # the device kernel's stride
            g[r, 1:1 + m] = keys[idx.astype(np.int64)]
    return g


def _sharded_sort(shards):
    from lsdradixsort_amd import sharded_thresholds

    world = len(shards)
    g = _gather(shards)
    received = [[] for _ in range(world)]
    for r, keys in enumerate(shards):
        th = sharded_thresholds(g, world, S, r)
        assert len(th) == world - 1 and all(0 <= t <= 1 << 32 for t in th)
        assert all(a <= b for a, b in zip(th, th[1:]))                               # ascending: buckets are ranges
        bucket = np.zeros(keys.size, dtype=np.int64)
        for t in th:
            bucket += keys.astype(np.uint64) >= np.uint64(t) if t < (1 << 32) else 0
        for b in range(world):
            received[b].append(keys[bucket == b])                                    # stable, source-rank order
    return [np.sort(np.concatenate(parts)) if parts else np.zeros(0, np.uint32) for parts in received]


def _shards(kind, world, n, rng):
    if kind == "uniform":
        return [rng.integers(0, 1 << 32, size=n, dtype=np.uint64).astype(np.uint32) for _ in range(world)]
    if kind == "small":            # top bits clear: MSB buckets send everything to rank 0
        return [rng.integers(0, 1 << 20, size=n, dtype=np.uint64).astype(np.uint32) for _ in range(world)]
    if kind == "constant":
        return [np.full(n, 0xDEADBEEF, dtype=np.uint32) for _ in range(world)]
    if kind == "all_ones":
        return [np.full(n, 0xFFFFFFFF, dtype=np.uint32) for _ in range(world)]
    if kind == "zeros_and_ones":
        return [np.where(rng.random(n) < 0.5, 0, 0xFFFFFFFF).astype(np.uint32) for _ in range(world)]
    if kind == "few":              # seven distinct values, one of them 70 % of the keys
        vals = np.array([3, 1 << 8, 1 << 16, 0x7FFFFFFF, 0x80000000, 0xFFFFFFFE, 0xFFFFFFFF], dtype=np.uint32)
        p = np.array([0.7, 0.05, 0.05, 0.05, 0.05, 0.05, 0.05])
        return [vals[rng.choice(7, size=n, p=p)] for _ in range(world)]
    if kind == "sorted_shards":    # rank r holds the r-th slice of an already sorted array
        allk = np.sort(rng.integers(0, 1 << 32, size=n * world, dtype=np.uint64).astype(np.uint32))
        return [allk[r * n:(r + 1) * n] for r in range(world)]
    if kind == "ragged":           # unequal shard sizes, one of them empty, one shorter than the sample
        sizes = [0, 7, n, 3 * n, S - 1, n // 2, 1, 2 * n][:world]
        return [rng.integers(0, 1 << 24, size=s, dtype=np.uint64).astype(np.uint32) for s in sizes]
    raise AssertionError(kind)


@pytest.mark.parametrize("world", [1, 2, 4, 8])
@pytest.mark.parametrize("kind", ["uniform", "small", "constant", "all_ones", "zeros_and_ones", "few", "sorted_shards", "ragged"])
def test_rule_cuts_consistently_and_balances(world, kind):
    rng = np.random.default_rng(world * 131 + len(kind))
    n = 20000
    shards = _shards(kind, world, n, rng)
    slices = _sharded_sort(shards)
    union = np.sort(np.concatenate(shards))
    assert np.array_equal(np.concatenate(slices), union)                             # rank order IS the global order
    total = union.size
    if total and kind != "ragged":
        # no rank receives more than its share plus one shard (a (value, rank) class cannot be cut) plus sampling noise
        worst = max(s.size for s in slices)
        assert worst <= total / world + n * (0.0 if kind in ("uniform", "small", "sorted_shards") else 1.0) + 0.15 * total / world + 64, \
            (kind, world, [s.size for s in slices])


def test_threshold_arithmetic_by_hand():
    from lsdradixsort_amd import errors, lib, sharded_thresholds
    import ctypes

    # two ranks, every sample equal to 5: the sorted sample is (5,0) x S, (5,1) x S, cut at (5,1):
    # rank 0's fives go below it (threshold 6), rank 1's go above (threshold 5)
    g = np.zeros((2, 1 + S), dtype=np.uint32)
    g[:, 0] = S
    g[:, 1:] = 5
    assert sharded_thresholds(g, 2, S, 0) == [6]
    assert sharded_thresholds(g, 2, S, 1) == [5]
    # the same with 0xFFFFFFFF: rank 0 keeps everything (2^32: above every key)
    g[:, 1:] = 0xFFFFFFFF
    assert sharded_thresholds(g, 2, S, 0) == [1 << 32]
    assert sharded_thresholds(g, 2, S, 1) == [0xFFFFFFFF]
    # no samples anywhere: nothing to cut
    g[:, 0] = 0
    assert sharded_thresholds(g, 2, S, 0) == [1 << 32]
    # bad arguments
    L = lib()
    out = (ctypes.c_uint64 * 8)()
    gp = g.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32))
    assert L.lsdsort_sharded_thresholds(gp, 3, S, 0, out) == errors.LSDSORT_ERR_INVALID_ARG
    assert L.lsdsort_sharded_thresholds(gp, 2, S, 2, out) == errors.LSDSORT_ERR_INVALID_ARG
    assert L.lsdsort_sharded_thresholds(None, 2, S, 0, out) == errors.LSDSORT_ERR_INVALID_ARG
    g[0, 0] = S + 1                                                                   # a count the row cannot hold
    assert L.lsdsort_sharded_thresholds(gp, 2, S, 0, out) == errors.LSDSORT_ERR_INVALID_ARG
