"""CPU suite: host-side logic of the C++ multi-GPU step (lsdradixsort_amd/csrc/sharded.hip) without a GPU.

lsdsort_sharded_plan is the arithmetic every rank does on the gathered count matrix before the exchange: where
its sends start in its partitioned shard, where each source's keys land in its output (source-rank order keeps
the exchange stable), how many keys it ends up with and where they sit in the global order.  Checked against
numpy for worlds of 1, 2, 4 and 8, uniform and skewed matrices; argument checks of the communicator entries need
no device either.
"""
import ctypes

import numpy as np
import pytest


def _plan(L, m, world, rank):
    mat = np.ascontiguousarray(m, dtype=np.uint64)
    send = np.zeros(world, dtype=np.uint64)
    recv = np.zeros(world, dtype=np.uint64)
    n_out = ctypes.c_uint64(0)
    off = ctypes.c_uint64(0)
    u64p = ctypes.POINTER(ctypes.c_uint64)
    st = L.lsdsort_sharded_plan(mat.ctypes.data_as(u64p), world, rank, send.ctypes.data_as(u64p), recv.ctypes.data_as(u64p),
                                ctypes.byref(n_out), ctypes.byref(off))
    return st, send, recv, n_out.value, off.value


@pytest.mark.parametrize("world", [1, 2, 4, 8])
@pytest.mark.parametrize("kind", ["uniform", "skew", "sparse"])
def test_plan_matches_numpy(world, kind):
    from lsdradixsort_amd import lib

    L = lib()
    rng = np.random.default_rng(world * 7 + len(kind))
    if kind == "uniform":
        m = rng.integers((1 << 24) - 5000, (1 << 24) + 5000, size=(world, world))
    elif kind == "skew":
        m = np.zeros((world, world), dtype=np.int64)
        m[:, 0] = rng.integers(1, 1 << 27, size=world)               # top bits clear: everything goes to rank 0
    else:
        m = rng.integers(0, 3, size=(world, world)) * rng.integers(0, 1000, size=(world, world))
    m = m.astype(np.uint64)
    total_before = 0
    for rank in range(world):
        st, send, recv, n_out, off = _plan(L, m, world, rank)
        assert st == 0
        assert np.array_equal(send, np.concatenate([[0], np.cumsum(m[rank])[:-1]]).astype(np.uint64))     # buckets of MY shard
        assert np.array_equal(recv, np.concatenate([[0], np.cumsum(m[:, rank])[:-1]]).astype(np.uint64))  # sources, in rank order
        assert n_out == int(m[:, rank].sum())
        assert off == total_before                                                                       # slices tile the global order
        total_before += n_out
    assert total_before == int(m.sum())


def test_plan_and_comm_argument_checks():
    from lsdradixsort_amd import errors, lib

    L = lib()
    m = np.ones((3, 3), dtype=np.uint64)
    assert _plan(L, m, 3, 0)[0] == errors.LSDSORT_ERR_INVALID_ARG          # worlds are 1, 2, 4, 8
    assert _plan(L, np.ones((2, 2), dtype=np.uint64), 2, 2)[0] == errors.LSDSORT_ERR_INVALID_ARG
    assert L.lsdsort_comm_unique_id(None) == errors.LSDSORT_ERR_INVALID_ARG
    handle = ctypes.c_void_p()
    assert L.lsdsort_comm_create(None, 2, 0, ctypes.byref(handle)) == errors.LSDSORT_ERR_INVALID_ARG
    ident = (ctypes.c_ubyte * 128)()
    assert L.lsdsort_comm_create(ident, 3, 0, ctypes.byref(handle)) == errors.LSDSORT_ERR_INVALID_ARG
    assert L.lsdsort_comm_create(ident, 2, 5, ctypes.byref(handle)) == errors.LSDSORT_ERR_INVALID_ARG
    assert L.lsdsort_comm_destroy(None) == errors.LSDSORT_OK
    assert L.lsdsort_sharded_workspace_bytes(1 << 20, 1 << 21, 3, 8) == 0
    assert L.lsdsort_sharded_workspace_bytes(1 << 20, 1 << 21, 8, 7) == 0
    a = L.lsdsort_sharded_workspace_bytes(1 << 20, 1 << 21, 8, 8)
    b = L.lsdsort_sharded_workspace_bytes(1 << 21, 1 << 22, 8, 8)
    assert 0 < a < b and a % 256 == 0
    # send buffer + local sort workspace (its ping-pong buffer) dominate: about 4 B/key each
    assert a >= 4 * (1 << 20) + 4 * (1 << 21)
    assert L.lsdsort_strerror(errors.LSDSORT_ERR_COMM)


@pytest.mark.parametrize("world,sub", [(1, 2), (2, 2), (2, 4), (4, 2), (4, 4), (8, 2), (8, 1)])
def test_sub_bucket_plan_simulated_exchange(world, sub):
    """lsdsort_sharded_plan_sub (sub-bucket pipelining, round 3): every rank's offsets from the [src][world * sub] bucket
    counts.  The whole exchange is then replayed in numpy -- each rank's shard partitioned into its buckets, every (source,
    bucket) piece copied to the receiver's offset -- and the concatenation of the ranks' outputs, each sub-bucket sorted on
    its own, must be the sorted union: the property the pipelined step relies on (a rank's sub-buckets are consecutive key
    ranges)."""
    from lsdradixsort_amd import lib

    L = lib()
    u64p = ctypes.POINTER(ctypes.c_uint64)
    rng = np.random.default_rng(100 * world + sub)
    B = world * sub
    bits = B.bit_length() - 1
    shards = [rng.integers(0, 1 << 32, size=int(rng.integers(0, 5000)), dtype=np.uint64).astype(np.uint32) for _ in range(world)]
    bucket_of = (lambda k: (k >> np.uint32(32 - bits)).astype(np.int64)) if bits else (lambda k: np.zeros(k.size, dtype=np.int64))
    m = np.array([np.bincount(bucket_of(s), minlength=B) for s in shards], dtype=np.uint64)
    parted = [s[np.argsort(bucket_of(s), kind="stable")] for s in shards]          # what the partition pass leaves
    plans = []
    for rank in range(world):
        send = np.zeros(B, dtype=np.uint64)
        recv = np.zeros(sub * world, dtype=np.uint64)
        sizes = np.zeros(sub, dtype=np.uint64)
        n_out, off = ctypes.c_uint64(0), ctypes.c_uint64(0)
        assert L.lsdsort_sharded_plan_sub(m.ctypes.data_as(u64p), world, sub, rank, send.ctypes.data_as(u64p), recv.ctypes.data_as(u64p),
                                          sizes.ctypes.data_as(u64p), ctypes.byref(n_out), ctypes.byref(off)) == 0
        assert np.array_equal(send, np.concatenate([[0], np.cumsum(m[rank])[:-1]]).astype(np.uint64))
        assert n_out.value == int(m[:, rank * sub:(rank + 1) * sub].sum()) and int(sizes.sum()) == n_out.value
        assert off.value == int(m[:, :rank * sub].sum())
        plans.append((send, recv, sizes, n_out.value, off.value))
    outs = []
    for rank in range(world):
        send, recv, sizes, n_out, _ = plans[rank]
        out = np.zeros(n_out, dtype=np.uint32)
        for j in range(sub):
            for src in range(world):
                cnt = int(m[src, rank * sub + j])
                s_off = int(plans[src][0][rank * sub + j])
                out[int(recv[j * world + src]): int(recv[j * world + src]) + cnt] = parted[src][s_off: s_off + cnt]
        begin = 0
        for j in range(sub):                                                       # each sub-bucket sorted on its own
            out[begin: begin + int(sizes[j])].sort()
            begin += int(sizes[j])
        outs.append(out)
    assert np.array_equal(np.concatenate(outs), np.sort(np.concatenate(shards)))
    assert L.lsdsort_sharded_plan_sub(m.ctypes.data_as(u64p), world, 3, 0, None, None, None, None, None) == -1


def test_thresholds_cut_into_parts():
    """lsdsort_sharded_thresholds_parts: the splitter rule with world * sub parts (<= 8).  With 4 ranks x 2 sub-buckets the
    seven thresholds ascend and cut a uniform sample into eight near-equal parts; parts = world is the old entry."""
    from lsdradixsort_amd import lib

    L = lib()
    world, samples = 4, 512
    rng = np.random.default_rng(3)
    g = np.zeros((world, 1 + samples), dtype=np.uint32)
    g[:, 0] = samples
    g[:, 1:] = np.sort(rng.integers(0, 1 << 32, size=(world, samples), dtype=np.uint64).astype(np.uint32), axis=1)
    u32p, u64p = ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint64)
    t8 = np.zeros(7, dtype=np.uint64)
    assert L.lsdsort_sharded_thresholds_parts(g.ctypes.data_as(u32p), world, samples, 1, 8, t8.ctypes.data_as(u64p)) == 0
    assert np.all(np.diff(t8.astype(np.int64)) >= 0)
    allk = np.sort(g[:, 1:].ravel())
    parts = np.searchsorted(allk, t8.astype(np.uint64), side="left")
    assert np.all(np.abs(np.diff(np.concatenate([[0], parts, [allk.size]])) - allk.size / 8) <= 2)
    t4a, t4b = np.zeros(3, dtype=np.uint64), np.zeros(3, dtype=np.uint64)
    assert L.lsdsort_sharded_thresholds_parts(g.ctypes.data_as(u32p), world, samples, 2, 4, t4a.ctypes.data_as(u64p)) == 0
    assert L.lsdsort_sharded_thresholds(g.ctypes.data_as(u32p), world, samples, 2, t4b.ctypes.data_as(u64p)) == 0
    assert np.array_equal(t4a, t4b)
    assert L.lsdsort_sharded_thresholds_parts(g.ctypes.data_as(u32p), world, samples, 2, 9, t8.ctypes.data_as(u64p)) == -1
