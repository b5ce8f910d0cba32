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
