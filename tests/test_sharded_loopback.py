"""The C++ sharded step (lsdradixsort_amd/csrc/sharded.hip) with MORE THAN ONE RANK, on the one GPU a test box has.

The reference is single-GPU (LSDRadixSort.cu:839-910); BASELINE configs[3] (4 GiB over 8 GPUs) is new work whose step --
partition, count exchange, capacity verdict, grouped exchange with send / receive offsets, local sort -- had only ever run
with W = 1, where the exchange loop is empty.  The loopback transport (lsdsort_comm_create_loopback) runs the
SAME step with W in {2, 4, 8} virtual ranks in this process: only the fabric calls differ (device copies ordered by events
instead of ncclAllGather / ncclSend / ncclRecv).  Checker: numpy's sort of the union (the output of a keys-only sort is
unique: == std::sort, which is what the reference asserts, .cu:120 / .cu:1018).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _shards(kind, W, seed):
    """W host arrays (uint32), ragged, one of them empty when W > 2."""
    rng = np.random.default_rng(seed)
    sizes = [int(rng.integers(150_000, 400_000)) + 17 * r for r in range(W)]
    if W > 2:
        sizes[1] = 0                                   # an empty shard
    sizes[-1] = 8191 * 5 + 3                           # a small ragged one
    out = []
    for r, n in enumerate(sizes):
        base = rng.integers(0, 1 << 32, size=n, dtype=np.uint64).astype(np.uint32)
        if kind == "uniform":
            k = base
        elif kind == "dead_top_bits":                  # every key < 2^20: MSB buckets send everything to rank 0
            k = base & np.uint32(0xFFFFF)
        elif kind == "all_equal":
            k = np.full(n, 0xC0FFEE11, dtype=np.uint32)
        elif kind == "all_ones":                       # the tail padding's own value
            k = np.full(n, 0xFFFFFFFF, dtype=np.uint32)
        elif kind == "two_values":
            k = np.where(base & 1, np.uint32(0x10), np.uint32(0xF0000000)).astype(np.uint32)
        elif kind == "heavy_value":                    # 70 % one value, the rest uniform
            k = np.where(base % 10 < 7, np.uint32(0x80000001), base).astype(np.uint32)
        elif kind == "presorted_shards":
            k = np.sort(base)
        else:
            raise ValueError(kind)
        out.append(k)
    return out


def _run(gpu, world, shards, partition, capacities=None, radix_bits=8, sub_buckets=1):
    from lsdradixsort_amd.dist import LoopbackWorld

    lw = LoopbackWorld(world, radix_bits, sub_buckets)
    try:
        dev = [gpu.to_device(s) for s in shards]
        res = lw.step(dev, capacities=capacities, partition=partition)
        return res, dev
    finally:
        lw.close()


@pytest.mark.parametrize("partition", ["msb", "splitters"])
@pytest.mark.parametrize("world", [2, 4, 8])
def test_step_with_virtual_ranks(gpu, world, partition):
    for kind in ("uniform", "dead_top_bits", "all_equal", "all_ones", "two_values", "heavy_value", "presorted_shards"):
        shards = _shards(kind, world, 1000 * world + len(kind))
        union = np.sort(np.concatenate(shards))
        res, dev = _run(gpu, world, shards, partition)
        pieces, expect_offset = [], 0
        for r, (st, out, n_out) in enumerate(res):
            assert st == 0, (kind, world, partition, r, st)
            got = gpu.to_host(out.keys)
            assert got.size == n_out
            assert out.global_offset == expect_offset, (kind, world, partition, r)
            # the count matrix every rank reports is the same, and its columns are what the ranks received
            assert np.array_equal(out.counts.numpy(), res[0][1].counts.numpy())
            assert int(out.counts[:, r].sum()) == n_out
            assert [int(x) for x in out.counts.sum(dim=1)] == [s.size for s in shards]
            pieces.append(got)
            expect_offset += n_out
        assert np.array_equal(np.concatenate(pieces), union), (kind, world, partition)
        for r, d in enumerate(dev):                       # the input shards are left untouched
            assert np.array_equal(gpu.to_host(d), shards[r]), (kind, r)
        if partition == "msb" and kind in ("uniform", "presorted_shards"):
            bits = world.bit_length() - 1
            for r, p in enumerate(pieces):                # ownership rule: rank b holds the keys whose top bits are b
                assert p.size == 0 or (int(p[0]) >> (32 - bits) == r and int(p[-1]) >> (32 - bits) == r)
        if partition == "splitters" and kind in ("uniform", "dead_top_bits", "presorted_shards"):
            share = union.size / world                    # sampled splitters balance what MSB buckets cannot
            assert max(p.size for p in pieces) < 1.25 * share + 4096, (kind, [p.size for p in pieces])


@pytest.mark.parametrize("world", [2, 8])
def test_capacity_verdict_is_collective(gpu, world):
    """One rank cannot hold what it would receive: EVERY rank returns LSDSORT_ERR_CAPACITY before anything is exchanged,
    with the size it would have needed in n_out; the repeated step with those sizes succeeds."""
    from lsdradixsort_amd import errors

    shards = _shards("dead_top_bits", world, 77)          # everything belongs to rank 0 under MSB buckets
    total = sum(s.size for s in shards)
    caps = [max(s.size, 1) for s in shards]               # rank 0 would need `total`
    res, _ = _run(gpu, world, shards, "msb", capacities=caps)
    assert [st for st, _, _ in res] == [errors.LSDSORT_ERR_CAPACITY] * world
    needed = [n for _, _, n in res]
    assert needed[0] == total and sum(needed) == total
    res, _ = _run(gpu, world, shards, "msb", capacities=[max(n, 1) for n in needed])
    assert [st for st, _, _ in res] == [0] * world
    assert np.array_equal(gpu.to_host(res[0][1].keys), np.sort(np.concatenate(shards)))


def test_a_rank_that_fails_alone_does_not_strand_its_peers(gpu):
    """A rank with a local error (here: a workspace that is too small, refused before any collective) leaves the step;
    its peers, already inside the count exchange, must come back with LSDSORT_ERR_COMM instead of waiting for ever."""
    from lsdradixsort_amd import errors
    from lsdradixsort_amd.dist import LoopbackWorld

    world = 4
    shards = _shards("uniform", world, 5)
    lw = LoopbackWorld(world)
    try:
        dev = [gpu.to_device(s) for s in shards]
        res = lw.step(dev, workspace_bytes=[None, None, 256, None], timeout=60.0)    # raises TimeoutError on a hang
        assert res[2][0] == errors.LSDSORT_ERR_WORKSPACE
        assert all(res[r][0] == errors.LSDSORT_ERR_COMM for r in (0, 1, 3)), [st for st, _, _ in res]
        # the world is aborted for good: a further step fails at once on every rank, it does not hang either
        res = lw.step(dev, timeout=60.0)
        assert all(st == errors.LSDSORT_ERR_COMM for st, _, _ in res)
    finally:
        lw.close()


@pytest.mark.parametrize("virtual_gpus", [2, 4, 8])
def test_host_entry_with_virtual_gpus(gpu, oracle_mod, virtual_gpus):
    """lsdsort_u32_loopback = the code of lsdsort_u32_ex(num_gpus > 1) -- threads, set-up agreement, capacity retry, step,
    copy back -- on one device.  Uniform keys (first attempt fits) and keys with dead top bits (collective capacity verdict,
    agreement, second attempt with exact sizes), against std::sort; radix 8 and 4."""
    L = gpu.lib()
    n = (1 << 21) + 12345
    for r, mask in ((8, 0xFFFFFFFF), (8, 0x000FFFFF), (4, 0xFFFFFFFF), (8, 0)):
        keys = (oracle_mod.mt19937_keys(n, 30 + virtual_gpus) & np.uint32(mask)).astype(np.uint32)
        expect = oracle_mod.std_sort(keys)
        got = keys.copy()
        assert L.lsdsort_u32_loopback(got.ctypes.data, got.size, r, virtual_gpus) == 0, (r, hex(mask))
        assert np.array_equal(got, expect), (r, hex(mask), virtual_gpus)
    few = np.array([5, 3, 9], dtype=np.uint32)             # fewer keys than ranks: empty shards
    assert L.lsdsort_u32_loopback(few.ctypes.data, 3, 8, virtual_gpus) == 0 and list(few) == [3, 5, 9]
    assert L.lsdsort_u32_loopback(few.ctypes.data, 3, 8, 3) == -1    # not a power of two


@pytest.mark.parametrize("world,sub,partition", [(2, 2, "msb"), (2, 4, "msb"), (4, 2, "msb"), (4, 4, "msb"), (8, 2, "msb"),
                                                 (2, 2, "splitters"), (2, 4, "splitters"), (4, 2, "splitters")])
def test_sub_bucket_pipelining(gpu, world, sub, partition):
    """Sub-bucket pipelining (lsdsort_comm_set_sub_buckets): every rank's key range cut into `sub` consecutive sub-ranges,
    exchanged one grouped exchange after the other, each sorted on the communicator's internal stream while the next is
    exchanged; world x sub = 16 takes the 4-bit partition kernels.  Same checks as the plain step: the rank-order
    concatenation is the sorted union, offsets and counts agree, shards untouched, fault words clean."""
    for kind in ("uniform", "dead_top_bits", "all_equal", "all_ones", "heavy_value", "presorted_shards"):
        shards = _shards(kind, world, 31 * world + sub + len(kind))
        union = np.sort(np.concatenate(shards))
        res, dev = _run(gpu, world, shards, partition, sub_buckets=sub)
        pieces, expect_offset = [], 0
        for r, (st, out, n_out) in enumerate(res):
            assert st == 0, (kind, world, sub, partition, r, st)
            assert out.global_offset == expect_offset
            assert int(out.counts[:, r].sum()) == n_out
            pieces.append(gpu.to_host(out.keys))
            expect_offset += n_out
        assert np.array_equal(np.concatenate(pieces), union), (kind, world, sub, partition)
        for r, d in enumerate(dev):
            assert np.array_equal(gpu.to_host(d), shards[r]), (kind, r)


def test_sub_bucket_argument_checks(gpu):
    from lsdradixsort_amd import errors
    from lsdradixsort_amd.dist import LoopbackWorld

    with pytest.raises(Exception):
        LoopbackWorld(8, 8, 4)                      # 8 x 4 = 32 buckets: more than the partition pass takes
    with pytest.raises(Exception):
        LoopbackWorld(2, 8, 3)
    lw = LoopbackWorld(8, 8, 2)                     # 16 buckets: fine by bit field, refused by value (seven splitters at most)
    try:
        shards = _shards("uniform", 8, 9)
        res = lw.step([gpu.to_device(s) for s in shards], partition="splitters", timeout=60.0)
        assert all(st in (errors.LSDSORT_ERR_UNSUPPORTED, errors.LSDSORT_ERR_COMM) for st, _, _ in res), [st for st, _, _ in res]
    finally:
        lw.close()


def test_step_at_hybrid_sizes(gpu):
    """Two virtual ranks of 2^26 + ... keys each under the MSB partition: every rank's received shard (about 2^26 keys that share
    their top bit) is of the size where the local sort tries the hybrid form, planned below the shard's prefix
    (lsdsort_u32_device_prefixed).  The rank-order concatenation is the sorted union."""
    import torch

    world = 2
    gen = torch.Generator(device="cuda")
    gen.manual_seed(2026)
    dev = [torch.randint(-(1 << 31), (1 << 31) - 1, ((1 << 26) + 4099 * (r + 1),), dtype=torch.int32, device="cuda", generator=gen)
           for r in range(world)]
    union = torch.sort(torch.cat([(d.to(torch.int64) & 0xFFFFFFFF) for d in dev])).values
    from lsdradixsort_amd.dist import LoopbackWorld

    lw = LoopbackWorld(world, 8, 1)
    try:
        res = lw.step(dev, partition="msb")
        pieces = []
        for r, (st, out, n_out) in enumerate(res):
            assert st == 0, (r, st)
            assert out.keys.numel() == n_out
            pieces.append(out.keys.to(torch.int64) & 0xFFFFFFFF)
        assert torch.equal(torch.cat(pieces), union)
    finally:
        lw.close()
