"""CPU suite, part 3: the N>1 path (MSB-bucket exchange) with world_size 2 and 4 over gloo.

Exercises lsdradixsort_amd.dist.distributed_sort -- bucket counts all-gather, variable
all-to-all, global offsets -- on CPU tensors.  Per-rank compute comes from an oracle-backed
backend that lives in tests/_dist_worker.py; on the GPU box the same function runs with the
HIP backend over RCCL (bench.py --gpus N).
"""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return str(s.getsockname()[1])


@pytest.mark.parametrize("world,kind", [(2, "uniform"), (4, "uniform"), (2, "skew"), (2, "dup")])
def test_msb_bucket_exchange(tmp_path, oracle_mod, world, kind):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from _dist_worker import shard_keys

    n = 20000
    port = _free_port()
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_dist_worker.py"), str(r), str(world), port,
                               str(n), kind, str(tmp_path)]) for r in range(world)]
    for p in procs:
        assert p.wait(timeout=180) == 0
    everything = np.concatenate([shard_keys(r, n + 7 * r, kind) for r in range(world)])
    expect = oracle_mod.std_sort(everything)
    outs = [np.load(tmp_path / f"out_{r}.npy") for r in range(world)]
    metas = [np.load(tmp_path / f"meta_{r}.npy") for r in range(world)]
    got = np.concatenate(outs)
    assert got.size == expect.size
    assert np.array_equal(got, expect), "rank-order concatenation is not the global sort"
    offset = 0
    msb = world.bit_length() - 1
    for r in range(world):
        assert int(metas[r][0]) == offset
        offset += outs[r].size
        if outs[r].size:
            assert int(outs[r].min() >> np.uint32(32 - msb)) == r and int(outs[r].max() >> np.uint32(32 - msb)) == r
    counts = np.load(tmp_path / "counts_0.npy")
    assert counts.shape == (world, world) and int(counts.sum()) == expect.size
    for r in range(world):
        assert int(counts[r].sum()) == n + 7 * r and int(counts[:, r].sum()) == outs[r].size
        assert np.array_equal(np.load(tmp_path / f"counts_{r}.npy"), counts)
    if kind == "skew":
        assert outs[0].size == expect.size           # fixed MSB buckets do not balance skewed input


@pytest.mark.parametrize("world,kind", [(2, "skew"), (4, "skew"), (4, "uniform"), (2, "dup")])
def test_sampled_splitter_exchange(tmp_path, oracle_mod, world, kind):
    """partition="splitters" (SURVEY section 8f.2): the rank-order concatenation is still the global sort,
    and skewed keys that fixed MSB buckets pile onto rank 0 now spread over the ranks."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from _dist_worker import shard_keys

    n = 20000
    port = _free_port()
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_dist_worker.py"), str(r), str(world), port,
                               str(n), kind, str(tmp_path), "splitters"]) for r in range(world)]
    for p in procs:
        assert p.wait(timeout=180) == 0
    everything = np.concatenate([shard_keys(r, n + 7 * r, kind) for r in range(world)])
    expect = oracle_mod.std_sort(everything)
    outs = [np.load(tmp_path / f"out_{r}.npy") for r in range(world)]
    metas = [np.load(tmp_path / f"meta_{r}.npy") for r in range(world)]
    assert np.array_equal(np.concatenate(outs), expect), "rank-order concatenation is not the global sort"
    offset = 0
    for r in range(world):
        assert int(metas[r][0]) == offset
        offset += outs[r].size
    counts = np.load(tmp_path / "counts_0.npy")
    assert counts.shape == (world, world) and int(counts.sum()) == expect.size
    for r in range(world):
        assert int(counts[:, r].sum()) == outs[r].size
    if kind in ("skew", "uniform"):                      # distinct-ish keys: every rank within 10 % of its share
        share = expect.size / world
        assert all(abs(o.size - share) < 0.10 * share for o in outs), [o.size for o in outs]


def test_world_size_must_be_power_of_two():
    from lsdradixsort_amd.dist import _log2_exact

    assert [_log2_exact(w) for w in (1, 2, 4, 8)] == [0, 1, 2, 3]
    for bad in (0, 3, 6, 16):
        with pytest.raises(ValueError):
            _log2_exact(bad)


@pytest.mark.parametrize("mode", ["id_fails", "no_device"])
def test_sharded_sorter_setup_failure_is_seen_by_every_rank(tmp_path, mode):
    """ShardedSorter's collective set-up when it cannot succeed: rank 0's id call fails (id_fails: -6 on every rank, AFTER
    the broadcast all of them take part in), or no rank has a gfx950 device (no_device: this container; -2 everywhere, agreed
    through an all_reduce in front of the collective ncclCommInitRank).  Either way every rank raises the same status and
    the next collective of the caller still lines up (rank 0 raised before the broadcast and went on alone)."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("needs a machine without a usable GPU (on a GPU box the set-up would go on into ncclCommInitRank)")
    world = 2
    port = _free_port()
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_sharded_init_worker.py"), str(r), str(world), port,
                               str(tmp_path), mode]) for r in range(world)]
    for p in procs:
        assert p.wait(timeout=180) == 0
    outs = [open(tmp_path / f"out_{r}.txt").read().split(",") for r in range(world)]
    expect = "-6" if mode == "id_fails" else "-2"
    assert [o[0] for o in outs] == [expect] * world, outs
    assert [o[1] for o in outs] == ["3"] * world, outs      # 1 + 2: the all_reduce behind the failed set-up matched up
