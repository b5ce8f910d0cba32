"""Worker for tests/test_dist_gloo.py: one rank of a gloo process group on CPU tensors.

The exchange logic under test is lsdradixsort_amd.dist.distributed_sort; the per-rank compute
(partition, local sort) is supplied by an oracle-backed backend defined HERE, in tests/ --
the product package has only the HIP backend.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class OracleBackend:
    radix_bits = 8

    def __init__(self):
        import oracle

        self.oracle = oracle

    def msb_partition(self, keys, msb_bits):
        import torch

        out, counts = self.oracle.msb_partition(keys.numpy().view(np.uint32), msb_bits)
        return torch.from_numpy(out.view(np.int32)), torch.from_numpy(counts.astype(np.int64))

    def splitter_partition(self, keys, splitters):
        import torch

        k = keys.numpy().view(np.uint32)
        bucket = np.searchsorted(np.asarray(splitters, dtype=np.uint32), k, side="right")   # splitters <= key
        order = np.argsort(bucket, kind="stable")
        counts = np.bincount(bucket, minlength=len(splitters) + 1).astype(np.int64)
        return torch.from_numpy(k[order].view(np.int32)), torch.from_numpy(counts)

    def sort_inplace(self, keys):
        import torch

        s = self.oracle.lsd_sort(keys.numpy().view(np.uint32), self.radix_bits)
        keys.copy_(torch.from_numpy(s.view(np.int32)))
        return keys

    def empty_like(self, ref, n):
        import torch

        return torch.empty(n, dtype=ref.dtype)


def shard_keys(rank, n, dist_kind):
    import oracle

    keys = oracle.mt19937_keys(n, rank)                     # rank k fills its shard with seed k
    if dist_kind == "skew":
        keys = (keys >> np.uint32(3)).astype(np.uint32)      # top three bits clear: everything lands on rank 0
    elif dist_kind == "dup":
        keys = (keys % 5).astype(np.uint32) * np.uint32(0x33333333)
    return keys


def main():
    rank, world, port, n, dist_kind, outdir = (int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4]),
                                               sys.argv[5], sys.argv[6])
    partition = sys.argv[7] if len(sys.argv) > 7 else "msb"
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist

    from lsdradixsort_amd.dist import distributed_sort

    dist.init_process_group("gloo", rank=rank, world_size=world)
    n_local = n + rank * 7                                    # ragged shards
    keys = shard_keys(rank, n_local, dist_kind)
    res = distributed_sort(torch.from_numpy(keys.view(np.int32)), backend=OracleBackend(), partition=partition)
    np.save(os.path.join(outdir, f"out_{rank}.npy"), res.keys.numpy().view(np.uint32))
    np.save(os.path.join(outdir, f"meta_{rank}.npy"), np.array([res.global_offset, n_local], dtype=np.int64))
    np.save(os.path.join(outdir, f"counts_{rank}.npy"), res.counts.numpy())
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
