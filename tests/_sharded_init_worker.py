"""Worker for tests/test_dist_gloo.py::test_sharded_sorter_setup_failure_is_seen_by_every_rank: one rank of a gloo group
that constructs a ShardedSorter while rank 0's lsdsort_comm_unique_id fails (rank 0 used to raise in front of
the id broadcast and leave the others alone in it).  Every rank must raise the same error AFTER the broadcast and then still
be in step with the others for the caller's next collective.  Writes "status,sum" to out_<rank>.txt."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world, port, outdir, mode = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist

    import lsdradixsort_amd as lsd
    from lsdradixsort_amd.dist import ShardedSorter

    dist.init_process_group("gloo", rank=rank, world_size=world)
    L = lsd.lib()
    if mode == "id_fails" and rank == 0:
        L.lsdsort_comm_unique_id = lambda buf: -6          # LSDSORT_ERR_UNSUPPORTED, as when librccl cannot be loaded
    status = 0
    try:
        ShardedSorter()
    except lsd.LsdsortError as e:
        status = e.status
    # the caller's next collective (bench.py does an all_reduce of a "bad" flag here): all ranks must still be aligned
    t = torch.tensor([rank + 1], dtype=torch.int64)
    dist.all_reduce(t)
    with open(os.path.join(outdir, f"out_{rank}.txt"), "w") as f:
        f.write(f"{status},{int(t.item())}")
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
