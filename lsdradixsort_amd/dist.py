"""One-process-per-GPU sort across the GPUs of a node (BASELINE.json configs[3]).

New work: the reference is single-GPU.  Keys are range-partitioned by
their most significant bits -- rank b ends up owning every key whose top log2(world) bits
equal b -- with exactly ONE exchange step:

  1. local stable partition by the top bits          lsdsort_msb_partition_u32_device (HIP)
  2. all-gather of the world x world bucket counts   torch.distributed (RCCL over xGMI)
  3. variable all-to-all of the buckets              all_to_all_single (grouped send/recv: every
                                                     rank streams to its 7 peers at once, so all
                                                     xGMI links are busy; no ring)
  4. local LSD sort of what arrived                  lsdsort_u32_device (HIP)

The globally sorted array is the concatenation of the ranks' results in rank order.

Two drivers of the same step live here:

* ``ShardedSorter`` -- the product path: a thin face over the C++ step behind the C-ABI
  (``lsdsort_comm_*`` / ``lsdsort_sharded_u32_device``, lsdradixsort_amd/csrc/sharded.hip), which
  talks to RCCL itself (grouped ncclSend/ncclRecv on the sort's stream, the count exchange on a
  side stream while the partition pass runs).  ``torch.distributed`` only carries the 128-byte
  RCCL id at set-up.  This is what ``bench.py --gpus N`` times.
* ``distributed_sort`` -- the same step written against ``torch.distributed`` collectives, with the
  compute calls behind a small backend object, so that the exchange logic (and the sampled-
  splitter partition for skewed keys) can be exercised on CPU tensors with ``gloo`` in tests
  (tests inject an oracle-backed backend; the package itself has only the HIP backend).
"""
from __future__ import annotations

from dataclasses import dataclass


def _log2_exact(world: int) -> int:
    bits = world.bit_length() - 1
    if world < 1 or (1 << bits) != world or bits > 3:
        raise ValueError(f"world size must be 1, 2, 4 or 8 (MSB buckets), got {world}")
    return bits


class HipBackend:
    """The product's compute backend: liblsdsort.so on the current CUDA/HIP device."""

    def __init__(self, radix_bits: int = 8):
        from . import api

        self._api = api
        self.radix_bits = radix_bits
        self._ws = None

    def msb_partition(self, keys, msb_bits: int):
        return self._api.MSBPartition(keys, msb_bits)

    def splitter_partition(self, keys, splitters):
        return self._api.SplitterPartition(keys, splitters)

    def sort_inplace(self, keys):
        n = keys.numel()
        need = self._api.workspace_bytes(n, self.radix_bits)
        if self._ws is None or self._ws.numel() < need:
            self._ws = self._api.alloc_workspace(n, self.radix_bits, device=keys.device)
        self._api.GPULSDRadixSort(keys, self.radix_bits, workspace=self._ws)
        return keys

    def empty_like(self, ref, n: int):
        import torch

        return torch.empty(n, dtype=ref.dtype, device=ref.device)

    def check_fault(self) -> int:
        """lsdsort_check_device on the workspace this backend's sorts ran in (0 = ok; synchronises the stream)."""
        if self._ws is None:
            return 0
        import torch

        return int(self._api.lib().lsdsort_check_device(self._ws.data_ptr(), torch.cuda.current_stream().cuda_stream))


@dataclass
class ShardResult:
    keys: object            # this rank's slice of the globally sorted array (torch tensor)
    global_offset: int      # index of keys[0] in the global order
    counts: object          # world x world int64 matrix: counts[src][dst] = keys src sent to dst


class ShardedSorter:
    """One rank of the multi-GPU sort through the C++ step (``partition``: "msb" buckets, or "splitters" sampled
    and cut inside the step for keys MSB buckets would not balance).  Collective construction: rank 0 makes
    the RCCL id, ``torch.distributed`` (any backend) broadcasts its 128 bytes, every rank creates its communicator
    on its current CUDA/HIP device.  Without an initialised process group it is a world of one (the RCCL calls
    are still made: a one-GPU box rehearses the whole path)."""

    PARTITIONS = {"msb": 0, "splitters": 1}     # LSDSORT_PARTITION_MSB / LSDSORT_PARTITION_SPLITTERS

    def __init__(self, radix_bits: int = 8, group=None, slack: float = 0.25, partition: str = "msb", sub_buckets: int = 1):
        import ctypes

        import torch
        import torch.distributed as dist

        from . import api
        from .errors import check

        self._api, self._ctypes, self._check = api, ctypes, check
        if partition not in self.PARTITIONS:
            raise ValueError(f"partition must be 'msb' or 'splitters', got {partition!r}")
        self.partition = partition
        self.radix_bits = radix_bits
        self.slack = slack
        self.group = group
        live = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if live else 1
        self.rank = dist.get_rank(group) if live else 0
        _log2_exact(self.world)
        L = api.lib()
        # Rank 0 makes the RCCL id; EVERY rank takes part in the broadcast whatever happened on rank 0: the message carries a
        # status byte in front of the 128 id bytes (zeros on failure), and every rank raises AFTER it if the status is bad.  So
        # all ranks always run the same sequence of collectives (a rank-0 failure used to leave the others alone in the
        # broadcast while rank 0 went on to the caller's next collective: ADVICE r2).
        ident = (ctypes.c_ubyte * 128)()
        id_status = 0
        if self.rank == 0:
            id_status = int(L.lsdsort_comm_unique_id(ident))
            if id_status != 0:
                ident = (ctypes.c_ubyte * 128)()
        if self.world > 1:
            on_gpu = dist.get_backend(group) == "nccl"
            payload = [id_status & 0xFF] + list(ident)
            t = torch.tensor(payload, dtype=torch.uint8, device="cuda" if on_gpu else "cpu")
            dist.broadcast(t, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
            got = [int(x) for x in t.cpu()]
            id_status = got[0] - 256 if got[0] > 127 else got[0]
            ident = (ctypes.c_ubyte * 128)(*got[1:])
        check(id_status, "lsdsort_comm_unique_id (rank 0)")
        handle = ctypes.c_void_p()
        # ncclCommInitRank is itself collective: a rank that cannot even try (library refuses locally) must be known to all
        # before the others enter it -- a second tiny agreement over torch.distributed when there is more than one rank
        local = int(L.lsdsort_prepare_device())
        if self.world > 1:
            t = torch.tensor([local], dtype=torch.int32, device="cuda" if on_gpu else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
            local = int(t.item())
        check(local, "lsdsort_prepare_device (some rank)")
        check(L.lsdsort_comm_create(ident, self.world, self.rank, ctypes.byref(handle)), "lsdsort_comm_create")
        # sub-bucket pipelining (the exchange of sub-bucket j + 1 under the local sort of j): the same value on every rank
        check(L.lsdsort_comm_set_sub_buckets(handle, sub_buckets), "lsdsort_comm_set_sub_buckets")
        self._comm = handle
        self._ws = None
        self._out = None
        self._last = None

    def close(self):
        if getattr(self, "_comm", None):
            self._api.lib().lsdsort_comm_destroy(self._comm)
            self._comm = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _buffers(self, n_local: int, capacity: int, device):
        import torch

        L = self._api.lib()
        need = int(L.lsdsort_sharded_workspace_bytes(n_local, capacity, self.world, self.radix_bits))
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(max(need, 256), dtype=torch.uint8, device=device)
        if self._out is None or self._out.numel() < capacity:
            self._out = torch.empty(max(capacity, 1), dtype=torch.int32, device=device)
        return self._ws, self._out

    def sort(self, local_keys, capacity: int | None = None) -> "ShardResult":
        """Sort the union of every rank's ``local_keys`` (int32 CUDA tensors of uint32 bit patterns, left untouched);
        returns this rank's slice -- a view into a buffer the sorter owns and reuses on the next call.  Collective.
        ``capacity``: keys this rank can receive (default: its own share plus ``slack``); if any rank's is too small
        every rank learns it before the exchange and the step is repeated once with exact sizes."""
        import torch

        ctypes = self._ctypes
        L = self._api.lib()
        n_local = local_keys.numel()
        stream = torch.cuda.current_stream().cuda_stream
        cap = capacity if capacity is not None else int(n_local * (1.0 + self.slack)) + 4096
        for attempt in (0, 1):
            ws, out = self._buffers(n_local, cap, local_keys.device)
            n_out = ctypes.c_size_t(0)
            offset = ctypes.c_uint64(0)
            matrix = (ctypes.c_uint64 * (self.world * self.world))()
            st = L.lsdsort_sharded_u32_device_ex(self._comm, local_keys.data_ptr(), n_local, out.data_ptr(), cap, ctypes.byref(n_out),
                                                 ctypes.byref(offset), matrix, ws.data_ptr(), ws.numel(), self.radix_bits,
                                                 self.PARTITIONS[self.partition], stream)
            if st == -9 and attempt == 0:            # LSDSORT_ERR_CAPACITY: collective, EVERY rank got it (skewed keys): exact sizes this time
                cap = max(int(n_out.value), 1)
                continue
            self._check(st, "lsdsort_sharded_u32_device")
            self._last = (n_local, cap)
            break
        counts = torch.tensor(list(matrix), dtype=torch.int64).view(self.world, self.world)
        return ShardResult(out[: n_out.value], int(offset.value), counts)

    def check_fault(self) -> int:
        """Fault words of the last step's partition pass and local sort (0 = ok; synchronises the stream)."""
        import torch

        if self._ws is None or getattr(self, "_last", None) is None:
            return 0
        n_local, cap = self._last
        return int(self._api.lib().lsdsort_sharded_check_device(self._ws.data_ptr(), n_local, cap, self.world, self.radix_bits,
                                                                torch.cuda.current_stream().cuda_stream))


SAMPLES_PER_RANK = 4096


def choose_splitters(local_keys, world: int, group=None, samples: int = SAMPLES_PER_RANK):
    """world - 1 ascending uint32 splitters, identical on every rank: each rank contributes ``samples``
    keys taken at a regular stride through its shard (deterministic), the gathered sample is sorted and
    cut into ``world`` equal parts.  Collective.  With 4096 samples per rank the buckets of an arbitrary
    key distribution come out within a few per cent of n/world -- except for values that occur more
    often than that themselves: equal keys cannot be told apart, all of them go to one bucket."""
    import torch
    import torch.distributed as dist

    n = local_keys.numel()
    # a rank with fewer keys than samples repeats some; an empty rank contributes the largest key so
    # that it does not pull the quantiles down
    if n:
        idx = (torch.arange(samples, device=local_keys.device, dtype=torch.int64) * n) // samples
        mine = local_keys[idx].to(torch.int64) & 0xFFFFFFFF
    else:
        mine = torch.full((samples,), 0xFFFFFFFF, dtype=torch.int64, device=local_keys.device)
    everyone = torch.empty(world * samples, dtype=torch.int64, device=local_keys.device)
    dist.all_gather_into_tensor(everyone, mine.contiguous(), group=group)
    ordered = torch.sort(everyone).values.cpu()          # world * 4096 values: host-side work, not the hot path
    return [int(ordered[(b * ordered.numel()) // world]) for b in range(1, world)]


def distributed_sort(local_keys, backend=None, group=None, exchange_always: bool = False,
                     partition: str = "msb") -> ShardResult:
    """Sort the union of every rank's ``local_keys``; returns this rank's slice.

    ``local_keys``: int32 tensor of uint32 bit patterns on this rank's device.  Collective:
    every rank of ``group`` must call it.  Result slices concatenate in rank order.
    ``exchange_always`` runs partition, count exchange and all-to-all even for a world of one
    (a one-GPU box can then exercise the RCCL calls; the default skips them there).
    ``partition``: "msb" -- rank b owns the keys whose top log2(world) bits are b (balanced for
    uniform keys, no extra step); "splitters" -- rank b owns [splitter[b-1], splitter[b]) with
    splitters drawn from a gathered sample (``choose_splitters``): balanced for skewed keys too.
    """
    import torch
    import torch.distributed as dist

    if backend is None:
        backend = HipBackend()
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    msb_bits = _log2_exact(world)

    if world == 1 and not (exchange_always and dist.is_initialized()):
        out = local_keys.clone()
        backend.sort_inplace(out)
        n = out.numel()
        return ShardResult(out, 0, torch.tensor([[n]], dtype=torch.int64))

    # 1. bucket b = keys of rank b, contiguous, in input order
    if partition == "splitters" and world > 1:
        parted, counts = backend.splitter_partition(local_keys, choose_splitters(local_keys, world, group))
    elif partition in ("msb", "splitters"):
        parted, counts = backend.msb_partition(local_keys, msb_bits)
    else:
        raise ValueError(f"partition must be 'msb' or 'splitters', got {partition!r}")
    counts = counts.to(torch.int64)

    # 2. everyone learns the full count matrix (world x world words)
    matrix = torch.empty(world * world, dtype=torch.int64, device=counts.device)
    dist.all_gather_into_tensor(matrix, counts.contiguous(), group=group)
    matrix_host = matrix.cpu().view(world, world)          # split sizes are host-side arguments
    send_sizes = [int(x) for x in matrix_host[rank]]
    recv_sizes = [int(x) for x in matrix_host[:, rank]]

    # 3. one variable all-to-all
    received = backend.empty_like(parted, sum(recv_sizes))
    dist.all_to_all_single(received, parted, output_split_sizes=recv_sizes, input_split_sizes=send_sizes, group=group)

    # 4. local LSD sort (the top bits are constant within a rank; the passes still run on all 32)
    backend.sort_inplace(received)
    offset = int(matrix_host[:, :rank].sum())
    return ShardResult(received, offset, matrix_host)


class LoopbackWorld:
    """``world`` VIRTUAL ranks of the C++ sharded step on the current device (``lsdsort_comm_create_loopback``): the same
    step as over RCCL -- partition, count exchange, capacity verdict, grouped exchange, local sort -- with device copies for
    the fabric, so a one-GPU machine runs it with world > 1.  Each rank is driven by its own host thread on its own stream
    (the step is collective).  ``step`` returns, per rank, ``(status, ShardResult | None, n_out)``."""

    PARTITIONS = ShardedSorter.PARTITIONS

    def __init__(self, world: int, radix_bits: int = 8, sub_buckets: int = 1):
        import ctypes

        from . import api
        from .errors import check

        self._api = api
        _log2_exact(world)
        self.world = world
        self.radix_bits = radix_bits
        handles = (ctypes.c_void_p * world)()
        check(api.lib().lsdsort_comm_create_loopback(world, handles), "lsdsort_comm_create_loopback")
        self._comms = [ctypes.c_void_p(h) for h in handles]
        for c in self._comms:      # sub-bucket pipelining: a collective setting, the same on every rank
            check(api.lib().lsdsort_comm_set_sub_buckets(c, sub_buckets), "lsdsort_comm_set_sub_buckets")
        self.last = [None] * world      # (workspace tensor, n_local, capacity) of each rank's last step

    def close(self):
        for c in getattr(self, "_comms", []):
            self._api.lib().lsdsort_comm_destroy(c)
        self._comms = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def step(self, shards, capacities=None, partition: str = "msb", workspace_bytes=None, timeout: float = 120.0):
        """One collective step: ``shards[r]`` is rank r's int32 CUDA tensor (left untouched), ``capacities[r]`` its output
        capacity (default: everything, so that nothing can overflow).  ``workspace_bytes[r]`` overrides rank r's workspace
        size (tests use a too-small one to make a rank fail on its own)."""
        import ctypes
        import threading

        import torch

        L = self._api.lib()
        W = self.world
        total = sum(int(s.numel()) for s in shards)
        caps = list(capacities) if capacities is not None else [total] * W
        results = [None] * W
        device = torch.cuda.current_device()

        def run(r):
            torch.cuda.set_device(device)
            n_local = int(shards[r].numel())
            need = int(L.lsdsort_sharded_workspace_bytes(n_local, caps[r], W, self.radix_bits))
            given = need if workspace_bytes is None or workspace_bytes[r] is None else int(workspace_bytes[r])
            stream = torch.cuda.Stream()
            with torch.cuda.stream(stream):
                ws = torch.empty(max(given, 256), dtype=torch.uint8, device="cuda")
                out = torch.empty(max(caps[r], 1), dtype=torch.int32, device="cuda")
            n_out = ctypes.c_size_t(0)
            offset = ctypes.c_uint64(0)
            matrix = (ctypes.c_uint64 * (W * W))()
            st = L.lsdsort_sharded_u32_device_ex(self._comms[r], shards[r].data_ptr(), n_local, out.data_ptr(), caps[r], ctypes.byref(n_out),
                                                 ctypes.byref(offset), matrix, ws.data_ptr(), given, self.radix_bits,
                                                 self.PARTITIONS[partition], stream.cuda_stream)
            res = None
            if st == 0:
                fault = int(L.lsdsort_sharded_check_device(ws.data_ptr(), n_local, caps[r], W, self.radix_bits, stream.cuda_stream))
                if fault != 0:
                    st = fault
                else:
                    counts = torch.tensor(list(matrix), dtype=torch.int64).view(W, W)
                    res = ShardResult(out[: n_out.value], int(offset.value), counts)
            stream.synchronize()
            results[r] = (int(st), res, int(n_out.value))

        threads = [threading.Thread(target=run, args=(r,), daemon=True) for r in range(W)]
        for t in threads:
            t.start()
        for t in threads:
            t.join(timeout)
            if t.is_alive():
                raise TimeoutError("a virtual rank is still inside the step: a collective hang")
        return results
