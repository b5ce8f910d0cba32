"""Host-side mirror of the reference's sort path over the C-ABI (include/lsdsort.h).

Names follow the reference's own (``LSDRadixSort/LSDRadixSort.cu``): ``GPULSDRadixSort``
(.cu:839), ``BuildHistograms`` (.cu:660), ``LSDRadixSortKernel`` -> ``rank_scatter`` (.cu:795).
PyTorch is plumbing only: device memory (``torch.int32`` tensors holding the uint32 bit
patterns) and the current HIP stream.  All computation happens in ``liblsdsort.so``; nothing
here falls back to PyTorch or the CPU.
"""
from __future__ import annotations

import ctypes

import numpy as np

from . import errors
from ._lib import LsdsortTiming, lib
from .errors import LSDSORT_ALGO_ONESWEEP, LSDSORT_ALGO_STAGED, check

__all__ = [
    "sort", "sort_pairs", "to_device", "to_host", "workspace_bytes", "GPULSDRadixSort",
    "GPULSDRadixSortTimed", "BuildHistograms", "BuildOffsets", "RankScatter", "DigitHistograms",
    "MSBPartition", "SplitterPartition", "GPUSortTyped", "GPUSortWide", "GPUSortMulti", "set_hybrid", "tile_keys", "set_tile_config", "set_rank_method", "rank_method", "set_xcd_chunk", "LSDSORT_ALGO_ONESWEEP", "LSDSORT_ALGO_STAGED",
]


def _torch():
    import torch

    return torch


# ------------------------------------------------------------------------------ host-pointer entries
def _host_u32(a: np.ndarray, name: str) -> np.ndarray:
    if not isinstance(a, np.ndarray) or a.dtype != np.uint32 or a.ndim != 1 or not a.flags.c_contiguous:
        raise TypeError(f"{name} must be a contiguous 1-D numpy.uint32 array")
    if not a.flags.writeable:
        raise TypeError(f"{name} must be writeable (sorted in place)")
    return a


def sort(keys: np.ndarray, radix_bits: int = 8) -> np.ndarray:
    """``sort(uint32_t* keys, size_t n)``: host array, in place, ascending (``lsdsort_u32_ex``)."""
    _host_u32(keys, "keys")
    check(lib().lsdsort_u32_ex(keys.ctypes.data, keys.size, radix_bits, 1), "lsdsort_u32_ex")
    return keys


def sort_pairs(keys: np.ndarray, vals: np.ndarray) -> tuple[np.ndarray, np.ndarray]:
    """Key/value sort, stable by key, host arrays in place (``lsdsort_pairs_u32``)."""
    _host_u32(keys, "keys")
    _host_u32(vals, "vals")
    if keys.size != vals.size:
        raise ValueError("keys and vals differ in length")
    check(lib().lsdsort_pairs_u32(keys.ctypes.data, vals.ctypes.data, keys.size), "lsdsort_pairs_u32")
    return keys, vals


# ------------------------------------------------------------------------------ device plumbing
def to_device(a: np.ndarray, device: str = "cuda"):
    """uint32 host array -> int32 device tensor with the same bits."""
    torch = _torch()
    a = np.ascontiguousarray(a, dtype=np.uint32)
    return torch.from_numpy(a.view(np.int32)).to(device)


def to_host(t) -> np.ndarray:
    """int32 device tensor -> uint32 host array with the same bits."""
    return t.detach().cpu().numpy().view(np.uint32)


def _dev_i32(t, name: str):
    torch = _torch()
    if not isinstance(t, torch.Tensor) or t.dtype != torch.int32 or not t.is_cuda or not t.is_contiguous():
        raise TypeError(f"{name} must be a contiguous torch.int32 CUDA tensor (uint32 bit patterns)")
    return t


def _stream(stream=None) -> int:
    torch = _torch()
    s = stream if stream is not None else torch.cuda.current_stream()
    return int(s.cuda_stream)


def workspace_bytes(n: int, radix_bits: int = 8, pairs: bool = False, algorithm: int = LSDSORT_ALGO_ONESWEEP) -> int:
    return int(lib().lsdsort_workspace_bytes_ex(n, radix_bits, int(pairs), algorithm))


def alloc_workspace(n: int, radix_bits: int = 8, pairs: bool = False, algorithm: int = LSDSORT_ALGO_ONESWEEP,
                    device: str = "cuda", stream=None):
    """Workspace tensor for a sort of up to ``n`` keys.  ``stream``: the stream the sort will run on when that is not
    torch's current stream -- the block is then allocated under it, so the caching allocator orders its reuse
    after the sort's kernels instead of after whatever the current stream is doing."""
    torch = _torch()
    nbytes = workspace_bytes(n, radix_bits, pairs, algorithm)
    if nbytes == 0 and n > 0:
        raise errors.LsdsortError(errors.LSDSORT_ERR_INVALID_ARG, "lsdsort_workspace_bytes_ex", "bad (n, radix_bits)")
    # torch's caching allocator returns 512-byte aligned blocks; the ABI needs 256.
    if stream is not None:
        with torch.cuda.stream(stream):
            return torch.empty(max(nbytes, 256), dtype=torch.uint8, device=device)
    return torch.empty(max(nbytes, 256), dtype=torch.uint8, device=device)


def tile_keys(radix_bits: int) -> int:
    return int(lib().lsdsort_tile_keys(radix_bits))


def set_tile_config(radix_bits: int, config_id: int) -> None:
    check(lib().lsdsort_set_tile_config(radix_bits, config_id), "lsdsort_set_tile_config")


def set_xcd_chunk(chunk: int) -> None:
    """Consecutive tiles kept on one XCD by the rank-and-scatter kernel (0 = off; speed only)."""
    check(lib().lsdsort_set_xcd_chunk(chunk), "lsdsort_set_xcd_chunk")


def set_hybrid(on: bool) -> None:
    """The hybrid form of sorts of 2^25 .. 9.6e8 items with 8- or 4-bit digits (global passes on bits 16-31 + an LDS-resident local
    stage, decided on the device; ``lsdsort_set_hybrid``).  Default on; off = every digit through global memory, always."""
    check(lib().lsdsort_set_hybrid(1 if on else 0), "lsdsort_set_hybrid")


def set_small_sort(on: bool) -> None:
    """Sorts of up to 16384 items in one launch (``lsdsort_set_small_sort``).  Default on; off = the chained form at every size."""
    check(lib().lsdsort_set_small_sort(1 if on else 0), "lsdsort_set_small_sort")


def workspace_form(workspace, stream=None) -> int:
    """1 if the last sort queued in ``workspace`` ran the hybrid form, 0 if the ordinary passes (``lsdsort_workspace_form``)."""
    import ctypes

    out = ctypes.c_int(0)
    check(lib().lsdsort_workspace_form(workspace.data_ptr(), _stream(stream), ctypes.byref(out)), "lsdsort_workspace_form")
    return out.value


def set_pass_skipping(on: bool) -> None:
    """Skip passes whose digit is the same for every key (decided on the device from the digit counts; default on)."""
    check(lib().lsdsort_set_pass_skipping(1 if on else 0), "lsdsort_set_pass_skipping")


def set_rank_method(method: int) -> None:
    """-1 auto, 0 peer-mask forms only, 2 returning-LDS-add wherever the device probe passed."""
    check(lib().lsdsort_set_rank_method(method), "lsdsort_set_rank_method")


def rank_method(radix_bits: int) -> int:
    m = lib().lsdsort_rank_method(radix_bits)
    if m < 0:
        check(m, "lsdsort_rank_method")
    return int(m)


# ------------------------------------------------------------------------------ device-resident sort
def GPULSDRadixSort(d_keys, r: int = 8, d_vals=None, algorithm: int = LSDSORT_ALGO_ONESWEEP, workspace=None,
                    stream=None, check_fault: bool = False):
    """Device-resident sort in place: the reference's ``GPULSDRadixSort(a, b, h, ...)`` (.cu:839).

    ``d_keys`` (and ``d_vals``) are int32 CUDA tensors of uint32 bit patterns; the result lands
    in ``d_keys`` like the reference's ``a``.  Stream-ordered, no synchronisation unless
    ``check_fault`` (then the workspace fault word is read back).
    """
    _dev_i32(d_keys, "d_keys")
    n = d_keys.numel()
    pairs = d_vals is not None
    if pairs:
        _dev_i32(d_vals, "d_vals")
        if d_vals.numel() != n:
            raise ValueError("keys and vals differ in length")
    if workspace is None:
        # the call returns while the passes are still running: a temporary workspace must belong to THEIR stream
        workspace = alloc_workspace(n, r, pairs, algorithm, d_keys.device, stream=stream)
    st = lib().lsdsort_u32_device_ex(d_keys.data_ptr(), d_vals.data_ptr() if pairs else None, workspace.data_ptr(),
                                     workspace.numel(), n, r, algorithm, _stream(stream))
    check(st, "lsdsort_u32_device_ex")
    if check_fault and n:
        check(lib().lsdsort_check_device(workspace.data_ptr(), _stream(stream)), "lsdsort_check_device")
    return d_keys if not pairs else (d_keys, d_vals)


def GPUSortMulti(d_keys, payloads, r: int = 8, workspace=None, stream=None, check_fault: bool = False):
    """Keys with one to three 32-bit payload arrays (``lsdsort_multi_u32_device``): every array in ``payloads`` (int32 CUDA
    tensors as long as ``d_keys``) is permuted exactly like the keys, stable by key.  In place."""
    import ctypes

    _dev_i32(d_keys, "d_keys")
    n = d_keys.numel()
    payloads = list(payloads)
    if not 1 <= len(payloads) <= 3:
        raise ValueError("one to three payload arrays")
    for i, v in enumerate(payloads):
        _dev_i32(v, f"payloads[{i}]")
        if v.numel() != n:
            raise ValueError("keys and payloads differ in length")
    if workspace is None:
        need = int(lib().lsdsort_workspace_bytes(n, r, len(payloads)))
        torch = _torch()
        if stream is not None:
            with torch.cuda.stream(stream):
                workspace = torch.empty(max(need, 256), dtype=torch.uint8, device=d_keys.device)
        else:
            workspace = torch.empty(max(need, 256), dtype=torch.uint8, device=d_keys.device)
    ptrs = (ctypes.c_void_p * len(payloads))(*[v.data_ptr() for v in payloads])
    check(lib().lsdsort_multi_u32_device(d_keys.data_ptr(), ptrs, len(payloads), workspace.data_ptr(), workspace.numel(), n, r,
                                         _stream(stream)), "lsdsort_multi_u32_device")
    if check_fault and n:
        check(lib().lsdsort_check_device(workspace.data_ptr(), _stream(stream)), "lsdsort_check_device")
    return d_keys, payloads


_KEY_TYPES = {"uint32": 0, "int32": 1, "float32": 2}


def GPUSortTyped(d_keys, key_type: str = "int32", descending: bool = False, d_vals=None, r: int = 8, workspace=None,
                 stream=None, check_fault: bool = False):
    """Device-resident sort of 32-bit keys of another type or order (``lsdsort_keys_device``): ``key_type`` in
    "uint32" / "int32" / "float32" says how the 32 bits of each element of ``d_keys`` (an int32 or float32 CUDA
    tensor) compare; float32 uses IEEE total order.  Stable with ``d_vals`` (int32 payloads).  In place."""
    torch = _torch()
    if not (d_keys.is_cuda and d_keys.is_contiguous() and d_keys.dtype in (torch.int32, torch.float32)):
        raise TypeError("d_keys: a contiguous int32 or float32 CUDA tensor")
    n = d_keys.numel()
    pairs = d_vals is not None
    if pairs:
        _dev_i32(d_vals, "d_vals")
        if d_vals.numel() != n:
            raise ValueError("keys and vals differ in length")
    if workspace is None:
        workspace = alloc_workspace(n, r, pairs, LSDSORT_ALGO_ONESWEEP, d_keys.device, stream=stream)
    check(lib().lsdsort_keys_device(d_keys.data_ptr(), d_vals.data_ptr() if pairs else None, workspace.data_ptr(),
                                    workspace.numel(), n, r, _KEY_TYPES[key_type], int(bool(descending)), _stream(stream)),
          "lsdsort_keys_device")
    if check_fault and n:
        check(lib().lsdsort_check_device(workspace.data_ptr(), _stream(stream)), "lsdsort_check_device")
    return d_keys if not pairs else (d_keys, d_vals)


def GPUSortWide(d_keys, d_vals=None, r: int = 8, workspace=None, stream=None, check_fault: bool = False):
    """64-bit keys and / or 64-bit payloads, in place (``lsdsort_u64_device`` / ``lsdsort_records_device``).
    ``d_keys``: int64 CUDA tensor (uint64 bit patterns) or int32 (uint32 bit patterns); ``d_vals``: None (64-bit keys
    only), int32 or int64.  The 32/32 combination is ``GPULSDRadixSort``.  Stable by key."""
    torch = _torch()
    bits = {torch.int32: 32, torch.int64: 64}
    if not (isinstance(d_keys, torch.Tensor) and d_keys.is_cuda and d_keys.is_contiguous() and d_keys.dtype in bits):
        raise TypeError("d_keys: a contiguous int32 or int64 CUDA tensor")
    kb = bits[d_keys.dtype]
    vb = 0
    if d_vals is not None:
        if not (isinstance(d_vals, torch.Tensor) and d_vals.is_cuda and d_vals.is_contiguous() and d_vals.dtype in bits):
            raise TypeError("d_vals: a contiguous int32 or int64 CUDA tensor")
        if d_vals.numel() != d_keys.numel():
            raise ValueError("keys and vals differ in length")
        vb = bits[d_vals.dtype]
    if (kb, vb) in ((32, 0), (32, 32)):
        raise ValueError("32-bit keys with no or 32-bit payloads: use GPULSDRadixSort")
    n = d_keys.numel()
    need = int(lib().lsdsort_wide_workspace_bytes(n, r, kb, vb))
    if need == 0:
        raise errors.LsdsortError(errors.LSDSORT_ERR_INVALID_ARG, "lsdsort_wide_workspace_bytes", "bad (n, radix_bits)")
    if workspace is None:
        if stream is not None:
            with torch.cuda.stream(stream):
                workspace = torch.empty(need, dtype=torch.uint8, device=d_keys.device)
        else:
            workspace = torch.empty(need, dtype=torch.uint8, device=d_keys.device)
    if vb == 0:
        st = lib().lsdsort_u64_device(d_keys.data_ptr(), workspace.data_ptr(), workspace.numel(), n, r, _stream(stream))
        check(st, "lsdsort_u64_device")
    else:
        st = lib().lsdsort_records_device(d_keys.data_ptr(), d_vals.data_ptr(), kb, vb, workspace.data_ptr(), workspace.numel(), n, r,
                                          _stream(stream))
        check(st, "lsdsort_records_device")
    if check_fault and n:
        check(lib().lsdsort_wide_check_device(workspace.data_ptr(), n, r, kb, vb, _stream(stream)), "lsdsort_wide_check_device")
    return d_keys if d_vals is None else (d_keys, d_vals)


def GPULSDRadixSortTimed(d_keys, r: int = 8, d_vals=None, algorithm: int = LSDSORT_ALGO_ONESWEEP, workspace=None,
                         stream=None) -> dict:
    """Same sort with per-kernel hipEvent times (``lsdsort_u32_device_timed``).  Blocking."""
    _dev_i32(d_keys, "d_keys")
    n = d_keys.numel()
    pairs = d_vals is not None
    if workspace is None:
        workspace = alloc_workspace(n, r, pairs, algorithm, d_keys.device, stream=stream)
    t = LsdsortTiming()
    st = lib().lsdsort_u32_device_timed(d_keys.data_ptr(), d_vals.data_ptr() if pairs else None, workspace.data_ptr(),
                                        workspace.numel(), n, r, algorithm, _stream(stream), ctypes.byref(t))
    check(st, "lsdsort_u32_device_timed")
    return {
        "total_ms": t.total_ms, "clear_ms": t.clear_ms, "histogram_ms": t.histogram_ms, "scan_ms": t.scan_ms,
        "scatter_ms": [t.scatter_ms[i] for i in range(t.passes)], "passes": t.passes, "tile_keys": t.tile_keys,
        "tiles": t.tiles, "hybrid": t.hybrid, "local_ms": t.local_ms,
    }


# ------------------------------------------------------------------------------ stage entries
def BuildHistograms(d_keys, r: int, bit_group: int, stream=None):
    """h[tile][digit] for one digit: ``BuildHistogramsKernel`` (.cu:660-702)."""
    torch = _torch()
    _dev_i32(d_keys, "d_keys")
    n = d_keys.numel()
    tk = tile_keys(r)
    tiles = (n + tk - 1) // tk
    h = torch.empty((tiles, 1 << r), dtype=torch.int32, device=d_keys.device)
    check(lib().lsdsort_tile_histograms_u32_device(d_keys.data_ptr(), n, r, bit_group, h.data_ptr(), _stream(stream)),
          "lsdsort_tile_histograms_u32_device")
    return h


def BuildOffsets(d_hist, r: int, stream=None):
    """(local, global) offset tables from h[tile][digit]: the reference's .cu:862-895."""
    torch = _torch()
    _dev_i32(d_hist, "d_hist")
    tiles = d_hist.shape[0]
    local = torch.empty_like(d_hist)
    glob = torch.empty_like(d_hist)
    scratch = torch.empty(max(int(lib().lsdsort_tile_offsets_scratch_bytes(tiles, r)), 256), dtype=torch.uint8,
                          device=d_hist.device)
    check(lib().lsdsort_tile_offsets_u32_device(d_hist.data_ptr(), local.data_ptr(), glob.data_ptr(), tiles, r,
                                                scratch.data_ptr(), _stream(stream)),
          "lsdsort_tile_offsets_u32_device")
    return local, glob


def RankScatter(d_in, d_global, r: int, bit_group: int, d_vals=None, stream=None):
    """One rank-and-scatter pass from a global offset table: ``LSDRadixSortKernel`` (.cu:795-837)."""
    torch = _torch()
    _dev_i32(d_in, "d_in")
    _dev_i32(d_global, "d_global")
    out = torch.empty_like(d_in)
    vout = torch.empty_like(d_vals) if d_vals is not None else None
    check(lib().lsdsort_rank_scatter_u32_device(d_in.data_ptr(), out.data_ptr(),
                                                d_vals.data_ptr() if d_vals is not None else None,
                                                vout.data_ptr() if vout is not None else None, d_global.data_ptr(),
                                                d_in.numel(), r, bit_group, _stream(stream)),
          "lsdsort_rank_scatter_u32_device")
    return out if d_vals is None else (out, vout)


def DigitHistograms(d_keys, r: int, stream=None):
    """All 32/r digit histograms in one read (stage 1 of the default pass structure)."""
    torch = _torch()
    _dev_i32(d_keys, "d_keys")
    h = torch.empty((32 // r, 1 << r), dtype=torch.int32, device=d_keys.device)
    check(lib().lsdsort_digit_histograms_u32_device(d_keys.data_ptr(), d_keys.numel(), r, h.data_ptr(),
                                                    _stream(stream)), "lsdsort_digit_histograms_u32_device")
    return h


def MSBPartition(d_keys, msb_bits: int, stream=None):
    """Stable partition by the top ``msb_bits`` bits -> (partitioned keys, int64 bucket counts)."""
    torch = _torch()
    _dev_i32(d_keys, "d_keys")
    n = d_keys.numel()
    out = torch.empty_like(d_keys)
    counts = torch.zeros(1 << msb_bits, dtype=torch.int64, device=d_keys.device)
    ws = torch.empty(max(int(lib().lsdsort_msb_partition_workspace_bytes(n, msb_bits)), 256), dtype=torch.uint8,
                     device=d_keys.device)
    check(lib().lsdsort_msb_partition_u32_device(d_keys.data_ptr(), out.data_ptr(), n, msb_bits, counts.data_ptr(),
                                                 ws.data_ptr(), ws.numel(), _stream(stream)),
          "lsdsort_msb_partition_u32_device")
    if n:
        check(lib().lsdsort_check_device(ws.data_ptr(), _stream(stream)), "lsdsort_check_device")
    return out, counts


def ThresholdPartition(d_keys, thresholds, stream=None):
    """Stable partition by 64-bit thresholds (1, 3 or 7 ascending values in [0, 2^32]; 2^32 = above every key):
    bucket(key) = number of thresholds <= key -> (partitioned keys, int64 bucket counts).  The form the sharded step's
    splitter rule uses (``lsdsort_threshold_partition_u32_device``)."""
    torch = _torch()
    _dev_i32(d_keys, "d_keys")
    th = [int(x) for x in thresholds]
    if len(th) not in (0, 1, 3, 7):
        raise ValueError("threshold count must be 0, 1, 3 or 7 (2, 4 or 8 buckets)")
    bits = (len(th) + 1).bit_length() - 1
    n = d_keys.numel()
    out = torch.empty_like(d_keys)
    counts = torch.zeros(1 << bits, dtype=torch.int64, device=d_keys.device)
    ws = torch.empty(max(int(lib().lsdsort_msb_partition_workspace_bytes(n, bits)), 256), dtype=torch.uint8,
                     device=d_keys.device)
    arr = (ctypes.c_uint64 * max(len(th), 1))(*th)
    check(lib().lsdsort_threshold_partition_u32_device(d_keys.data_ptr(), out.data_ptr(), n, bits, arr, counts.data_ptr(),
                                                       ws.data_ptr(), ws.numel(), _stream(stream)),
          "lsdsort_threshold_partition_u32_device")
    if n:
        check(lib().lsdsort_check_device(ws.data_ptr(), _stream(stream)), "lsdsort_check_device")
    return out, counts


def sharded_thresholds(gathered, world: int, samples_per_rank: int, rank: int):
    """Host arithmetic of the sharded step's splitter rule (``lsdsort_sharded_thresholds``; no GPU): ``gathered`` is a
    [world][1 + samples_per_rank] uint32 array (valid count, then samples, per source rank) -> world - 1 thresholds
    of rank ``rank`` (Python ints in [0, 2^32])."""
    import numpy as np

    g = np.ascontiguousarray(gathered, dtype=np.uint32).reshape(world, 1 + samples_per_rank)
    out = (ctypes.c_uint64 * max(world - 1, 1))()
    check(lib().lsdsort_sharded_thresholds(g.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)), world, samples_per_rank, rank, out),
          "lsdsort_sharded_thresholds")
    return [int(out[i]) for i in range(world - 1)]


def SplitterPartition(d_keys, splitters, stream=None):
    """Stable partition by value: bucket(key) = number of ``splitters`` (ascending uint32 values, 1, 3 or 7 of
    them) <= key -> (partitioned keys, int64 bucket counts).  No splitters: one bucket."""
    torch = _torch()
    _dev_i32(d_keys, "d_keys")
    sp = [int(x) & 0xFFFFFFFF for x in splitters]
    if len(sp) not in (0, 1, 3, 7):
        raise ValueError("splitter count must be 0, 1, 3 or 7 (2, 4 or 8 buckets)")
    bits = (len(sp) + 1).bit_length() - 1
    n = d_keys.numel()
    out = torch.empty_like(d_keys)
    counts = torch.zeros(1 << bits, dtype=torch.int64, device=d_keys.device)
    ws = torch.empty(max(int(lib().lsdsort_msb_partition_workspace_bytes(n, bits)), 256), dtype=torch.uint8,
                     device=d_keys.device)
    arr = (ctypes.c_uint32 * max(len(sp), 1))(*sp)
    check(lib().lsdsort_splitter_partition_u32_device(d_keys.data_ptr(), out.data_ptr(), n, bits, arr, counts.data_ptr(),
                                                      ws.data_ptr(), ws.numel(), _stream(stream)),
          "lsdsort_splitter_partition_u32_device")
    if n:
        check(lib().lsdsort_check_device(ws.data_ptr(), _stream(stream)), "lsdsort_check_device")
    return out, counts
