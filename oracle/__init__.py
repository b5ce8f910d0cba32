"""ctypes face of the CPU oracle.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package, and only as the checker.  The product (``lsdradixsort_amd``) never does.

Two libraries:

* ``liboracle.so``  -- our restatement (``lsd_oracle.c`` + ``std_sort.cpp``); built by
  ``make -C oracle``; travels to the GPU box as a built file.
* ``_ref/libref_lsd.so`` -- the reference's own CPU functions compiled from
  ``/root/reference`` in place (``make -C oracle ref``); optional (``ref_available()``).

All arrays are ``numpy.uint32`` and C-contiguous; functions that sort do so on copies unless
the name says ``inplace``.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")
_REF_PATH = os.path.join(_HERE, "_ref", "libref_lsd.so")

_u32p = ctypes.POINTER(ctypes.c_uint32)
_u64p = ctypes.POINTER(ctypes.c_uint64)


def build(with_ref: bool | None = None) -> None:
    """Compile the oracle (and the reference build when /root/reference exists)."""
    subprocess.check_call(["make", "-s", "-C", _HERE])
    if with_ref is None:
        with_ref = os.path.isdir("/root/reference/LSDRadixSort")
    if with_ref:
        subprocess.check_call(["make", "-s", "-C", _HERE, "ref"])


def _load(path: str) -> ctypes.CDLL:
    if not os.path.exists(path):
        raise FileNotFoundError(f"{path} missing -- run `make -C oracle` (or oracle.build())")
    return ctypes.CDLL(path)


_lib = None
_ref = None


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build(with_ref=False)
        L = _load(_LIB_PATH)
        if not hasattr(L, "oracle_std_sort_u64"):       # a library built before the 64-bit legs were added
            build(with_ref=False)
            L = _load(_LIB_PATH)
        sz = ctypes.c_size_t
        L.oracle_get_r_bits.restype = ctypes.c_uint32
        L.oracle_get_r_bits.argtypes = [ctypes.c_uint32, ctypes.c_int, ctypes.c_int]
        L.oracle_lsd_pass.argtypes = [_u32p, _u32p, sz, _u32p, ctypes.c_int, ctypes.c_int]
        L.oracle_lsd_sort.argtypes = [_u32p, _u32p, sz, _u32p, ctypes.c_int]
        L.oracle_lsd_pass_pairs.argtypes = [_u32p, _u32p, _u32p, _u32p, sz, _u32p, ctypes.c_int, ctypes.c_int]
        L.oracle_lsd_sort_pairs.argtypes = [_u32p, _u32p, _u32p, _u32p, sz, _u32p, ctypes.c_int]
        L.oracle_exclusive_scan.argtypes = [_u32p, sz]
        L.oracle_tile_histograms.argtypes = [_u32p, _u32p, sz, sz, ctypes.c_int, ctypes.c_int]
        L.oracle_local_offsets.argtypes = [_u32p, sz, ctypes.c_int]
        L.oracle_global_offsets.argtypes = [_u32p, _u32p, sz, ctypes.c_int]
        L.oracle_rank_scatter.argtypes = [_u32p, _u32p, _u32p, _u32p, sz, sz, ctypes.c_int, ctypes.c_int]
        L.oracle_staged_sort.restype = ctypes.c_void_p
        L.oracle_staged_sort.argtypes = [_u32p, _u32p, _u32p, sz, sz, ctypes.c_int]
        L.oracle_digit_histograms.argtypes = [_u32p, sz, ctypes.c_int, _u64p]
        L.oracle_msb_partition.argtypes = [_u32p, _u32p, sz, ctypes.c_int, _u64p]
        L.oracle_first_mismatch.restype = sz
        L.oracle_first_mismatch.argtypes = [_u32p, _u32p, sz]
        L.oracle_std_sort.argtypes = [_u32p, sz]
        L.oracle_std_stable_sort_pairs.argtypes = [_u32p, _u32p, sz]
        L.oracle_std_sort_u64.argtypes = [_u64p, sz]
        L.oracle_std_stable_sort_records.argtypes = [_u64p, _u64p, sz]
        L.oracle_fill_mt19937.argtypes = [_u32p, sz, ctypes.c_uint32]
        L.oracle_time_std_sort.restype = ctypes.c_double
        L.oracle_time_std_sort.argtypes = [_u32p, sz]
        L.oracle_time_lsd_sort.restype = ctypes.c_double
        L.oracle_time_lsd_sort.argtypes = [_u32p, _u32p, sz, ctypes.c_int]
        _lib = L
    return _lib


def ref_available() -> bool:
    return os.path.exists(_REF_PATH)


def ref() -> ctypes.CDLL:
    """The reference's own CPU functions (LSDRadixSort.cu:25-69,128-139,643-658)."""
    global _ref
    if _ref is None:
        R = _load(_REF_PATH)
        i = ctypes.c_int
        R.ref_lsd_pass.argtypes = [_u32p, _u32p, i, _u32p, i, i]
        R.ref_lsd_sort.argtypes = [_u32p, _u32p, i, _u32p, i]
        R.ref_prefix_sum.argtypes = [_u32p, i]
        R.ref_build_histograms.argtypes = [_u32p, _u32p, i, i, i, i, i]
        R.ref_rng_fill.argtypes = [_u32p, ctypes.c_size_t, ctypes.c_uint, ctypes.c_uint32, ctypes.c_uint32]
        _ref = R
    return _ref


def _p(a: np.ndarray):
    assert a.dtype == np.uint32 and a.flags.c_contiguous, (a.dtype, a.flags)
    return a.ctypes.data_as(_u32p)


def _p64(a: np.ndarray):
    assert a.dtype == np.uint64 and a.flags.c_contiguous
    return a.ctypes.data_as(_u64p)


def _u32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.uint32)


# ----------------------------------------------------------------------------- inputs
def mt19937_keys(n: int, seed: int = 0) -> np.ndarray:
    """Raw std::mt19937(seed) outputs -- the portable input stream."""
    out = np.empty(n, dtype=np.uint32)
    lib().oracle_fill_mt19937(_p(out), n, seed & 0xFFFFFFFF)
    return out


# ----------------------------------------------------------------------------- restatement
def get_r_bits(key: int, r: int, bit_group: int) -> int:
    return int(lib().oracle_get_r_bits(key & 0xFFFFFFFF, r, bit_group))


def std_sort(keys) -> np.ndarray:
    out = _u32(keys).copy()
    lib().oracle_std_sort(_p(out), out.size)
    return out


def std_stable_sort_pairs(keys, vals):
    k, v = _u32(keys).copy(), _u32(vals).copy()
    lib().oracle_std_stable_sort_pairs(_p(k), _p(v), k.size)
    return k, v


def std_sort_u64(keys) -> np.ndarray:
    """std::sort on uint64 keys (no reference counterpart: the reference is uint32 only, .cu:62)."""
    out = np.ascontiguousarray(keys, dtype=np.uint64).copy()
    lib().oracle_std_sort_u64(_p64(out), out.size)
    return out


def std_stable_sort_records(keys, vals):
    """std::stable_sort by key over (key, payload) records; both widened to uint64."""
    k = np.ascontiguousarray(keys, dtype=np.uint64).copy()
    v = np.ascontiguousarray(vals, dtype=np.uint64).copy()
    lib().oracle_std_stable_sort_records(_p64(k), _p64(v), k.size)
    return k, v


def lsd_pass(keys, r: int, bit_group: int) -> np.ndarray:
    a = _u32(keys).copy()
    b = np.empty_like(a)
    h = np.zeros(1 << r, dtype=np.uint32)
    lib().oracle_lsd_pass(_p(a), _p(b), a.size, _p(h), r, bit_group)
    return b if a.size else a


def lsd_sort(keys, r: int = 8) -> np.ndarray:
    a = _u32(keys).copy()
    b = np.empty_like(a)
    h = np.zeros(1 << r, dtype=np.uint32)
    lib().oracle_lsd_sort(_p(a), _p(b), a.size, _p(h), r)
    return a


def lsd_sort_pairs(keys, vals, r: int = 8):
    k, v = _u32(keys).copy(), _u32(vals).copy()
    k2, v2 = np.empty_like(k), np.empty_like(v)
    h = np.zeros(1 << r, dtype=np.uint32)
    lib().oracle_lsd_sort_pairs(_p(k), _p(v), _p(k2), _p(v2), k.size, _p(h), r)
    return k, v


def exclusive_scan(a) -> np.ndarray:
    out = _u32(a).copy()
    lib().oracle_exclusive_scan(_p(out), out.size)
    return out


def tile_histograms(keys, tile: int, r: int, bit_group: int) -> np.ndarray:
    a = _u32(keys)
    tiles = (a.size + tile - 1) // tile
    h = np.zeros((tiles, 1 << r), dtype=np.uint32)
    lib().oracle_tile_histograms(_p(a), _p(h), a.size, tile, r, bit_group)
    return h


def local_offsets(hist: np.ndarray, r: int) -> np.ndarray:
    out = _u32(hist).copy()
    lib().oracle_local_offsets(_p(out), out.shape[0], r)
    return out


def global_offsets(hist: np.ndarray, r: int) -> np.ndarray:
    h = _u32(hist)
    g = np.empty_like(h)
    lib().oracle_global_offsets(_p(h), _p(g), h.shape[0], r)
    return g


def rank_scatter(keys, local: np.ndarray, glob: np.ndarray, tile: int, r: int, bit_group: int) -> np.ndarray:
    a = _u32(keys)
    b = np.empty_like(a)
    lib().oracle_rank_scatter(_p(a), _p(b), _p(_u32(local)), _p(_u32(glob)), a.size, tile, r, bit_group)
    return b


def staged_sort(keys, tile: int, r: int) -> np.ndarray:
    a = _u32(keys).copy()
    b = np.empty_like(a)
    tiles = max(1, (a.size + tile - 1) // tile)
    h = np.zeros(2 * tiles * (1 << r), dtype=np.uint32)
    res = lib().oracle_staged_sort(_p(a), _p(b), _p(h), a.size, tile, r)
    return a if (res is None or res == a.ctypes.data or a.size == 0) else b


def digit_histograms(keys, r: int) -> np.ndarray:
    a = _u32(keys)
    out = np.zeros((32 // r, 1 << r), dtype=np.uint64)
    lib().oracle_digit_histograms(_p(a), a.size, r, _p64(out))
    return out


def msb_partition(keys, msb_bits: int):
    a = _u32(keys)
    out = np.empty_like(a)
    counts = np.zeros(1 << msb_bits, dtype=np.uint64)
    lib().oracle_msb_partition(_p(a), _p(out), a.size, msb_bits, _p64(counts))
    return out, counts


def first_mismatch(a, b) -> int:
    a, b = _u32(a), _u32(b)
    assert a.size == b.size
    return int(lib().oracle_first_mismatch(_p(a), _p(b), a.size))


def time_std_sort(keys) -> float:
    """Milliseconds for one-thread std::sort of a copy of ``keys`` (reference .cu:96-99)."""
    a = _u32(keys).copy()
    return float(lib().oracle_time_std_sort(_p(a), a.size))


def time_lsd_sort(keys, r: int = 8) -> float:
    a = _u32(keys).copy()
    s = np.empty_like(a)
    return float(lib().oracle_time_lsd_sort(_p(a), _p(s), a.size, r))


# ----------------------------------------------------------------------------- the real reference
def time_ref_lsd_sort(keys, r: int = 8) -> float:
    """Milliseconds the reference's own compiled LSDRadixSort (.cu:62-69, oracle/_ref) takes on ``keys`` (buffers
    prepared outside the timed call, as for the restatement)."""
    import time

    a = _u32(keys).copy()
    b = np.empty_like(a)
    b.fill(0)                                   # pages touched outside the timed call
    h = np.zeros(1 << r, dtype=np.uint32)
    R = ref()
    t0 = time.perf_counter()
    R.ref_lsd_sort(_p(a), _p(b), a.size, _p(h), r)
    return (time.perf_counter() - t0) * 1e3


def ref_lsd_sort(keys, r: int) -> np.ndarray:
    a = _u32(keys).copy()
    b = np.zeros_like(a)
    h = np.zeros(1 << r, dtype=np.uint32)
    ref().ref_lsd_sort(_p(a), _p(b), a.size, _p(h), r)
    return b if a.size else a


def ref_lsd_pass(keys, r: int, bit_group: int) -> np.ndarray:
    a = _u32(keys).copy()
    b = np.zeros_like(a)
    h = np.zeros(1 << r, dtype=np.uint32)
    ref().ref_lsd_pass(_p(a), _p(b), a.size, _p(h), r, bit_group)
    return b


def ref_prefix_sum(a) -> np.ndarray:
    out = _u32(a).copy()
    ref().ref_prefix_sum(_p(out), out.size)
    return out


def ref_build_histograms(keys, block: int, r: int, bit_group: int) -> np.ndarray:
    a = _u32(keys)
    assert a.size % block == 0, "the reference requires count % block == 0"
    grid = a.size // block
    h = np.zeros((grid, 1 << r), dtype=np.uint32)
    ref().ref_build_histograms(_p(a), _p(h), a.size, r, bit_group, grid, block)
    return h


def ref_rng_keys(n: int, seed: int = 0, lo: int = 0, hi: int = 0xFFFFFFFF) -> np.ndarray:
    out = np.empty(n, dtype=np.uint32)
    ref().ref_rng_fill(_p(out), n, seed, lo, hi)
    return out
