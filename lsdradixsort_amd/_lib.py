"""ctypes binding of liblsdsort.so (include/lsdsort.h).

The HIP library is the product; there is no Python or CPU fallback.  If the shared object is
missing or fails to load, importing this module's ``lib()`` raises immediately and loudly.
"""
from __future__ import annotations

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
PRODUCT_LIB = os.path.join(_HERE, "liblsdsort.so")


def _library_path() -> str:
    """The product library, always -- except for the in-tree DIAGNOSTIC and A/B builds of this very source tree
    (``make faultinject | stats | variant`` write ``liblsdsort_<tag>.so`` next to the product): tests/test_fault_path.py and
    tools/ab_bench.sh select one with ``LSDSORT_LIB``.  Anything else in that variable -- another directory, another name --
    is refused loudly: an environment variable must not be able to put a foreign library behind this package."""
    want = os.environ.get("LSDSORT_LIB")
    if not want:
        return PRODUCT_LIB
    real = os.path.realpath(want)
    name = os.path.basename(real)
    if real == os.path.realpath(PRODUCT_LIB):
        return PRODUCT_LIB
    if os.path.dirname(real) != os.path.realpath(_HERE) or not (name.startswith("liblsdsort_") and name.endswith(".so")):
        raise ImportError(f"LSDSORT_LIB={want!r} is not an in-tree diagnostic build ({_HERE}/liblsdsort_<tag>.so); "
                          "unset it to use the product library")
    return real


LIB_PATH = _library_path()

c_u32p = ctypes.c_void_p   # device or host addresses are passed as integers
c_size = ctypes.c_size_t
c_int = ctypes.c_int

LSDSORT_MAX_PASSES = 32


class LsdsortTiming(ctypes.Structure):
    """Mirror of ``lsdsort_timing`` (include/lsdsort.h)."""

    _fields_ = [
        ("total_ms", ctypes.c_float),
        ("clear_ms", ctypes.c_float),
        ("histogram_ms", ctypes.c_float),
        ("scan_ms", ctypes.c_float),
        ("scatter_ms", ctypes.c_float * LSDSORT_MAX_PASSES),
        ("passes", ctypes.c_int),
        ("tile_keys", ctypes.c_int),
        ("tiles", ctypes.c_int),
        ("hybrid", ctypes.c_int),
        ("local_ms", ctypes.c_float),
    ]


# name -> (restype, argtypes); every symbol include/lsdsort.h declares
SIGNATURES = {
    "lsdsort_u32": (c_int, [c_u32p, c_size]),
    "lsdsort_u32_ex": (c_int, [c_u32p, c_size, c_int, c_int]),
    "lsdsort_u32_loopback": (c_int, [c_u32p, c_size, c_int, c_int]),
    "lsdsort_pairs_u32": (c_int, [c_u32p, c_u32p, c_size]),
    "lsdsort_release_host_cache": (c_int, []),
    "lsdsort_workspace_bytes": (c_size, [c_size, c_int, c_int]),
    "lsdsort_workspace_bytes_ex": (c_size, [c_size, c_int, c_int, c_int]),
    "lsdsort_u32_device": (c_int, [c_u32p, ctypes.c_void_p, c_size, c_size, c_int, ctypes.c_void_p]),
    "lsdsort_pairs_u32_device": (c_int, [c_u32p, c_u32p, ctypes.c_void_p, c_size, c_size, c_int, ctypes.c_void_p]),
    "lsdsort_u32_device_ex": (c_int, [c_u32p, c_u32p, ctypes.c_void_p, c_size, c_size, c_int, c_int, ctypes.c_void_p]),
    "lsdsort_multi_u32_device": (c_int, [c_u32p, ctypes.POINTER(ctypes.c_void_p), c_int, ctypes.c_void_p, c_size, c_size, c_int, ctypes.c_void_p]),
    "lsdsort_keys_device": (c_int, [ctypes.c_void_p, c_u32p, ctypes.c_void_p, c_size, c_size, c_int, c_int, c_int, ctypes.c_void_p]),
    "lsdsort_wide_workspace_bytes": (c_size, [c_size, c_int, c_int, c_int]),
    "lsdsort_u64_device": (c_int, [ctypes.c_void_p, ctypes.c_void_p, c_size, c_size, c_int, ctypes.c_void_p]),
    "lsdsort_records_device": (c_int, [ctypes.c_void_p, ctypes.c_void_p, c_int, c_int, ctypes.c_void_p, c_size, c_size, c_int,
                                       ctypes.c_void_p]),
    "lsdsort_wide_check_device": (c_int, [ctypes.c_void_p, c_size, c_int, c_int, c_int, ctypes.c_void_p]),
    "lsdsort_check_device": (c_int, [ctypes.c_void_p, ctypes.c_void_p]),
    "lsdsort_u32_device_timed": (c_int, [c_u32p, c_u32p, ctypes.c_void_p, c_size, c_size, c_int, c_int,
                                         ctypes.c_void_p, ctypes.POINTER(LsdsortTiming)]),
    "lsdsort_tile_keys": (c_size, [c_int]),
    "lsdsort_tile_histograms_u32_device": (c_int, [c_u32p, c_size, c_int, c_int, c_u32p, ctypes.c_void_p]),
    "lsdsort_tile_offsets_scratch_bytes": (c_size, [c_size, c_int]),
    "lsdsort_tile_offsets_u32_device": (c_int, [c_u32p, c_u32p, c_u32p, c_size, c_int, ctypes.c_void_p, ctypes.c_void_p]),
    "lsdsort_rank_scatter_u32_device": (c_int, [c_u32p, c_u32p, c_u32p, c_u32p, c_u32p, c_size, c_int, c_int,
                                                ctypes.c_void_p]),
    "lsdsort_local_sort_u32_device": (c_int, [c_u32p, c_u32p, c_u32p, c_size, c_int, ctypes.c_void_p]),
    "lsdsort_digit_histograms_u32_device": (c_int, [c_u32p, c_size, c_int, c_u32p, ctypes.c_void_p]),
    "lsdsort_msb_partition_workspace_bytes": (c_size, [c_size, c_int]),
    "lsdsort_msb_partition_u32_device": (c_int, [c_u32p, c_u32p, c_size, c_int, ctypes.c_void_p, ctypes.c_void_p,
                                                 c_size, ctypes.c_void_p]),
    "lsdsort_splitter_partition_u32_device": (c_int, [c_u32p, c_u32p, c_size, c_int, ctypes.POINTER(ctypes.c_uint32),
                                                      ctypes.c_void_p, ctypes.c_void_p, c_size, ctypes.c_void_p]),
    "lsdsort_threshold_partition_u32_device": (c_int, [c_u32p, c_u32p, c_size, c_int, ctypes.POINTER(ctypes.c_uint64),
                                                       ctypes.c_void_p, ctypes.c_void_p, c_size, ctypes.c_void_p]),
    "lsdsort_comm_unique_id": (c_int, [ctypes.c_void_p]),
    "lsdsort_comm_create": (c_int, [ctypes.c_void_p, c_int, c_int, ctypes.POINTER(ctypes.c_void_p)]),
    "lsdsort_comm_create_loopback": (c_int, [c_int, ctypes.POINTER(ctypes.c_void_p)]),
    "lsdsort_comm_destroy": (c_int, [ctypes.c_void_p]),
    "lsdsort_comm_set_sub_buckets": (c_int, [ctypes.c_void_p, c_int]),
    "lsdsort_comm_world": (c_int, [ctypes.c_void_p]),
    "lsdsort_comm_rank": (c_int, [ctypes.c_void_p]),
    "lsdsort_sharded_workspace_bytes": (c_size, [c_size, c_size, c_int, c_int]),
    "lsdsort_sharded_u32_device": (c_int, [ctypes.c_void_p, c_u32p, c_size, c_u32p, c_size, ctypes.POINTER(c_size),
                                           ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64), ctypes.c_void_p,
                                           c_size, c_int, ctypes.c_void_p]),
    "lsdsort_sharded_u32_device_ex": (c_int, [ctypes.c_void_p, c_u32p, c_size, c_u32p, c_size, ctypes.POINTER(c_size),
                                              ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64), ctypes.c_void_p,
                                              c_size, c_int, c_int, ctypes.c_void_p]),
    "lsdsort_sharded_thresholds": (c_int, [ctypes.POINTER(ctypes.c_uint32), c_int, c_int, c_int, ctypes.POINTER(ctypes.c_uint64)]),
    "lsdsort_sharded_thresholds_parts": (c_int, [ctypes.POINTER(ctypes.c_uint32), c_int, c_int, c_int, c_int, ctypes.POINTER(ctypes.c_uint64)]),
    "lsdsort_sharded_plan_sub": (c_int, [ctypes.POINTER(ctypes.c_uint64), c_int, c_int, c_int, ctypes.POINTER(ctypes.c_uint64),
                                         ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64),
                                         ctypes.POINTER(ctypes.c_uint64)]),
    "lsdsort_sharded_check_device": (c_int, [ctypes.c_void_p, c_size, c_size, c_int, c_int, ctypes.c_void_p]),
    "lsdsort_sharded_plan": (c_int, [ctypes.POINTER(ctypes.c_uint64), c_int, c_int, ctypes.POINTER(ctypes.c_uint64),
                                     ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64),
                                     ctypes.POINTER(ctypes.c_uint64)]),
    "lsdsort_last_comm_error": (ctypes.c_char_p, []),
    "lsdsort_strerror": (ctypes.c_char_p, [c_int]),
    "lsdsort_last_hip_error": (c_int, []),
    "lsdsort_last_hip_error_string": (ctypes.c_char_p, []),
    "lsdsort_version": (ctypes.c_char_p, []),
    "lsdsort_device_count": (c_int, []),
    "lsdsort_set_tile_config": (c_int, [c_int, c_int]),
    "lsdsort_prepare_device": (c_int, []),
    "lsdsort_set_xcd_chunk": (c_int, [c_int]),
    "lsdsort_set_pass_skipping": (c_int, [c_int]),
    "lsdsort_set_hybrid": (c_int, [c_int]),
    "lsdsort_set_small_sort": (c_int, [c_int]),
    "lsdsort_workspace_form": (c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(c_int)]),
    "lsdsort_u32_device_prefixed": (c_int, [c_u32p, ctypes.c_void_p, c_size, c_size, c_int, c_int, ctypes.c_void_p]),
    "lsdsort_set_rank_method": (c_int, [c_int]),
    "lsdsort_rank_method": (c_int, [c_int]),
}

_lib = None


def lib() -> ctypes.CDLL:
    """Load liblsdsort.so and bind every entry point.  Raises if the HIP build is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build the HIP extension first "
                "(`make -C lsdradixsort_amd/csrc` or `python -c 'import __graft_entry__ as g; g.build()'`). "
                "lsdradixsort_amd has no CPU fallback."
            )
        L = ctypes.CDLL(LIB_PATH)
        for name, (restype, argtypes) in SIGNATURES.items():
            fn = getattr(L, name)   # AttributeError if the export is missing
            fn.restype = restype
            fn.argtypes = argtypes
        _lib = L
    return _lib
