"""lsdradixsort_amd -- MI355X-native (gfx950) LSD radix sort for uint32 keys and key/value pairs.

The product is ``liblsdsort.so`` (hand-written HIP kernels behind the C-ABI in
``include/lsdsort.h``); this package is its thin Python face for tests, ``bench.py`` and the
one-process-per-GPU driver (``dist.py``).  There is no CPU fallback: without the built
library, or without a gfx950 device, every entry raises.
"""
from .errors import (LSDSORT_ALGO_ONESWEEP, LSDSORT_ALGO_STAGED, LSDSORT_MAX_KEYS, LsdsortError)  # noqa: F401
from ._lib import LIB_PATH, lib  # noqa: F401
from .api import (BuildHistograms, BuildOffsets, DigitHistograms, GPULSDRadixSort, GPULSDRadixSortTimed, GPUSortMulti, GPUSortTyped, GPUSortWide,  # noqa: F401
                  MSBPartition, RankScatter, SplitterPartition, ThresholdPartition, sharded_thresholds, alloc_workspace, rank_method, set_hybrid, set_pass_skipping, set_small_sort, workspace_form, set_rank_method, set_tile_config, set_xcd_chunk, sort,
                  sort_pairs, tile_keys,
                  to_device, to_host, workspace_bytes)

__version__ = "0.1.0"
