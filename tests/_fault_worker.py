"""Worker for tests/test_fault_path.py: runs in its own process with LSDSORT_LIB pointing at the diagnostic
build liblsdsort_faultinject.so (make -C lsdradixsort_amd/csrc faultinject), never the product library.

Modes keys / pairs: one status row is muted (its tile never publishes), the spin limit is small: the tiles behind it
must give up -- fault word raised, grid drained, no prefix published from a partial sum, nothing stored from an unknown
base.  Modes counts / counts_pairs / counts_sum: the digit counts are falsified behind stage 1, so the pass tables no
longer describe the keys: the destination guard (rank_scatter.hpp) and the sum check (scan_regions_kernel) must turn
that into LSDSORT_ERR_DEVICE_FAULT with every store inside the buffers.  In every mode the sort after it, on the same
workspace, must be clean again.  Prints one JSON line.
"""
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch

import lsdradixsort_amd as lsd

assert "faultinject" in lsd.LIB_PATH, lsd.LIB_PATH
L = lsd.lib()
raw = ctypes.CDLL(lsd.LIB_PATH)
raw.lsdsort_debug_fault_inject.argtypes = [ctypes.c_uint, ctypes.c_uint]
raw.lsdsort_debug_fault_inject.restype = ctypes.c_int
raw.lsdsort_debug_corrupt_counts.argtypes = [ctypes.c_uint, ctypes.c_uint, ctypes.c_uint, ctypes.c_uint]
raw.lsdsort_debug_corrupt_counts.restype = ctypes.c_int

mode = sys.argv[1] if len(sys.argv) > 1 else "keys"
pairs = mode in ("pairs", "counts_pairs")

if mode == "wide":
    # lsdsort_u64_device runs TWO key/value sorts in one shared workspace, and the second sort's opening memset
    # clears the fault word the first one raised.  The muted row applies to the FIRST inner sort only
    # (lsdsort_debug_fault_inject_sorts(1)); lsdsort_wide_check_device must still report it (sticky word in the wide layout).
    raw.lsdsort_debug_fault_inject_sorts.argtypes = [ctypes.c_int]
    n64 = (1 << 22) + 77
    rng = np.random.default_rng(9)
    host64 = rng.integers(0, 1 << 63, size=n64, dtype=np.uint64)
    d = torch.from_numpy(host64.view(np.int64)).cuda()
    wb = int(L.lsdsort_wide_workspace_bytes(n64, 8, 64, 0))
    ws64 = torch.empty(wb, dtype=torch.uint8, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    raw.lsdsort_debug_fault_inject(4000, 5 + 1)
    raw.lsdsort_debug_fault_inject_sorts(1)
    out = {"mode": mode}
    t0 = time.time()
    out["launch_status"] = L.lsdsort_u64_device(d.data_ptr(), ws64.data_ptr(), wb, n64, 8, stream)
    out["check_status"] = L.lsdsort_wide_check_device(ws64.data_ptr(), n64, 8, 64, 0, stream)
    out["drain_seconds"] = round(time.time() - t0, 3)
    raw.lsdsort_debug_fault_inject(0, 0)
    raw.lsdsort_debug_fault_inject_sorts(-1)
    d.copy_(torch.from_numpy(host64.view(np.int64)).cuda())
    out["second_status"] = L.lsdsort_u64_device(d.data_ptr(), ws64.data_ptr(), wb, n64, 8, stream)
    out["second_check"] = L.lsdsort_wide_check_device(ws64.data_ptr(), n64, 8, 64, 0, stream)
    out["second_sorted"] = bool(np.array_equal(d.cpu().numpy().view(np.uint64), np.sort(host64)))
    print(json.dumps(out), flush=True)
    sys.exit(0)
r = 8
n = (1 << 23) + 123 if mode != "hybrid_counts" else (1 << 26) + 123     # hybrid_counts: a size where the hybrid form runs
GUARD = 1 << 16                     # int32 words on either side of the keys / bytes on either side of the workspace
SENT = 0x7E7E7E7E
rng = np.random.default_rng(5)
host = rng.integers(0, 1 << 32, size=n, dtype=np.uint64).astype(np.uint32)


def guarded_i32(fill):
    big = torch.full((GUARD + n + GUARD,), SENT, dtype=torch.int32, device="cuda")
    big[GUARD:GUARD + n] = fill
    return big, big[GUARD:GUARD + n]


kbig, keys = guarded_i32(torch.from_numpy(host.view(np.int32)).cuda())
vbig, vals = guarded_i32(torch.arange(n, dtype=torch.int32, device="cuda")) if pairs else (None, None)
wbytes = lsd.workspace_bytes(n, r, pairs)
wbig = torch.full((GUARD + wbytes + GUARD,), 0x7E, dtype=torch.uint8, device="cuda")
ws = wbig[GUARD:GUARD + wbytes]
assert ws.data_ptr() % 256 == 0
stream = torch.cuda.current_stream().cuda_stream


def guards_intact():
    ok = bool((kbig[:GUARD] == SENT).all()) and bool((kbig[GUARD + n:] == SENT).all())
    ok = ok and bool((wbig[:GUARD] == 0x7E).all()) and bool((wbig[GUARD + wbytes:] == 0x7E).all())
    if pairs:
        ok = ok and bool((vbig[:GUARD] == SENT).all()) and bool((vbig[GUARD + n:] == SENT).all())
    return ok


out = {"pairs": pairs, "n": n}
if mode in ("keys", "pairs"):
    # --- 1. a muted tile in the middle of region 0's chain, small spin limit
    MUTED_ROW = 5
    raw.lsdsort_debug_fault_inject(4000, MUTED_ROW + 1)
else:
    # --- 1'. counts that do not describe the keys (what a miscounting stage-1 variant leaves).  The count
    # table is [pass][digit][region] (8-bit digits, 8 regions).  "counts": 1000 keys of the last pass's highest digit are
    # booked on its lowest digit instead -- every sum is still n, but the bases of all higher digits are 1000 too large and the
    # top digit's run would end 1000 keys behind the output: the passes' destination guard must refuse those stores.
    # "counts_sum": five keys too many in pass 0 -- stage 2's sum check must say so.
    word = lambda p, d, x: (p * 256 + d) * 8 + x
    if mode == "counts_sum":
        raw.lsdsort_debug_corrupt_counts(0, word(0, 0, 0), 5, 0)
    elif mode == "hybrid_counts":
        # the same falsification in the hybrid form's second global pass (field B = the second of its [pass][digit][region]
        # fields; keep_sum bit 1 selects them): the planner has said yes from the bucket counts, the pass's tables are wrong,
        # its destination guard must refuse the stores that would land behind the output
        raw.lsdsort_debug_corrupt_counts(word(1, 255, 7), word(1, 0, 0), 1000, 3)
    else:
        raw.lsdsort_debug_corrupt_counts(word(3, 255, 7), word(3, 0, 0), 1000, 1)
torch.cuda.synchronize()
t0 = time.time()
st = L.lsdsort_u32_device_ex(keys.data_ptr(), vals.data_ptr() if pairs else None, ws.data_ptr(), wbytes, n, r, 0, stream)
out["launch_status"] = st
out["check_status"] = L.lsdsort_check_device(ws.data_ptr(), stream)     # synchronises: the grid has drained
out["drain_seconds"] = round(time.time() - t0, 3)
out["guards_intact_after_fault"] = guards_intact()
# --- 2. the same workspace, fault injection off: a clean sort
raw.lsdsort_debug_fault_inject(0, 0)
raw.lsdsort_debug_corrupt_counts(0, 0, 0, 1)
keys.copy_(torch.from_numpy(host.view(np.int32)).cuda())
if pairs:
    vals.copy_(torch.arange(n, dtype=torch.int32, device="cuda"))
st2 = L.lsdsort_u32_device_ex(keys.data_ptr(), vals.data_ptr() if pairs else None, ws.data_ptr(), wbytes, n, r, 0, stream)
out["second_status"] = st2
out["second_check"] = L.lsdsort_check_device(ws.data_ptr(), stream)
got = keys.cpu().numpy().view(np.uint32)
out["second_sorted"] = bool(np.array_equal(got, np.sort(host)))
if pairs:
    order = np.argsort(host, kind="stable").astype(np.uint32)
    out["second_payload_stable"] = bool(np.array_equal(vals.cpu().numpy().view(np.uint32), order))
out["guards_intact_after_clean_sort"] = guards_intact()
print(json.dumps(out), flush=True)
