"""The bounded-wait expiry path of the chained scan (rank_scatter.hpp "Bounded wait").

Nothing in normal operation reaches it, so a diagnostic build (liblsdsort_faultinject.so: -DLSD_FAULT_INJECT, built
next to the product by `make faultinject` / __graft_entry__.build()) mutes one status row and shrinks the spin
limit.  Expected: LSDSORT_ERR_DEVICE_FAULT from lsdsort_check_device, a drained grid within seconds, every store
inside the buffers it was given (guard zones around keys, payloads and workspace untouched), and a clean sort
on the same workspace afterwards.  One run each for keys and pairs; the fault is provoked once, never in a loop.
"""
import ctypes
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAULT_LIB = os.path.join(ROOT, "lsdradixsort_amd", "liblsdsort_faultinject.so")


def test_product_library_has_no_fault_injection_hook():
    from lsdradixsort_amd import _lib

    product = ctypes.CDLL(_lib.LIB_PATH)
    assert not hasattr(product, "lsdsort_debug_fault_inject")
    assert not hasattr(product, "lsdsort_debug_corrupt_counts")
    assert "faultinject" not in os.path.basename(_lib.LIB_PATH)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["keys", "pairs"])
def test_spin_expiry_gives_up_cleanly(mode):
    assert os.path.exists(FAULT_LIB), "build the diagnostic library first: make -C lsdradixsort_amd/csrc faultinject"
    env = dict(os.environ, LSDSORT_LIB=FAULT_LIB)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_fault_worker.py"), mode], env=env, capture_output=True,
                       text=True, timeout=240)
    assert p.returncode == 0, p.stderr[-2000:]
    out = json.loads(p.stdout.strip().splitlines()[-1])
    assert out["launch_status"] == 0
    assert out["check_status"] == -7, out                  # LSDSORT_ERR_DEVICE_FAULT
    assert out["drain_seconds"] < 30.0, out                 # everybody behind the first tile to give up drains at once
    assert out["guards_intact_after_fault"], "a tile that gave up still stored something out of bounds"
    assert out["second_status"] == 0 and out["second_check"] == 0, out
    assert out["second_sorted"], "the sort after a faulted one (same workspace) is wrong"
    if mode == "pairs":
        assert out["second_payload_stable"]
    assert out["guards_intact_after_clean_sort"]


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["counts", "counts_pairs", "counts_sum"])
def test_inconsistent_counts_never_store_out_of_bounds(mode):
    """The round-2 memory fault (gpurun_out/dist8.log): an uncommitted stage-1 variant lost counts,
    the pass tables stopped describing the keys and a top digit's run was stored behind the output.  Falsified counts
    (diagnostic build: lsdsort_debug_corrupt_counts) must now end in LSDSORT_ERR_DEVICE_FAULT with the guard zones around
    keys, payloads and workspace untouched -- by the destination guard of the passes ("counts": every sum still n) and by
    stage 2's sum check ("counts_sum").  Provoked once per mode."""
    assert os.path.exists(FAULT_LIB), "build the diagnostic library first: make -C lsdradixsort_amd/csrc faultinject"
    env = dict(os.environ, LSDSORT_LIB=FAULT_LIB)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_fault_worker.py"), mode], env=env, capture_output=True,
                       text=True, timeout=240)
    assert p.returncode == 0, p.stderr[-2000:]
    out = json.loads(p.stdout.strip().splitlines()[-1])
    assert out["launch_status"] == 0
    assert out["check_status"] == -7, out
    assert out["guards_intact_after_fault"], "stores left the buffers on inconsistent counts"
    assert out["second_status"] == 0 and out["second_check"] == 0, out
    assert out["second_sorted"]
    if mode == "counts_pairs":
        assert out["second_payload_stable"]
    assert out["guards_intact_after_clean_sort"]


@pytest.mark.gpu
def test_inconsistent_hybrid_fields_never_store_out_of_bounds():
    """The hybrid form's global passes run from count fields of their own (hybrid.hip).  Falsified behind the planner
    (sum kept, 1000 keys booked on the wrong digit of the second pass), the pass kernel's destination guard must turn them into
    LSDSORT_ERR_DEVICE_FAULT with the guard zones around keys and workspace intact; the next sort in the workspace is clean."""
    assert os.path.exists(FAULT_LIB), "build the diagnostic library first: make -C lsdradixsort_amd/csrc faultinject"
    env = dict(os.environ, LSDSORT_LIB=FAULT_LIB)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_fault_worker.py"), "hybrid_counts"], env=env, capture_output=True,
                       text=True, timeout=240)
    assert p.returncode == 0, p.stderr[-2000:]
    out = json.loads(p.stdout.strip().splitlines()[-1])
    assert out["launch_status"] == 0
    assert out["check_status"] == -7, out
    assert out["guards_intact_after_fault"], "stores left the buffers on inconsistent hybrid fields"
    assert out["second_status"] == 0 and out["second_check"] == 0, out
    assert out["second_sorted"]
    assert out["guards_intact_after_clean_sort"]


@pytest.mark.gpu
def test_wide_sort_keeps_the_first_inner_sorts_fault():
    """lsdsort_u64_device = two key/value sorts sharing one workspace; the second one's opening memset clears the fault word
    (ADVICE r2).  A bounded-wait give-up in the FIRST inner sort only must still come out of lsdsort_wide_check_device
    (sticky word), and the next call on the same workspace must be clean."""
    assert os.path.exists(FAULT_LIB)
    env = dict(os.environ, LSDSORT_LIB=FAULT_LIB)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_fault_worker.py"), "wide"], env=env, capture_output=True,
                       text=True, timeout=240)
    assert p.returncode == 0, p.stderr[-2000:]
    out = json.loads(p.stdout.strip().splitlines()[-1])
    assert out["launch_status"] == 0
    assert out["check_status"] == -7, out
    assert out["drain_seconds"] < 30.0, out
    assert out["second_status"] == 0 and out["second_check"] == 0 and out["second_sorted"], out
