"""CPU suite: the hybrid form's kernels compile for gfx950 WITHOUT scratch.

A register spill in these kernels does not break a parity test -- it multiplies the kernel's memory traffic (round 3: a loop
around the local stage's body spilled 49 registers and took the stage from 0.55 to 2.07 ms; the parity tests stayed green).
hipcc's own resource remarks are the check: ScratchSize 0 and no VGPR spill for every kernel of local_sort.hip and
hybrid.hip (seconds to compile; the rank-and-scatter translation units take minutes and are read by hand)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "lsdradixsort_amd", "csrc")


@pytest.mark.parametrize("source", ["local_sort.hip", "hybrid.hip"])
def test_no_scratch(source, tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc on this machine")
    p = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "--cuda-device-only",
                        "-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(CSRC, source), "-o", str(tmp_path / "x.o")],
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    names = re.findall(r"Function Name: (\S+)", p.stderr)
    scratch = [int(x) for x in re.findall(r"ScratchSize \[bytes/lane\]: (\d+)", p.stderr)]
    spills = [int(x) for x in re.findall(r"VGPRs Spill: (\d+)", p.stderr)]
    assert names and len(names) == len(scratch) == len(spills)
    for name, sc, sp in zip(names, scratch, spills):
        assert sc == 0 and sp == 0, f"{name}: scratch {sc} B/lane, {sp} VGPRs spilled"
