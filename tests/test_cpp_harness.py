"""The C++ face (include/lsdsort.hpp) through a harness shaped like the reference's
TestGPULSDRadixSort (LSDRadixSort.cu:912-1030).  Compiles everywhere; runs on the GPU box."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "test_lsd_sort.cpp")


def _build(tmp_path):
    exe = str(tmp_path / "test_lsd_sort")
    libdir = os.path.join(ROOT, "lsdradixsort_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-I", os.path.join(ROOT, "include"), SRC, "-o", exe,
                           "-L", libdir, "-l:liblsdsort.so", f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def _build_sharded(tmp_path):
    exe = str(tmp_path / "test_sharded")
    libdir = os.path.join(ROOT, "lsdradixsort_amd")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-std=c++17", "-O2", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "test_sharded.cpp"), "-o", exe, "-L", libdir, "-l:liblsdsort.so",
                           f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def test_sharded_cpp_test_compiles_and_links(tmp_path):
    assert os.path.exists(_build_sharded(tmp_path))


@pytest.mark.gpu
def test_sharded_cpp_world_of_one(tmp_path, gpu):
    """The C++ multi-GPU entry (lsdsort_comm_*, lsdsort_sharded_u32_device) with a communicator of one rank."""
    out = subprocess.run([_build_sharded(tmp_path)], capture_output=True, text=True, timeout=300,
                         env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert out.returncode == 0, out.stderr[-3000:]
    assert "sharded cpp test ok" in out.stdout


def test_cpp_harness_compiles_and_links(tmp_path):
    assert os.path.exists(_build(tmp_path))


@pytest.mark.gpu
def test_cpp_harness_runs(tmp_path, gpu):
    out = subprocess.run([_build(tmp_path)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    assert "cpp harness ok" in out.stdout
