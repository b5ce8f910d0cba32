"""bench.py's input generator must emit the portable stream BASELINE.md names: raw
std::mt19937(seed) outputs (not the reference's implementation-defined RNG, Utils.h:24-33)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_bench_generator_is_std_mt19937(oracle_mod, golden):
    import bench

    for seed in (0, 1, 7, 5489):
        assert np.array_equal(bench.mt19937_keys(100000, seed), oracle_mod.mt19937_keys(100000, seed)), seed
    seed, index, value = (int(x) for x in golden["mt19937_kat"])
    assert int(bench.mt19937_keys(index + 1, seed)[index]) == value
    assert np.array_equal(bench.mt19937_keys(16384, 0), golden["in__uniform_16384_seed0"])
