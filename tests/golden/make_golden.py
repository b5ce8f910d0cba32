#!/usr/bin/env python3
"""Generate tests/golden/lsd_golden.npz from the REFERENCE's own CPU code.

Run in the build container (needs /root/reference):  python tests/golden/make_golden.py

Expected outputs come from oracle/_ref/libref_lsd.so -- the reference's LSDRadixSortPass /
LSDRadixSort / PrefixSum / BuildHistogramsCPU (LSDRadixSort/LSDRadixSort.cu:25-69, 128-139,
643-658) compiled in place by `make -C oracle ref` -- and from std::sort / std::stable_sort
(the reference's other CPU leg, .cu:97).  At generation time every case asserts
ref LSD (r in 1,2,4,8,16) == std::sort, the same transitive check the reference makes
(.cu:120).  The reference ships no fixture files of its own, so these
vectors are what pins the oracle and the HIP path on machines where /root/reference is absent.

Fixtures are data only: inputs and expected outputs, no source text.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import oracle  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lsd_golden.npz")


def cases():
    c = {}
    u = oracle.mt19937_keys(16384, 0)
    c["uniform_16384_seed0"] = u
    c["uniform_12345_seed1"] = oracle.mt19937_keys(12345, 1)          # not a multiple of any tile
    c["refrng_4096_seed0"] = oracle.ref_rng_keys(4096, 0)             # the harness's own stream (libstdc++)
    c["dup7_8192"] = (oracle.mt19937_keys(8192, 2) % 7).astype(np.uint32) * np.uint32(0x01010101)
    c["allequal_5000"] = np.full(5000, 0xDEADBEEF, dtype=np.uint32)
    c["allzero_4097"] = np.zeros(4097, dtype=np.uint32)
    c["allmax_4099"] = np.full(4099, 0xFFFFFFFF, dtype=np.uint32)    # collides with tail padding
    c["sorted_8192"] = np.sort(oracle.mt19937_keys(8192, 3))
    c["reverse_8192"] = np.sort(oracle.mt19937_keys(8192, 4))[::-1].copy()
    c["lowbits_6000"] = (oracle.mt19937_keys(6000, 5) & 0xFF).astype(np.uint32)        # one live digit
    c["highbits_6000"] = (oracle.mt19937_keys(6000, 6) & 0xFF000000).astype(np.uint32)
    c["n0"] = np.zeros(0, dtype=np.uint32)
    c["n1"] = np.array([42], dtype=np.uint32)
    c["n2"] = np.array([7, 3], dtype=np.uint32)
    c["n3"] = np.array([0xFFFFFFFF, 0, 0x80000000], dtype=np.uint32)
    for n in (63, 64, 65, 255, 256, 257, 1023, 1025):
        c[f"edge_{n}"] = oracle.mt19937_keys(n, 100 + n)
    return c


def main():
    oracle.build()
    assert oracle.ref_available(), "needs /root/reference (make -C oracle ref)"
    out = {}
    names = []
    for name, keys in cases().items():
        keys = np.ascontiguousarray(keys, dtype=np.uint32)
        expect = oracle.std_sort(keys)
        if keys.size:
            for r in (1, 2, 4, 8, 16):
                got = oracle.ref_lsd_sort(keys, r)
                assert np.array_equal(got, expect), (name, r)
        out[f"in__{name}"] = keys
        out[f"sorted__{name}"] = expect
        names.append(name)
    out["case_names"] = np.array(names)

    # Per-pass states of the reference's LSDRadixSortPass: state[g] = array after passes 0..g.
    src = cases()["uniform_12345_seed1"][:8192].copy()
    out["passes_in"] = src
    for r in (4, 8):
        cur = src.copy()
        states = []
        for g in range(32 // r):
            cur = oracle.ref_lsd_pass(cur, r, g)
            states.append(cur.copy())
        assert np.array_equal(states[-1], np.sort(src))
        out[f"passes_r{r}"] = np.stack(states)

    # Stage-level vectors from the reference's BuildHistogramsCPU / PrefixSum.
    hsrc = cases()["uniform_16384_seed0"][:8192].copy()
    out["hist_in"] = hsrc
    out["hist_block1024_r8_bg1"] = oracle.ref_build_histograms(hsrc, 1024, 8, 1)
    out["hist_block256_r4_bg5"] = oracle.ref_build_histograms(hsrc, 256, 4, 5)
    out["hist_block512_r2_bg9"] = oracle.ref_build_histograms(hsrc, 512, 2, 9)
    out["hist_block128_r1_bg18"] = oracle.ref_build_histograms(hsrc, 128, 1, 18)      # BenchmarkBuildHistogram.md:6
    scan_in = (oracle.mt19937_keys(1000, 9) % 1000).astype(np.uint32)
    out["scan_in"] = scan_in
    out["scan_out"] = oracle.ref_prefix_sum(scan_in)
    out["scan_kat_in"] = np.array([3, 1, 4, 1, 5], dtype=np.uint32)
    out["scan_kat_out"] = oracle.ref_prefix_sum(out["scan_kat_in"])

    # Key/value vectors: std::stable_sort by key (no reference counterpart).
    pk = (oracle.mt19937_keys(10000, 11) % 513).astype(np.uint32) * np.uint32(0x00800801)
    pv = np.arange(10000, dtype=np.uint32)
    ek, ev = oracle.std_stable_sort_pairs(pk, pv)
    out["pairs_keys"], out["pairs_vals"] = pk, pv
    out["pairs_sorted_keys"], out["pairs_sorted_vals"] = ek, ev

    # 64-bit keys and payloads (the reference is uint32 only, .cu:62): std::sort / std::stable_sort.
    lo, hi = oracle.mt19937_keys(9001, 21).astype(np.uint64), oracle.mt19937_keys(9001, 22).astype(np.uint64)
    k64 = (hi << np.uint64(32)) | lo
    k64[::7] = (k64[::7] & np.uint64(0xFFFFFFFF)) | np.uint64(5 << 32)              # many equal high words
    k64[1::11] = np.uint64(0xFFFFFFFFFFFFFFFF)                                      # collides with tail padding in both words
    k64[2::13] = k64[0]                                                             # exact duplicates
    out["u64_keys"] = k64
    out["u64_sorted"] = oracle.std_sort_u64(k64)
    v64 = (np.arange(9001, dtype=np.uint64) << np.uint64(33)) | np.uint64(1)
    rk, rv = oracle.std_stable_sort_records(k64, v64)
    out["u64_vals"], out["u64_records_keys"], out["u64_records_vals"] = v64, rk, rv
    k32 = (oracle.mt19937_keys(9001, 23) % 257).astype(np.uint32)
    rk32, rv32 = oracle.std_stable_sort_records(k32, v64)
    out["u32v64_keys"], out["u32v64_sorted_keys"], out["u32v64_sorted_vals"] = k32, rk32.astype(np.uint32), rv32

    # mt19937 known-answer: the 10000th output of default-seeded std::mt19937 (C++ standard).
    out["mt19937_kat"] = np.array([5489, 9999, 4123659995], dtype=np.uint64)

    np.savez_compressed(OUT, **out)
    print(f"wrote {OUT}: {os.path.getsize(OUT)} bytes, {len(names)} sort cases")


if __name__ == "__main__":
    main()
