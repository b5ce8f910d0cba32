"""CPU suite, part 1: pin the oracle.

The oracle (oracle/lsd_oracle.c, oracle/std_sort.cpp) is a restatement of the reference's CPU
path; before anything trusts it, it is checked against
  * the committed golden vectors (tests/golden/lsd_golden.npz), which were produced by the
    reference's own functions compiled in place (tests/golden/make_golden.py), and
  * that reference build itself (oracle/_ref/libref_lsd.so) when present on this machine.
Mirrors the reference's own checks: CPU LSD == std::sort (LSDRadixSort.cu:120), per-block scan
vs PrefixSum (.cu:257), histograms vs BuildHistogramsCPU (.cu:785).
"""
import numpy as np
import pytest


def test_mt19937_known_answer(oracle_mod, golden):
    seed, index, value = (int(x) for x in golden["mt19937_kat"])
    keys = oracle_mod.mt19937_keys(index + 1, seed)
    assert int(keys[index]) == value            # C++ standard: 10000th output of mt19937() is 4123659995


def test_get_r_bits(oracle_mod):
    # GET_R_BITS, Utils.h:22
    assert oracle_mod.get_r_bits(0xABCD1234, 8, 0) == 0x34
    assert oracle_mod.get_r_bits(0xABCD1234, 8, 3) == 0xAB
    assert oracle_mod.get_r_bits(0xABCD1234, 4, 5) == 0xC
    assert oracle_mod.get_r_bits(0xFFFFFFFF, 1, 31) == 1
    assert oracle_mod.get_r_bits(0x80000000, 16, 1) == 0x8000


@pytest.mark.parametrize("r", [1, 2, 4, 8, 16])
def test_oracle_sort_matches_golden(oracle_mod, golden, r):
    for name in golden["case_names"]:
        keys, expect = golden[f"in__{name}"], golden[f"sorted__{name}"]
        got = oracle_mod.lsd_sort(keys, r)
        assert oracle_mod.first_mismatch(got, expect) == expect.size, (name, r)


def test_std_sort_matches_golden(oracle_mod, golden):
    for name in golden["case_names"]:
        assert np.array_equal(oracle_mod.std_sort(golden[f"in__{name}"]), golden[f"sorted__{name}"]), name


@pytest.mark.parametrize("r", [4, 8])
def test_oracle_pass_states_match_reference_passes(oracle_mod, golden, r):
    cur = golden["passes_in"].copy()
    states = golden[f"passes_r{r}"]
    for g in range(32 // r):
        cur = oracle_mod.lsd_pass(cur, r, g)
        assert np.array_equal(cur, states[g]), (r, g)


@pytest.mark.parametrize("block,r,bg", [(1024, 8, 1), (256, 4, 5), (512, 2, 9), (128, 1, 18)])
def test_tile_histograms_match_reference(oracle_mod, golden, block, r, bg):
    got = oracle_mod.tile_histograms(golden["hist_in"], block, r, bg)
    assert np.array_equal(got, golden[f"hist_block{block}_r{r}_bg{bg}"])


def test_exclusive_scan_matches_reference(oracle_mod, golden):
    assert np.array_equal(oracle_mod.exclusive_scan(golden["scan_kat_in"]), golden["scan_kat_out"])
    assert list(golden["scan_kat_out"]) == [0, 3, 4, 8, 9]
    assert np.array_equal(oracle_mod.exclusive_scan(golden["scan_in"]), golden["scan_out"])


def test_pairs_match_stable_sort(oracle_mod, golden):
    k, v = oracle_mod.lsd_sort_pairs(golden["pairs_keys"], golden["pairs_vals"], 8)
    assert np.array_equal(k, golden["pairs_sorted_keys"])
    assert np.array_equal(v, golden["pairs_sorted_vals"])
    k4, v4 = oracle_mod.lsd_sort_pairs(golden["pairs_keys"], golden["pairs_vals"], 4)
    assert np.array_equal(k4, k) and np.array_equal(v4, v)
    ks, vs = oracle_mod.std_stable_sort_pairs(golden["pairs_keys"], golden["pairs_vals"])
    assert np.array_equal(ks, k) and np.array_equal(vs, v)


@pytest.mark.parametrize("tile,r", [(32, 1), (64, 2), (128, 4), (256, 8), (1024, 8), (4096, 4), (8192, 8)])
def test_staged_restatement_sorts(oracle_mod, golden, tile, r):
    """The stage data flow of GPULSDRadixSort (.cu:845-905) restated on the CPU gives the sort."""
    for name in ("uniform_16384_seed0", "uniform_12345_seed1", "dup7_8192", "allmax_4099", "n1", "edge_257"):
        keys = golden[f"in__{name}"]
        assert np.array_equal(oracle_mod.staged_sort(keys, tile, r), golden[f"sorted__{name}"]), (name, tile, r)


def test_stage_tables_consistent(oracle_mod, golden):
    keys = golden["in__uniform_12345_seed1"]
    tile, r, bg = 1024, 8, 2
    h = oracle_mod.tile_histograms(keys, tile, r, bg)
    assert int(h.sum()) == keys.size
    local = oracle_mod.local_offsets(h, r)
    glob = oracle_mod.global_offsets(h, r)
    assert np.array_equal(local[:, 0], np.zeros(h.shape[0], dtype=np.uint32))
    # digit-major exclusive scan, checked against numpy on the transposed table
    flat = h.T.reshape(-1).astype(np.uint64)
    expect = (np.cumsum(flat) - flat).reshape(h.shape[1], h.shape[0]).T
    assert np.array_equal(glob.astype(np.uint64), expect)
    out = oracle_mod.rank_scatter(keys, local, glob, tile, r, bg)
    assert np.array_equal(out, oracle_mod.lsd_pass(keys, r, bg))


def test_digit_histograms(oracle_mod, golden):
    keys = golden["in__uniform_16384_seed0"]
    for r in (1, 2, 4, 8):
        dh = oracle_mod.digit_histograms(keys, r)
        assert dh.shape == (32 // r, 1 << r)
        for g in range(32 // r):
            col = oracle_mod.tile_histograms(keys, 1024, r, g).sum(axis=0)
            assert np.array_equal(dh[g], col.astype(np.uint64))


@pytest.mark.parametrize("msb_bits", [0, 1, 2, 3])
def test_msb_partition(oracle_mod, golden, msb_bits):
    keys = golden["in__uniform_12345_seed1"]
    out, counts = oracle_mod.msb_partition(keys, msb_bits)
    assert int(counts.sum()) == keys.size
    bucket = keys >> np.uint32(32 - msb_bits) if msb_bits else np.zeros_like(keys)
    order = np.argsort(bucket, kind="stable")
    assert np.array_equal(out, keys[order])
    assert np.array_equal(counts, np.bincount(bucket, minlength=1 << msb_bits).astype(np.uint64))


# ----------------------------------------------------------------------------- against the real reference
def _need_ref(oracle_mod):
    if not oracle_mod.ref_available():
        pytest.skip("oracle/_ref/libref_lsd.so not built (needs /root/reference; golden vectors cover this machine)")


@pytest.mark.parametrize("r", [1, 2, 4, 8, 16])
def test_oracle_equals_reference_build(oracle_mod, r):
    _need_ref(oracle_mod)
    for n, seed in ((1, 1), (2, 2), (1000, 3), (65536, 0), (100003, 4)):
        keys = oracle_mod.mt19937_keys(n, seed)
        assert np.array_equal(oracle_mod.lsd_sort(keys, r), oracle_mod.ref_lsd_sort(keys, r)), (n, r)
    skew = (oracle_mod.mt19937_keys(50000, 9) % 5).astype(np.uint32) << np.uint32(13)
    assert np.array_equal(oracle_mod.lsd_sort(skew, r), oracle_mod.ref_lsd_sort(skew, r))


def test_reference_rng_stream(oracle_mod, golden):
    _need_ref(oracle_mod)
    # RNG(0, 0, UINT32_MAX) on libstdc++; MSVC differs.
    assert list(oracle_mod.ref_rng_keys(4, 0)) == [282475248, 2617694917, 1457850877, 3262921810]
    assert np.array_equal(oracle_mod.ref_rng_keys(4096, 0), golden["in__refrng_4096_seed0"])


def test_golden_regenerates_identically(oracle_mod, golden):
    """Where the reference is present, re-derive a sample of the fixtures from it."""
    _need_ref(oracle_mod)
    keys = golden["in__uniform_16384_seed0"]
    assert np.array_equal(oracle_mod.ref_lsd_sort(keys, 8), golden["sorted__uniform_16384_seed0"])
    assert np.array_equal(oracle_mod.ref_build_histograms(golden["hist_in"], 1024, 8, 1), golden["hist_block1024_r8_bg1"])
    assert np.array_equal(oracle_mod.ref_prefix_sum(golden["scan_in"]), golden["scan_out"])


def test_u64_legs_against_golden_and_numpy(oracle_mod, golden):
    """The 64-bit legs of the oracle (std::sort on uint64, std::stable_sort over records; SURVEY 8f.4 -- the reference is
    uint32 only, .cu:62, so these are pinned by the committed vectors and by numpy's own stable sort)."""
    k = golden["u64_keys"]
    assert np.array_equal(oracle_mod.std_sort_u64(k), golden["u64_sorted"])
    assert np.array_equal(golden["u64_sorted"], np.sort(k))
    rk, rv = oracle_mod.std_stable_sort_records(k, golden["u64_vals"])
    order = np.argsort(k, kind="stable")
    assert np.array_equal(rk, k[order]) and np.array_equal(rv, golden["u64_vals"][order])
    assert np.array_equal(rk, golden["u64_records_keys"]) and np.array_equal(rv, golden["u64_records_vals"])
