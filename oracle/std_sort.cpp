// std_sort.cpp -- the C++ standard-library legs of the oracle.
//
// TEST INFRASTRUCTURE ONLY (see the header of lsd_oracle.c).  Never linked into the product.
//
//  * oracle_std_sort            a10: the reference's "STD Sort", std::sort(c, c + count),
//                               LSDRadixSort/LSDRadixSort.cu:97 -- the CPU baseline that
//                               BASELINE.json's metric names, one thread exactly as there.
//  * oracle_std_stable_sort_pairs   the unique answer for key/value input (stable by key);
//                               the reference has no pair sort.
//  * oracle_fill_mt19937        portable input generator: raw std::mt19937 outputs.  The
//                               reference's RNG (Utils.h:24-33: default_random_engine +
//                               uniform_int_distribution) is implementation-defined, so
//                               "same input" means the same bytes, not the same seed.
//  * oracle_time_*              steady_clock timings for bench.py's cpu_baseline leg; the
//                               reference's non-MSVC timer is time(NULL), 1 s resolution
//                               (Utils.cpp:46-59), and cannot be used.
#include <algorithm>
#include <chrono>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <numeric>
#include <random>
#include <vector>

#define ORACLE_API extern "C" __attribute__((visibility("default")))

extern "C" void oracle_lsd_sort(uint32_t* in, uint32_t* out, size_t count, uint32_t* histogram, int r);

ORACLE_API void oracle_std_sort(uint32_t* keys, size_t count)
{
    std::sort(keys, keys + count);
}

ORACLE_API void oracle_std_stable_sort_pairs(uint32_t* keys, uint32_t* vals, size_t count)
{
    std::vector<uint64_t> packed(count);
    // key in the high half, original index in the low half: a plain sort of the packed
    // words is a stable sort by key.
    for (size_t i = 0; i < count; i++) packed[i] = ((uint64_t)keys[i] << 32) | (uint64_t)(uint32_t)i;
    if (count <= 0xffffffffull) {
        std::sort(packed.begin(), packed.end());
        std::vector<uint32_t> v(vals, vals + count);
        for (size_t i = 0; i < count; i++) {
            keys[i] = (uint32_t)(packed[i] >> 32);
            vals[i] = v[(uint32_t)packed[i]];
        }
    } else {
        std::vector<size_t> idx(count);
        std::iota(idx.begin(), idx.end(), (size_t)0);
        std::stable_sort(idx.begin(), idx.end(), [&](size_t a, size_t b) { return keys[a] < keys[b]; });
        std::vector<uint32_t> k(keys, keys + count), v(vals, vals + count);
        for (size_t i = 0; i < count; i++) { keys[i] = k[idx[i]]; vals[i] = v[idx[i]]; }
    }
}

// 64-bit keys / payloads (no reference counterpart, .cu:62 is uint32 only): the unique answers are std::sort on
// uint64 and std::stable_sort by key over (key, payload) records.  `vals` may be null (keys only).
ORACLE_API void oracle_std_sort_u64(uint64_t* keys, size_t count)
{
    std::sort(keys, keys + count);
}

ORACLE_API void oracle_std_stable_sort_records(uint64_t* keys, uint64_t* vals, size_t count)
{
    std::vector<size_t> idx(count);
    std::iota(idx.begin(), idx.end(), (size_t)0);
    std::stable_sort(idx.begin(), idx.end(), [&](size_t a, size_t b) { return keys[a] < keys[b]; });
    std::vector<uint64_t> k(keys, keys + count), v(vals, vals + count);
    for (size_t i = 0; i < count; i++) { keys[i] = k[idx[i]]; vals[i] = v[idx[i]]; }
}

ORACLE_API void oracle_fill_mt19937(uint32_t* out, size_t count, uint32_t seed)
{
    std::mt19937 gen(seed);
    for (size_t i = 0; i < count; i++) out[i] = (uint32_t)gen();
}

static double now_ms()
{
    using clk = std::chrono::steady_clock;
    return std::chrono::duration<double, std::milli>(clk::now().time_since_epoch()).count();
}

// Sorts `keys` in place with std::sort on the calling thread; returns elapsed milliseconds.
ORACLE_API double oracle_time_std_sort(uint32_t* keys, size_t count)
{
    const double t0 = now_ms();
    std::sort(keys, keys + count);
    return now_ms() - t0;
}

// Sorts `keys` with the restated reference CPU LSD sort (.cu:62-69) at radix `r`;
// `scratch` is count words.  Returns elapsed milliseconds (allocation excluded).
ORACLE_API double oracle_time_lsd_sort(uint32_t* keys, uint32_t* scratch, size_t count, int r)
{
    std::vector<uint32_t> histogram((size_t)1 << r);
    const double t0 = now_ms();
    oracle_lsd_sort(keys, scratch, count, histogram.data(), r);
    return now_ms() - t0;
}
