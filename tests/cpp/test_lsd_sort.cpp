// C++ harness shaped like the reference's TestGPULSDRadixSort (LSDRadixSort/LSDRadixSort.cu:912-1030):
// fill -> CPU std::sort -> lsd::sort (H2D, device sort, D2H inside) -> element-wise compare.
// Built and run by tests/test_cpp_harness.py on the GPU box: g++ test_lsd_sort.cpp -llsdsort.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "lsdsort.hpp"

static int run(size_t count, int r, uint32_t seed)
{
    std::vector<uint32_t> a(count), expect;
    std::mt19937 gen(seed);                       // portable stream; the reference's RNG is not (Utils.h:24-33)
    for (auto& k : a) k = (uint32_t)gen();
    expect = a;
    std::sort(expect.begin(), expect.end());      // .cu:97
    lsd::sort(a.data(), a.size(), r);             // .cu:1001-1005
    for (size_t i = 0; i < count; i++)
        if (a[i] != expect[i]) {                  // CheckArrays, Utils.cpp:62-68, with a message instead of a crash
            std::fprintf(stderr, "mismatch at %zu: %u != %u (count %zu r %d)\n", i, a[i], expect[i], count, r);
            return 1;
        }
    std::vector<uint32_t> keys(count), vals(count);
    for (size_t i = 0; i < count; i++) { keys[i] = (uint32_t)gen() % 1000u; vals[i] = (uint32_t)i; }
    std::vector<std::pair<uint32_t, uint32_t>> pe(count);
    for (size_t i = 0; i < count; i++) pe[i] = {keys[i], vals[i]};
    std::stable_sort(pe.begin(), pe.end(), [](auto& x, auto& y) { return x.first < y.first; });
    lsd::sort_pairs(keys.data(), vals.data(), count);
    for (size_t i = 0; i < count; i++)
        if (keys[i] != pe[i].first || vals[i] != pe[i].second) {
            std::fprintf(stderr, "pair mismatch at %zu (count %zu)\n", i, count);
            return 1;
        }
    return 0;
}

int main()
{
    try {
        lsd::sort(nullptr, 0);
        for (int r : {1, 2, 4, 8})
            for (size_t count : {(size_t)1, (size_t)1000, (size_t)(1 << 20) + 7})
                if (run(count, r, (uint32_t)(count + r))) return 1;
        bool threw = false;
        try {
            uint32_t x = 0;
            lsd::sort(&x, 1, 7);                  // bad radix: an error, not a crash
        } catch (const lsd::sort_error& e) {
            threw = e.status() == LSDSORT_ERR_INVALID_ARG;
        }
        if (!threw) { std::fprintf(stderr, "expected LSDSORT_ERR_INVALID_ARG\n"); return 1; }
    } catch (const std::exception& e) {
        std::fprintf(stderr, "exception: %s\n", e.what());
        return 1;
    }
    std::printf("cpp harness ok\n");
    return 0;
}
