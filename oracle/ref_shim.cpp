// ref_shim.cpp -- C-ABI doorway onto the reference's OWN CPU functions.
//
// TEST INFRASTRUCTURE ONLY.  This file contains no reference code: it declares the four
// CUDA-free functions the reference defines in LSDRadixSort/LSDRadixSort.cu and forwards to
// them.  oracle/Makefile (target `ref`) compiles those functions from the reference tree in
// place -- the line ranges .cu:25-69, .cu:128-139, .cu:643-658 are streamed to g++ on stdin,
// plus LSDRadixSort/Utils.cpp as-is -- and links them with this shim into
// oracle/_ref/libref_lsd.so.  The whole .cu cannot be built here (it includes
// cuda_runtime.h through CudaUtils.h and there is no CUDA toolkit; no stand-in headers are
// written), so only its host-only functions are.  Outputs live under oracle/_ref/ only,
// which is git-ignored; no reference source is copied into this repository.
#include <cstddef>
#include <cstdint>
#include <random>

// Declarations of the reference's definitions (signatures as at the cited lines).
void LSDRadixSortPass(uint32_t* in, uint32_t* out, int count, uint32_t* histogram, int r, int bit_group); // .cu:25
void LSDRadixSort(uint32_t* in, uint32_t* out, int count, uint32_t* histogram, int r);                    // .cu:62
void PrefixSum(uint32_t* a, int count);                                                                   // .cu:128
void BuildHistogramsCPU(uint32_t* a, uint32_t* h, int count, int r, int bit_group, int grid, int block);  // .cu:643

#define REF_API extern "C" __attribute__((visibility("default")))

REF_API void ref_lsd_pass(uint32_t* in, uint32_t* out, int count, uint32_t* histogram, int r, int bit_group)
{
    LSDRadixSortPass(in, out, count, histogram, r, bit_group);
}

REF_API void ref_lsd_sort(uint32_t* in, uint32_t* out, int count, uint32_t* histogram, int r)
{
    LSDRadixSort(in, out, count, histogram, r);
}

REF_API void ref_prefix_sum(uint32_t* a, int count)
{
    PrefixSum(a, count);
}

// BuildHistogramsCPU accumulates into h (.cu:655); the caller zeroes h first.
REF_API void ref_build_histograms(uint32_t* a, uint32_t* h, int count, int r, int bit_group, int grid, int block)
{
    BuildHistogramsCPU(a, h, count, r, bit_group, grid, block);
}

// The reference harness's input stream, RNG(seed, min, max).Get() (Utils.h:24-33,
// Utils.cpp:12-15) as this platform's libstdc++ defines it.  Implementation-defined: MSVC,
// where the published numbers were taken, yields different bytes.
#include "Utils.h"
REF_API void ref_rng_fill(uint32_t* out, size_t count, unsigned seed, uint32_t lo, uint32_t hi)
{
    RNG rng(seed, lo, hi);
    for (size_t i = 0; i < count; i++) out[i] = rng.Get();
}
