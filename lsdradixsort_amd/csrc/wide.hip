// wide.hip -- 64-bit keys and 64-bit payloads over the 32-bit pass kernels.
//
// No reference counterpart: the reference sorts ascending uint32 keys only (LSDRadixSort.cu:62).  An LSD sort on a
// 64-bit key is an LSD sort on its low word followed by a stable LSD sort on its high word, so nothing new is needed
// in the pass kernels: the key/value kernel (the other word, or an index, rides as the payload) does all of it.
//
//   uint64 keys only            : split (lo[], hi[]) | pairs sort by lo carrying hi | pairs sort by hi carrying lo | merge.
//                                 8 passes at 8-bit digits, 16 B/key/pass -- what a native 64-bit-key pass would move --
//                                 plus the split and the merge (16 B/key each) and the two upfront histogram reads.
//   uint64 keys + 32/64-bit payloads, uint32 keys + 64-bit payloads: sort an index instead of the records --
//                                 idx = 0..n-1 rides through the pairs sorts (for 64-bit keys: by lo, then by the high
//                                 words gathered into that order) and the records are gathered once at the end.
// Everything is stream-ordered on the caller's stream and allocates nothing (workspace), like the 32-bit entries.
#define LSDSORT_BUILD 1
#include "../../include/lsdsort.h"

#include <hip/hip_runtime.h>

#include "lsd_kernels.hpp"

namespace lsd {
void set_last_hip_error(hipError_t e);
}

namespace {

constexpr size_t kAlign = 256;
size_t align_up(size_t x) { return (x + kAlign - 1) / kAlign * kAlign; }

#define W_HIP(expr)                           \
    do {                                      \
        hipError_t e__ = (expr);              \
        if (e__ != hipSuccess) {              \
            lsd::set_last_hip_error(e__);     \
            (void)hipGetLastError();          \
            return LSDSORT_ERR_HIP;           \
        }                                     \
    } while (0)
#define W_TRY(expr)                        \
    do {                                   \
        int s__ = (expr);                  \
        if (s__ != LSDSORT_OK) return s__; \
    } while (0)

constexpr int kThreads = 256;
uint32_t grid_for(size_t n, size_t per_thread = 4)
{
    const size_t blocks = (n + kThreads * per_thread - 1) / (kThreads * per_thread);
    return (uint32_t)(blocks < 1 ? 1 : (blocks > 65536 ? 65536 : blocks));
}

__global__ void __launch_bounds__(kThreads) split_u64_kernel(const uint2* __restrict__ in, uint32_t* __restrict__ lo,
                                                            uint32_t* __restrict__ hi, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (size_t)gridDim.x * kThreads) {
        const uint2 k = in[i];   // little-endian: x = low word
        lo[i] = k.x;
        if (hi) hi[i] = k.y;
    }
}

__global__ void __launch_bounds__(kThreads) merge_u64_kernel(const uint32_t* __restrict__ lo, const uint32_t* __restrict__ hi,
                                                            uint2* __restrict__ out, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (size_t)gridDim.x * kThreads)
        out[i] = make_uint2(lo[i], hi[i]);
}

__global__ void __launch_bounds__(kThreads) iota_kernel(uint32_t* __restrict__ out, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (size_t)gridDim.x * kThreads) out[i] = (uint32_t)i;
}

// out[i] = high word of src[idx[i]]
__global__ void __launch_bounds__(kThreads) gather_hi_kernel(const uint2* __restrict__ src, const uint32_t* __restrict__ idx,
                                                            uint32_t* __restrict__ out, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (size_t)gridDim.x * kThreads) out[i] = src[idx[i]].y;
}

template <typename V>
__global__ void __launch_bounds__(kThreads) gather_kernel(const V* __restrict__ src, const uint32_t* __restrict__ idx,
                                                         V* __restrict__ out, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (size_t)gridDim.x * kThreads) out[i] = src[idx[i]];
}

struct WideLayout {
    size_t a = 0, b = 0, c = 0;     // three uint32[n] arrays (lo / hi / idx / gathered high words, by entry)
    size_t rec_keys = 0;            // uint64[n]: gathered keys      (record forms)
    size_t rec_vals = 0;            // uint64[n]: gathered payloads  (record forms)
    size_t sort_ws = 0;             // workspace of the pairs sort
    size_t sort_ws_bytes = 0;
    size_t total = 0;
};

// key_bits 32 | 64, val_bits 0 | 32 | 64 (32/32 and 32/0 are the ordinary entries, not served here)
WideLayout make_wide_layout(size_t n, int radix_bits, int key_bits, int val_bits)
{
    WideLayout L;
    size_t off = 0;
    const size_t words = align_up(n * sizeof(uint32_t));
    L.a = off; off += words;
    L.b = off; off += words;
    if (val_bits != 0 && key_bits == 64) { L.c = off; off += words; }
    if (val_bits != 0) {
        L.rec_keys = off; off += align_up(n * sizeof(uint64_t));
        L.rec_vals = off; off += align_up(n * sizeof(uint64_t));
    }
    L.sort_ws = off;
    L.sort_ws_bytes = lsdsort_workspace_bytes(n, radix_bits, 1);
    off += align_up(L.sort_ws_bytes);
    L.total = off;
    return L;
}

bool wide_combo(int key_bits, int val_bits)
{
    return (key_bits == 64 && (val_bits == 0 || val_bits == 32 || val_bits == 64)) || (key_bits == 32 && val_bits == 64);
}

int check_common(const void* d_keys, void* ws, size_t ws_bytes, size_t n, int radix_bits, const WideLayout& L)
{
    if (n > LSDSORT_MAX_KEYS) return LSDSORT_ERR_TOO_LARGE;
    if (lsdsort_workspace_bytes(1, radix_bits, 1) == 0) return LSDSORT_ERR_INVALID_ARG;
    if (n == 0) return LSDSORT_OK;
    if (!d_keys) return LSDSORT_ERR_INVALID_ARG;
    if (!ws || (reinterpret_cast<uintptr_t>(ws) & (kAlign - 1)) || ws_bytes < L.total) return LSDSORT_ERR_WORKSPACE;
    return LSDSORT_OK;
}

}  // namespace

extern "C" {

size_t lsdsort_wide_workspace_bytes(size_t n, int radix_bits, int key_bits, int val_bits)
{
    if (!wide_combo(key_bits, val_bits) || n > LSDSORT_MAX_KEYS || lsdsort_workspace_bytes(1, radix_bits, 1) == 0) return 0;
    return make_wide_layout(n, radix_bits, key_bits, val_bits).total;
}

int lsdsort_u64_device(uint64_t* d_keys, void* d_workspace, size_t workspace_bytes, size_t n, int radix_bits, void* hip_stream)
{
    const WideLayout L = make_wide_layout(n, radix_bits, 64, 0);
    W_TRY(check_common(d_keys, d_workspace, workspace_bytes, n, radix_bits, L));
    if (n == 0) return LSDSORT_OK;
    W_TRY(lsdsort_prepare_device());
    hipStream_t s = static_cast<hipStream_t>(hip_stream);
    char* ws = static_cast<char*>(d_workspace);
    uint32_t* lo = reinterpret_cast<uint32_t*>(ws + L.a);
    uint32_t* hi = reinterpret_cast<uint32_t*>(ws + L.b);
    hipLaunchKernelGGL(split_u64_kernel, dim3(grid_for(n)), dim3(kThreads), 0, s, reinterpret_cast<const uint2*>(d_keys), lo, hi, n);
    W_HIP(hipGetLastError());
    // low word first, then a stable sort on the high word: sorted by (hi, lo) -- the LSD argument, one word at a time
    W_TRY(lsdsort_pairs_u32_device(lo, hi, ws + L.sort_ws, L.sort_ws_bytes, n, radix_bits, s));
    W_TRY(lsdsort_pairs_u32_device(hi, lo, ws + L.sort_ws, L.sort_ws_bytes, n, radix_bits, s));
    hipLaunchKernelGGL(merge_u64_kernel, dim3(grid_for(n)), dim3(kThreads), 0, s, lo, hi, reinterpret_cast<uint2*>(d_keys), n);
    W_HIP(hipGetLastError());
    return LSDSORT_OK;
}

// Records: keys of key_bits (32 | 64) with payloads of val_bits (32 | 64, not both 32); stable by key.
int lsdsort_records_device(void* d_keys, void* d_vals, int key_bits, int val_bits, void* d_workspace, size_t workspace_bytes,
                           size_t n, int radix_bits, void* hip_stream)
{
    if (!wide_combo(key_bits, val_bits) || val_bits == 0) return LSDSORT_ERR_INVALID_ARG;
    const WideLayout L = make_wide_layout(n, radix_bits, key_bits, val_bits);
    W_TRY(check_common(d_keys, d_workspace, workspace_bytes, n, radix_bits, L));
    if (n == 0) return LSDSORT_OK;
    if (!d_vals) return LSDSORT_ERR_INVALID_ARG;
    W_TRY(lsdsort_prepare_device());
    hipStream_t s = static_cast<hipStream_t>(hip_stream);
    char* ws = static_cast<char*>(d_workspace);
    uint32_t* word = reinterpret_cast<uint32_t*>(ws + L.a);     // the key word being sorted on
    uint32_t* idx = reinterpret_cast<uint32_t*>(ws + L.b);
    const uint32_t g = grid_for(n);
    hipLaunchKernelGGL(iota_kernel, dim3(g), dim3(kThreads), 0, s, idx, n);
    if (key_bits == 64) {
        uint32_t* high = reinterpret_cast<uint32_t*>(ws + L.c);
        hipLaunchKernelGGL(split_u64_kernel, dim3(g), dim3(kThreads), 0, s, reinterpret_cast<const uint2*>(d_keys), word, (uint32_t*)nullptr, n);
        W_HIP(hipGetLastError());
        W_TRY(lsdsort_pairs_u32_device(word, idx, ws + L.sort_ws, L.sort_ws_bytes, n, radix_bits, s));            // order by the low word
        hipLaunchKernelGGL(gather_hi_kernel, dim3(g), dim3(kThreads), 0, s, reinterpret_cast<const uint2*>(d_keys), idx, high, n);
        W_HIP(hipGetLastError());
        W_TRY(lsdsort_pairs_u32_device(high, idx, ws + L.sort_ws, L.sort_ws_bytes, n, radix_bits, s));            // stable, by the high word
    } else {
        W_HIP(hipMemcpyAsync(word, d_keys, n * sizeof(uint32_t), hipMemcpyDeviceToDevice, s));
        W_TRY(lsdsort_pairs_u32_device(word, idx, ws + L.sort_ws, L.sort_ws_bytes, n, radix_bits, s));
    }
    // idx[i] = where the i-th smallest record sits: gather the records once, copy them back
    void* rk = ws + L.rec_keys;
    void* rv = ws + L.rec_vals;
    if (key_bits == 64) {
        hipLaunchKernelGGL((gather_kernel<uint2>), dim3(g), dim3(kThreads), 0, s, static_cast<const uint2*>(d_keys), idx, static_cast<uint2*>(rk), n);
        W_HIP(hipGetLastError());
        W_HIP(hipMemcpyAsync(d_keys, rk, n * sizeof(uint64_t), hipMemcpyDeviceToDevice, s));
    } else {
        W_HIP(hipMemcpyAsync(d_keys, word, n * sizeof(uint32_t), hipMemcpyDeviceToDevice, s));   // the sorted key words themselves
    }
    if (val_bits == 64) {
        hipLaunchKernelGGL((gather_kernel<uint2>), dim3(g), dim3(kThreads), 0, s, static_cast<const uint2*>(d_vals), idx, static_cast<uint2*>(rv), n);
        W_HIP(hipGetLastError());
        W_HIP(hipMemcpyAsync(d_vals, rv, n * sizeof(uint64_t), hipMemcpyDeviceToDevice, s));
    } else {
        hipLaunchKernelGGL((gather_kernel<uint32_t>), dim3(g), dim3(kThreads), 0, s, static_cast<const uint32_t*>(d_vals), idx, static_cast<uint32_t*>(rv), n);
        W_HIP(hipGetLastError());
        W_HIP(hipMemcpyAsync(d_vals, rv, n * sizeof(uint32_t), hipMemcpyDeviceToDevice, s));
    }
    return LSDSORT_OK;
}

int lsdsort_wide_check_device(void* d_workspace, size_t n, int radix_bits, int key_bits, int val_bits, void* hip_stream)
{
    if (!d_workspace) return LSDSORT_ERR_WORKSPACE;
    if (!wide_combo(key_bits, val_bits) || lsdsort_workspace_bytes(1, radix_bits, 1) == 0) return LSDSORT_ERR_INVALID_ARG;
    const WideLayout L = make_wide_layout(n, radix_bits, key_bits, val_bits);
    return lsdsort_check_device(static_cast<char*>(d_workspace) + L.sort_ws, hip_stream);
}

}  // extern "C"
