// wide.hip -- 64-bit keys and 64-bit payloads over the 32-bit pass kernels.
//
// No reference counterpart: the reference sorts ascending uint32 keys only (LSDRadixSort.cu:62).  An LSD sort on a
// 64-bit key is an LSD sort on its low word followed by a stable LSD sort on its high word, so nothing new is needed
// in the pass kernels: the key/value kernel (the other word, or an index, rides as the payload) does all of it.
//
//   uint64 keys only            : split (lo[], hi[]) | pairs sort by lo carrying hi | pairs sort by hi carrying lo | merge.
//                                 8 passes at 8-bit digits, 16 B/key/pass -- what a native 64-bit-key pass would move --
//                                 plus the split and the merge (16 B/key each) and the two upfront histogram reads.
//   uint64 keys + 32/64-bit payloads, uint32 keys + 64-bit payloads (records): every word that is not the key word being
//                                 sorted on rides through the passes as a payload array of its own (the key/value kernel
//                                 carries up to three: lsdsort_multi_u32_device) -- 64/64: by lo carrying (hi, vlo, vhi), then by
//                                 hi carrying (lo, vlo, vhi).  No gather: round 2 sorted an index and gathered the records at
//                                 random at the end, which cost more than the passes (13.5 ms of a 64/64 sort of 2^27 records).
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

struct WideLayout {
    size_t sticky = 0;              // u32: fault words of every sort inside the call, ORed together (each sort's opening
                                    // memset clears the shared sort workspace's own word, so the first sort's would be lost)
    size_t a = 0, b = 0;            // uint32[n]: low / high key words (64-bit keys)
    size_t c = 0, d = 0;            // uint32[n]: low / high payload words (64-bit payloads)
    size_t sort_ws = 0;             // workspace of the sorts inside
    size_t sort_ws_bytes = 0;
    int payloads = 1;               // payload arrays the sorts inside carry
    size_t total = 0;
};

// key_bits 32 | 64, val_bits 0 | 32 | 64 (32/32 and 32/0 are the ordinary entries, not served here)
WideLayout make_wide_layout(size_t n, int radix_bits, int key_bits, int val_bits)
{
    WideLayout L;
    size_t off = 0;
    L.sticky = off; off += kAlign;
    const size_t words = align_up(n * sizeof(uint32_t));
    if (key_bits == 64) {
        L.a = off; off += words;
        L.b = off; off += words;
    }
    if (val_bits == 64) {
        L.c = off; off += words;
        L.d = off; off += words;
    }
    // what rides with the key word: the other key word (64-bit keys) and the payload's words
    L.payloads = (key_bits == 64 ? 1 : 0) + val_bits / 32;
    L.sort_ws = off;
    L.sort_ws_bytes = lsdsort_workspace_bytes(n, radix_bits, L.payloads);
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
    uint32_t* sticky = reinterpret_cast<uint32_t*>(ws + L.sticky);
    const uint32_t* fault = reinterpret_cast<const uint32_t*>(ws + L.sort_ws);   // the sorts' fault word: first word of their workspace
    W_HIP(hipMemsetAsync(sticky, 0, sizeof(uint32_t), s));
    hipLaunchKernelGGL(split_u64_kernel, dim3(grid_for(n)), dim3(kThreads), 0, s, reinterpret_cast<const uint2*>(d_keys), lo, hi, n);
    W_HIP(hipGetLastError());
    // low word first, then a stable sort on the high word: sorted by (hi, lo) -- the LSD argument, one word at a time
    W_TRY(lsdsort_pairs_u32_device(lo, hi, ws + L.sort_ws, L.sort_ws_bytes, n, radix_bits, s));
    W_HIP(lsd::launch_keep_fault(sticky, fault, s));   // the next sort's memset clears that word
    W_TRY(lsdsort_pairs_u32_device(hi, lo, ws + L.sort_ws, L.sort_ws_bytes, n, radix_bits, s));
    W_HIP(lsd::launch_keep_fault(sticky, fault, s));
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
    const uint32_t g = grid_for(n);
    uint32_t* sticky = reinterpret_cast<uint32_t*>(ws + L.sticky);
    const uint32_t* fault = reinterpret_cast<const uint32_t*>(ws + L.sort_ws);
    W_HIP(hipMemsetAsync(sticky, 0, sizeof(uint32_t), s));
    // the words of the records as arrays of their own: 32-bit members are used where they lie
    uint32_t* klo = key_bits == 64 ? reinterpret_cast<uint32_t*>(ws + L.a) : static_cast<uint32_t*>(d_keys);
    uint32_t* khi = key_bits == 64 ? reinterpret_cast<uint32_t*>(ws + L.b) : nullptr;
    uint32_t* vlo = val_bits == 64 ? reinterpret_cast<uint32_t*>(ws + L.c) : static_cast<uint32_t*>(d_vals);
    uint32_t* vhi = val_bits == 64 ? reinterpret_cast<uint32_t*>(ws + L.d) : nullptr;
    if (key_bits == 64) {
        hipLaunchKernelGGL(split_u64_kernel, dim3(g), dim3(kThreads), 0, s, static_cast<const uint2*>(d_keys), klo, khi, n);
        W_HIP(hipGetLastError());
    }
    if (val_bits == 64) {
        hipLaunchKernelGGL(split_u64_kernel, dim3(g), dim3(kThreads), 0, s, static_cast<const uint2*>(d_vals), vlo, vhi, n);
        W_HIP(hipGetLastError());
    }
    // LSD over the key's words, low word first; everything else rides as payload arrays (stable: ties keep their order)
    {
        uint32_t* pay[3];
        int np = 0;
        if (khi) pay[np++] = khi;
        pay[np++] = vlo;
        if (vhi) pay[np++] = vhi;
        W_TRY(lsdsort_multi_u32_device(klo, pay, np, ws + L.sort_ws, L.sort_ws_bytes, n, radix_bits, s));
        W_HIP(lsd::launch_keep_fault(sticky, fault, s));
    }
    if (khi) {
        uint32_t* pay[3];
        int np = 0;
        pay[np++] = klo;
        pay[np++] = vlo;
        if (vhi) pay[np++] = vhi;
        W_TRY(lsdsort_multi_u32_device(khi, pay, np, ws + L.sort_ws, L.sort_ws_bytes, n, radix_bits, s));
        W_HIP(lsd::launch_keep_fault(sticky, fault, s));
        hipLaunchKernelGGL(merge_u64_kernel, dim3(g), dim3(kThreads), 0, s, klo, khi, static_cast<uint2*>(d_keys), n);
        W_HIP(hipGetLastError());
    }
    if (vhi) {
        hipLaunchKernelGGL(merge_u64_kernel, dim3(g), dim3(kThreads), 0, s, vlo, vhi, static_cast<uint2*>(d_vals), n);
        W_HIP(hipGetLastError());
    }
    return LSDSORT_OK;
}

int lsdsort_wide_check_device(void* d_workspace, size_t n, int radix_bits, int key_bits, int val_bits, void* hip_stream)
{
    if (!d_workspace) return LSDSORT_ERR_WORKSPACE;
    if (!wide_combo(key_bits, val_bits) || lsdsort_workspace_bytes(1, radix_bits, 1) == 0) return LSDSORT_ERR_INVALID_ARG;
    const WideLayout L = make_wide_layout(n, radix_bits, key_bits, val_bits);
    if (n == 0) return LSDSORT_OK;   // an empty call touches nothing
    // the sticky word holds every inner sort's fault word (lsdsort_check_device reads the first word of what it is given)
    return lsdsort_check_device(static_cast<char*>(d_workspace) + L.sticky, hip_stream);
}

}  // extern "C"
