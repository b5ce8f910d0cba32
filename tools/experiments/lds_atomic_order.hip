// Experiment (not product): does a returning LDS atomic add, issued by the 64 lanes of one
// wave instruction with colliding addresses, hand out its return values in lane order on
// gfx950?  Compares ds_add_rtn_u32 against the ballot-derived stable rank for many patterns
// under realistic occupancy.  Prints the number of disagreements.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

typedef __attribute__((address_space(3))) uint32_t lds_u32;

__device__ __forceinline__ uint32_t hash32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x;
}

template <int R>
__device__ __forceinline__ uint64_t match_ballot(uint32_t d) {
    uint64_t m = ~0ull;
#pragma unroll
    for (int b = 0; b < R; b++) { const bool bit = (d >> b) & 1u; const uint64_t bal = __ballot(bit); m &= bit ? bal : ~bal; }
    return m;
}

__global__ void __launch_bounds__(512) order_test(uint32_t iters, uint32_t seed, unsigned long long* mismatches,
                                                  unsigned long long* ops)
{
    __shared__ uint32_t s_cnt[8 * 256];
    __shared__ uint32_t s_ref[8 * 256];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    volatile lds_u32* cnt = (volatile lds_u32*)s_cnt + wave * 256;
    volatile lds_u32* ref = (volatile lds_u32*)s_ref + wave * 256;
    for (int j = lane; j < 256; j += 64) { cnt[j] = 0; ref[j] = 0; }
    unsigned long long bad = 0;
    for (uint32_t it = 0; it < iters; it++) {
        const uint32_t h = hash32(seed ^ (it * 0x9E3779B9u) ^ (blockIdx.x * 0x85EBCA6Bu) ^ (tid * 0xC2B2AE35u));
        const uint32_t mode = hash32(it ^ seed ^ blockIdx.x) % 7;     // wave-uniform-ish per block; fine
        uint32_t d;
        switch (mode) {
            case 0: d = h & 0xFF; break;            // 256 bins, few collisions
            case 1: d = h & 0x0F; break;            // 16 bins
            case 2: d = h & 0x03; break;            // 4 bins
            case 3: d = h & 0x01; break;            // 2 bins
            case 4: d = 7; break;                   // all collide
            case 5: d = (lane >> 2) & 0xFF; break;  // neighbours collide
            default: d = (h & 0xFF) * ((h >> 8) & 1); break;   // half the lanes on bin 0
        }
        const uint64_t peers = match_ballot<8>(d);
        const uint32_t r_in = __builtin_amdgcn_mbcnt_hi((uint32_t)(peers >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)peers, 0u));
        const uint32_t before = ref[d];
        const uint32_t old = __hip_atomic_fetch_add((lds_u32*)&cnt[d], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        if (old != before + r_in) bad++;
        if (r_in == 0) ref[d] = before + __builtin_popcountll(peers);
    }
    if (bad) atomicAdd(mismatches, bad);
    if (tid == 0) atomicAdd(ops, (unsigned long long)iters * 512ull);
}

int main() {
    unsigned long long *d_bad, *d_ops, h_bad = 0, h_ops = 0;
    hipMalloc(&d_bad, 8); hipMalloc(&d_ops, 8);
    hipMemset(d_bad, 0, 8); hipMemset(d_ops, 0, 8);
    for (int rep = 0; rep < 20; rep++) {
        hipLaunchKernelGGL(order_test, dim3(256 * 4), dim3(512), 0, 0, 4000u, 1234u + rep, d_bad, d_ops);
    }
    hipDeviceSynchronize();
    hipMemcpy(&h_bad, d_bad, 8, hipMemcpyDeviceToHost);
    hipMemcpy(&h_ops, d_ops, 8, hipMemcpyDeviceToHost);
    printf("lane-ops=%llu mismatches=%llu (%s)\n", h_ops, h_bad, hipGetErrorString(hipGetLastError()));
    return 0;
}
