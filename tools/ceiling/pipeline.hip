// pipeline.hip -- does a persistent workgroup that prefetches its next tile reach the memory floor of a pass?
//
// A pass-shaped kernel with the real LDS work of the rank-and-scatter kernel (one returning LDS add per key on
// wave-private counters, a digit-total scan, scattered LDS writes, linear read-back) but no chain and the
// "perfect" destination addresses of tools/ceiling/ceiling.hip, in three forms:
//   A  one tile per workgroup (grid = tiles), as the product kernel runs today
//   B  persistent workgroups (grid = CUs), tile t, t + grid, ...: load -> work -> store, nothing overlapped
//   C  persistent + the next tile's keys prefetched into registers before the current tile is worked on
// 1024 threads x 32 keys (32768-key tiles, one workgroup per CU).  Build: hipcc --offload-arch=gfx950 -O3 -o pipeline pipeline.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <functional>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int T = 1024, K = 32, TILE = T * K, W = T / 64, H = 256;
typedef __attribute__((address_space(3))) uint32_t lds_u32;

__global__ void fill_kernel(uint32_t* out, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        uint32_t h = (uint32_t)i * 0x9E3779B9u;
        h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16;
        out[i] = h;
    }
}

// the work of one tile whose keys are in registers: rank, totals, LDS reorder, stores to the "perfect" addresses
__device__ __forceinline__ void work_and_store(uint32_t (&key)[K], uint32_t tile, uint32_t* __restrict__ out, uint32_t per_digit,
                                               lds_u32* s_keys, volatile lds_u32* s_cnt, lds_u32* s_misc)
{
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int j = 0; j < H / 64; j++) s_cnt[wave * H + j * 64 + lane] = 0;
    // count (no-return adds), scan, then a second, returning add per key against counters that start at the wave's
    // base: its result IS the key's position in the tile -- no rank registers live across the scan
#pragma unroll
    for (int i = 0; i < K; i++)
        __hip_atomic_fetch_add((lds_u32*)&s_cnt[wave * H + (key[i] & 255u)], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    __syncthreads();
    uint32_t total = 0;
    if (tid < H) {
#pragma unroll
        for (int w = 0; w < W; w++) total += s_cnt[w * H + tid];
    }
    uint32_t incl = tid < H ? total : 0;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { uint32_t up = __shfl_up(incl, off, 64); if (lane >= off) incl += up; }
    if (lane == 63) s_misc[1 + wave] = incl;
    __syncthreads();
    uint32_t carry = 0;
#pragma unroll
    for (int w = 0; w < H / 64; w++) if ((uint32_t)w < wave) carry += s_misc[1 + w];
    if (tid < H) {
        uint32_t run = incl + carry - total;
#pragma unroll
        for (int w = 0; w < W; w++) { const uint32_t c = s_cnt[w * H + tid]; s_cnt[w * H + tid] = run; run += c; }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < K; i++) {
        const uint32_t pos = __hip_atomic_fetch_add((lds_u32*)&s_cnt[wave * H + (key[i] & 255u)], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        s_keys[pos] = key[i];
    }
    __syncthreads();
    constexpr uint32_t RUN = TILE / 256;
    uint32_t pd = per_digit;
    asm volatile("" : "+s"(pd));   // per iteration: keeps the 32 destination addresses from being hoisted out of the tile loop
    uint32_t tq = tid;
    asm volatile("" : "+v"(tq));
#pragma unroll
    for (int s = 0; s < K; s++) {
        const uint32_t q = s * T + tq;
        const uint32_t k = s_keys[q];
        const uint32_t d = q / RUN;   // where a uniform tile's digit-d run would sit
        out[(size_t)d * pd + (size_t)tile * RUN + (q % RUN)] = k;
    }
}

__device__ __forceinline__ void load_tile(uint32_t (&key)[K], const uint32_t* __restrict__ in, uint32_t tile)
{
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t* src = in + (size_t)tile * TILE + wave * (64 * K) + lane;
#pragma unroll
    for (int i = 0; i < K; i++) key[i] = src[i * 64];   // one 64-bit base, constant offsets
}

template <int MODE>
__global__ void __launch_bounds__(T) pass_kernel(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, uint32_t n, uint32_t tiles)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    lds_u32* s_keys = (lds_u32*)smem;
    volatile lds_u32* s_cnt = (volatile lds_u32*)(s_keys + TILE);
    lds_u32* s_misc = (lds_u32*)(s_keys + TILE + W * H);
    const uint32_t per_digit = n / 256;
    if (MODE == 0) {
        uint32_t key[K];
        load_tile(key, in, blockIdx.x);
        work_and_store(key, blockIdx.x, out, per_digit, s_keys, s_cnt, s_misc);
    } else if (MODE == 1) {
        for (uint32_t t = blockIdx.x; t < tiles; t += gridDim.x) {
            uint32_t key[K];
            load_tile(key, in, t);
            work_and_store(key, t, out, per_digit, s_keys, s_cnt, s_misc);
            __syncthreads();
        }
    } else {
        uint32_t cur[K], nxt[K];
        uint32_t t = blockIdx.x;
        if (t < tiles) load_tile(cur, in, t);
        while (t < tiles) {
            const uint32_t tn = t + gridDim.x;
            if (tn < tiles) load_tile(nxt, in, tn);
            work_and_store(cur, t, out, per_digit, s_keys, s_cnt, s_misc);
            __syncthreads();
#pragma unroll
            for (int i = 0; i < K; i++) cur[i] = nxt[i];
            t = tn;
        }
    }
}

static float time_ms(hipEvent_t a, hipEvent_t b, int reps, const std::function<void()>& fn)
{
    std::vector<float> ts;
    for (int i = 0; i < reps + 2; i++) {
        CHECK(hipEventRecord(a, 0));
        fn();
        CHECK(hipEventRecord(b, 0));
        CHECK(hipEventSynchronize(b));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, a, b));
        if (i >= 2) ts.push_back(ms);
    }
    std::sort(ts.begin(), ts.end());
    return ts[ts.size() / 2];
}

int main()
{
    const size_t n = (size_t)1 << 28;
    uint32_t *in, *out;
    CHECK(hipMalloc(&in, n * 4));
    CHECK(hipMalloc(&out, n * 4));
    hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, in, n);
    CHECK(hipDeviceSynchronize());
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    const uint32_t tiles = (uint32_t)(n / TILE);
    const size_t lds = (size_t)(TILE + W * H + 64) * 4;
    auto k0 = pass_kernel<0>; auto k1 = pass_kernel<1>; auto k2 = pass_kernel<2>;
    CHECK(hipFuncSetAttribute((const void*)k0, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CHECK(hipFuncSetAttribute((const void*)k1, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CHECK(hipFuncSetAttribute((const void*)k2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    for (int rep = 0; rep < 2; rep++) {
        float ms = time_ms(a, b, 9, [&] { hipLaunchKernelGGL(k0, dim3(tiles), dim3(T), lds, 0, in, out, (uint32_t)n, tiles); });
        printf("A one tile per workgroup      grid=%5u  %.4f ms  %.2f TB/s\n", tiles, ms, 2.0 * n * 4 / ms / 1e9);
        for (int grid : {256, 512}) {
            ms = time_ms(a, b, 9, [&] { hipLaunchKernelGGL(k1, dim3(grid), dim3(T), lds, 0, in, out, (uint32_t)n, tiles); });
            printf("B persistent, no prefetch     grid=%5d  %.4f ms  %.2f TB/s\n", grid, ms, 2.0 * n * 4 / ms / 1e9);
            ms = time_ms(a, b, 9, [&] { hipLaunchKernelGGL(k2, dim3(grid), dim3(T), lds, 0, in, out, (uint32_t)n, tiles); });
            printf("C persistent + reg prefetch   grid=%5d  %.4f ms  %.2f TB/s\n", grid, ms, 2.0 * n * 4 / ms / 1e9);
        }
        fflush(stdout);
    }
    CHECK(hipGetLastError());
    CHECK(hipDeviceSynchronize());
    return 0;
}
