// ceiling.hip -- what this box's memory system gives the access shapes a radix pass is made of.
//
// hipMemcpy D2D (5.2 TB/s) is not the ceiling a pass is held against; the
// guide records 6.29 TB/s for a float4 copy.  This tool times, on n = 2^28 uint32 (1 GiB in, 1 GiB out):
//   copy16      : 16 B/lane loads and stores, grid-stride, several grid sizes
//   copy4       : 4 B/lane loads and stores (the shape of the pass's key loads and stores)
//   read16/4    : read only (upfront histogram's ceiling)
//   write16/4   : write only
//   tile_scatter: every workgroup reads one tile linearly (4 B/lane, wave-striped exactly as the
//                 rank-and-scatter kernel does) and stores it as 256 runs of tile/256 keys each, at the
//                 addresses a pass on perfectly uniform digits would use -- no LDS, no ranking, no chain:
//                 the memory-system floor of a pass.  Tiles are dealt to XCDs in chunks of C consecutive
//                 tiles (C = 0: round-robin), tile sizes 8192..65536 keys.
// Build: hipcc --offload-arch=gfx950 -O3 -o ceiling ceiling.hip      Run: ./ceiling [log2_n]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <functional>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void __launch_bounds__(256) copy16_kernel(const uint4* __restrict__ in, uint4* __restrict__ out, size_t n16)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) out[i] = in[i];
}

// four 16-byte loads in flight per lane
__global__ void __launch_bounds__(256) copy16x4_kernel(const uint4* __restrict__ in, uint4* __restrict__ out, size_t n16)
{
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * stride < n16; i += 4 * stride) {
        uint4 a = in[i], b = in[i + stride], c = in[i + 2 * stride], d = in[i + 3 * stride];
        out[i] = a; out[i + stride] = b; out[i + 2 * stride] = c; out[i + 3 * stride] = d;
    }
    for (; i < n16; i += stride) out[i] = in[i];
}

// one workgroup copies a contiguous block (the shape of a tile), 16 B per lane
__global__ void __launch_bounds__(1024) copy16_block_kernel(const uint4* __restrict__ in, uint4* __restrict__ out, uint32_t per_block16)
{
    const size_t base = (size_t)blockIdx.x * per_block16;
    for (uint32_t i = threadIdx.x; i < per_block16; i += 1024) out[base + i] = in[base + i];
}

template <int K>
__global__ void __launch_bounds__(1024) copy4_tile_kernel(const uint32_t* __restrict__ in, uint32_t* __restrict__ out)
{
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const size_t first = (size_t)blockIdx.x * (1024 * K) + wave * (64 * K) + lane;
    uint32_t k[K];
#pragma unroll
    for (int i = 0; i < K; i++) k[i] = in[first + i * 64];
#pragma unroll
    for (int i = 0; i < K; i++) out[first + i * 64] = k[i];
}

__global__ void __launch_bounds__(256) read16_kernel(const uint4* __restrict__ in, uint32_t* __restrict__ sink, size_t n16)
{
    uint32_t acc = 0;
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * stride < n16; i += 4 * stride) {
        uint4 a = in[i], b = in[i + stride], c = in[i + 2 * stride], d = in[i + 3 * stride];
        acc += a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y ^ b.z ^ b.w ^ c.x ^ c.y ^ c.z ^ c.w ^ d.x ^ d.y ^ d.z ^ d.w;
    }
    for (; i < n16; i += stride) { uint4 a = in[i]; acc += a.x ^ a.y ^ a.z ^ a.w; }
    if (acc == 0x12345678u) sink[0] = acc;
}

__global__ void __launch_bounds__(256) read4_kernel(const uint32_t* __restrict__ in, uint32_t* __restrict__ sink, size_t n)
{
    uint32_t acc = 0;
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + 7 * stride < n; i += 8 * stride) {
        uint32_t v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = in[i + u * stride];
#pragma unroll
        for (int u = 0; u < 8; u++) acc += v[u];
    }
    for (; i < n; i += stride) acc += in[i];
    if (acc == 0x12345678u) sink[0] = acc;
}

__global__ void __launch_bounds__(256) write16_kernel(uint4* __restrict__ out, size_t n16)
{
    const uint4 v = make_uint4(blockIdx.x, threadIdx.x, 3, 4);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) out[i] = v;
}

__global__ void __launch_bounds__(256) write4_kernel(uint32_t* __restrict__ out, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = (uint32_t)i;
}

// The pass's memory shape without its work.  T threads, K keys per thread, tile = T*K keys; bins = 256.
// Tile t (after XCD chunk remap) reads its keys wave-striped and writes position q of the tile to
//   dst = d * (n / 256) + t * RUN + (q % RUN),  d = q / RUN, RUN = tile / 256
// i.e. 256 contiguous runs of RUN keys, each behind the previous tile's run of the same digit.
template <int T, int K>
__global__ void __launch_bounds__(T) tile_scatter_kernel(const uint32_t* __restrict__ in, uint32_t* __restrict__ out,
                                                         uint32_t n, uint32_t num_tiles, uint32_t chunk)
{
    constexpr uint32_t TILE = T * K;
    constexpr uint32_t RUN = TILE / 256;
    uint32_t tile = blockIdx.x;
    if (chunk) {
        const uint32_t group = 8u * chunk;
        const uint32_t g0 = (tile / group) * group;
        if (g0 + group <= num_tiles) {
            const uint32_t k = tile - g0;
            tile = g0 + (k % 8u) * chunk + (k / 8u);
        }
    }
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t* src = in + (size_t)tile * TILE + wave * (64 * K) + lane;
    uint32_t key[K];
#pragma unroll
    for (int i = 0; i < K; i++) key[i] = src[i * 64];
    const uint32_t per_digit = n / 256;
#pragma unroll
    for (int s = 0; s < K; s++) {
        const uint32_t q = s * T + threadIdx.x;
        const uint32_t d = q / RUN;
        out[(size_t)d * per_digit + (size_t)tile * RUN + (q % RUN)] = key[s];
    }
}

// Same through LDS: keys are written to LDS at a permuted position and read back linearly, then stored as above
// (adds the LDS round trip a real pass makes, still no ranking atomics and no chain).
template <int T, int K>
__global__ void __launch_bounds__(T) tile_scatter_lds_kernel(const uint32_t* __restrict__ in, uint32_t* __restrict__ out,
                                                             uint32_t n, uint32_t num_tiles, uint32_t chunk)
{
    constexpr uint32_t TILE = T * K;
    constexpr uint32_t RUN = TILE / 256;
    extern __shared__ uint32_t s_keys[];
    uint32_t tile = blockIdx.x;
    if (chunk) {
        const uint32_t group = 8u * chunk;
        const uint32_t g0 = (tile / group) * group;
        if (g0 + group <= num_tiles) {
            const uint32_t k = tile - g0;
            tile = g0 + (k % 8u) * chunk + (k / 8u);
        }
    }
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t* src = in + (size_t)tile * TILE + wave * (64 * K) + lane;
    uint32_t key[K];
#pragma unroll
    for (int i = 0; i < K; i++) key[i] = src[i * 64];
#pragma unroll
    for (int i = 0; i < K; i++) {
        const uint32_t p = wave * (64 * K) + i * 64 + lane;
        s_keys[(p * 2654435761u) % TILE] = key[i];   // odd multiplier mod 2^k: a permutation
    }
    __syncthreads();
    const uint32_t per_digit = n / 256;
#pragma unroll
    for (int s = 0; s < K; s++) {
        const uint32_t q = s * T + threadIdx.x;
        const uint32_t d = q / RUN;
        out[(size_t)d * per_digit + (size_t)tile * RUN + (q % RUN)] = s_keys[q];
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

int main(int argc, char** argv)
{
    const int log2n = argc > 1 ? atoi(argv[1]) : 28;
    const size_t n = (size_t)1 << log2n;
    const size_t bytes = n * 4;
    uint32_t *in, *out, *sink;
    CHECK(hipMalloc(&in, bytes));
    CHECK(hipMalloc(&out, bytes));
    CHECK(hipMalloc(&sink, 256));
    CHECK(hipMemset(in, 1, bytes));
    CHECK(hipMemset(out, 0, bytes));
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    const int reps = 9;
    auto report = [&](const char* name, int grid, float ms, double moved) {
        printf("%-34s grid=%7d  %.4f ms  %.2f TB/s\n", name, grid, ms, moved / ms / 1e9);
        fflush(stdout);
    };
    {
        float ms = time_ms(a, b, reps, [&] { CHECK(hipMemcpyAsync(out, in, bytes, hipMemcpyDeviceToDevice, 0)); });
        report("hipMemcpy D2D", 0, ms, 2.0 * bytes);
    }
    for (int grid : {1024, 2048, 4096, 8192, 16384, 65536}) {
        float ms = time_ms(a, b, reps, [&] { hipLaunchKernelGGL(copy16_kernel, dim3(grid), dim3(256), 0, 0, (const uint4*)in, (uint4*)out, n / 4); });
        report("copy16 grid-stride", grid, ms, 2.0 * bytes);
    }
    for (int grid : {512, 1024, 2048, 4096, 8192}) {
        float ms = time_ms(a, b, reps, [&] { hipLaunchKernelGGL(copy16x4_kernel, dim3(grid), dim3(256), 0, 0, (const uint4*)in, (uint4*)out, n / 4); });
        report("copy16x4 grid-stride", grid, ms, 2.0 * bytes);
    }
    for (uint32_t per : {8192u, 16384u, 32768u, 65536u}) {   // keys per block
        const int grid = (int)(n / per);
        float ms = time_ms(a, b, reps, [&] { hipLaunchKernelGGL(copy16_block_kernel, dim3(grid), dim3(1024), 0, 0, (const uint4*)in, (uint4*)out, per / 4); });
        char nm[64]; snprintf(nm, sizeof nm, "copy16 block of %u keys", per);
        report(nm, grid, ms, 2.0 * bytes);
    }
    {
        float ms = time_ms(a, b, reps, [&] { hipLaunchKernelGGL((copy4_tile_kernel<32>), dim3(n / 32768), dim3(1024), 0, 0, in, out); });
        report("copy4 tile 1024x32", (int)(n / 32768), ms, 2.0 * bytes);
        ms = time_ms(a, b, reps, [&] { hipLaunchKernelGGL((copy4_tile_kernel<16>), dim3(n / 16384), dim3(1024), 0, 0, in, out); });
        report("copy4 tile 1024x16", (int)(n / 16384), ms, 2.0 * bytes);
        ms = time_ms(a, b, reps, [&] { hipLaunchKernelGGL((copy4_tile_kernel<8>), dim3(n / 8192), dim3(1024), 0, 0, in, out); });
        report("copy4 tile 1024x8", (int)(n / 8192), ms, 2.0 * bytes);
    }
    for (int grid : {1024, 2048, 4096, 8192}) {
        float ms = time_ms(a, b, reps, [&] { hipLaunchKernelGGL(read16_kernel, dim3(grid), dim3(256), 0, 0, (const uint4*)in, sink, n / 4); });
        report("read16", grid, ms, 1.0 * bytes);
    }
    for (int grid : {2048, 4096, 8192}) {
        float ms = time_ms(a, b, reps, [&] { hipLaunchKernelGGL(read4_kernel, dim3(grid), dim3(256), 0, 0, in, sink, n); });
        report("read4 (8 in flight)", grid, ms, 1.0 * bytes);
    }
    for (int grid : {2048, 8192, 65536}) {
        float ms = time_ms(a, b, reps, [&] { hipLaunchKernelGGL(write16_kernel, dim3(grid), dim3(256), 0, 0, (uint4*)out, n / 4); });
        report("write16", grid, ms, 1.0 * bytes);
        ms = time_ms(a, b, reps, [&] { hipLaunchKernelGGL(write4_kernel, dim3(grid), dim3(256), 0, 0, out, n); });
        report("write4", grid, ms, 1.0 * bytes);
    }
    for (uint32_t chunk : {0u, 4u, 16u, 64u}) {
        char nm[64];
        {
            const uint32_t tiles = (uint32_t)(n / 32768);
            float ms = time_ms(a, b, reps, [&] { hipLaunchKernelGGL((tile_scatter_kernel<1024, 32>), dim3(tiles), dim3(1024), 0, 0, in, out, (uint32_t)n, tiles, chunk); });
            snprintf(nm, sizeof nm, "tile_scatter 1024x32 C=%u", chunk);
            report(nm, (int)tiles, ms, 2.0 * bytes);
        }
        {
            const uint32_t tiles = (uint32_t)(n / 16384);
            float ms = time_ms(a, b, reps, [&] { hipLaunchKernelGGL((tile_scatter_kernel<512, 32>), dim3(tiles), dim3(512), 0, 0, in, out, (uint32_t)n, tiles, chunk); });
            snprintf(nm, sizeof nm, "tile_scatter 512x32 C=%u", chunk);
            report(nm, (int)tiles, ms, 2.0 * bytes);
        }
        {
            const uint32_t tiles = (uint32_t)(n / 16384);
            float ms = time_ms(a, b, reps, [&] { hipLaunchKernelGGL((tile_scatter_kernel<1024, 16>), dim3(tiles), dim3(1024), 0, 0, in, out, (uint32_t)n, tiles, chunk); });
            snprintf(nm, sizeof nm, "tile_scatter 1024x16 C=%u", chunk);
            report(nm, (int)tiles, ms, 2.0 * bytes);
        }
        {
            const uint32_t tiles = (uint32_t)(n / 8192);
            float ms = time_ms(a, b, reps, [&] { hipLaunchKernelGGL((tile_scatter_kernel<512, 16>), dim3(tiles), dim3(512), 0, 0, in, out, (uint32_t)n, tiles, chunk); });
            snprintf(nm, sizeof nm, "tile_scatter 512x16 C=%u", chunk);
            report(nm, (int)tiles, ms, 2.0 * bytes);
        }
    }
    for (uint32_t chunk : {0u, 16u}) {
        char nm[64];
        {
            const uint32_t tiles = (uint32_t)(n / 32768);
            auto k = tile_scatter_lds_kernel<1024, 32>;
            CHECK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 32768 * 4));
            float ms = time_ms(a, b, reps, [&] { hipLaunchKernelGGL(k, dim3(tiles), dim3(1024), 32768 * 4, 0, in, out, (uint32_t)n, tiles, chunk); });
            snprintf(nm, sizeof nm, "tile_scatter+LDS 1024x32 C=%u", chunk);
            report(nm, (int)tiles, ms, 2.0 * bytes);
        }
        {
            const uint32_t tiles = (uint32_t)(n / 16384);
            auto k = tile_scatter_lds_kernel<512, 32>;
            CHECK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 16384 * 4));
            float ms = time_ms(a, b, reps, [&] { hipLaunchKernelGGL(k, dim3(tiles), dim3(512), 16384 * 4, 0, in, out, (uint32_t)n, tiles, chunk); });
            snprintf(nm, sizeof nm, "tile_scatter+LDS 512x32 C=%u", chunk);
            report(nm, (int)tiles, ms, 2.0 * bytes);
        }
    }
    CHECK(hipDeviceSynchronize());
    return 0;
}
