// ceiling2.hip -- round 3: is 5.5 TB/s this part's copy rate, or only this tool's?  (VERDICT r2 "missing" item 1)
//
// tools/ceiling/ceiling.hip measured plain loads and stores only.  The MI355X guide records 6.29 TB/s for a float4 copy and
// 6.4-6.8 TB/s chip-wide reads with LDS-DMA / nt.  This tool adds, on n = 2^28 uint32 (1 GiB in, 1 GiB out):
//   copy16 nt      : 16 B/lane, four in flight, non-temporal loads / stores / both (__builtin_nontemporal_*)
//   read16 nt, write16 nt
//   glds           : global_load_lds_dwordx4 (LDS-DMA: no VGPR destination) -> ds_read_b128 -> global store; default and nt
//   read glds      : LDS-DMA reads only
//   tile_scatter nt: the pass's memory shape (ceiling.hip) with non-temporal key loads
//   clock          : every workgroup stamps s_memtime (shader clock) and s_memrealtime (100 MHz) around its work; the
//                    median ratio x 100 MHz is the clock the chip held while it streamed
//   working set    : the same copy on 2^22 .. 2^28 keys, 16 launches back to back (what the 256 MiB Infinity Cache gives
//                    a pass whose buffers fit it)
// Build: hipcc --offload-arch=gfx950 -O3 -o ceiling2 ceiling2.hip      Run: ./ceiling2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <functional>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef uint32_t v4u __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void global_cvoid;

template <bool LDNT, bool STNT>
__global__ void __launch_bounds__(256) copy16_kernel(const v4u* __restrict__ in, v4u* __restrict__ out, size_t n16, unsigned long long* clk)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * stride < n16; i += 4 * stride) {
        v4u v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) v[u] = LDNT ? __builtin_nontemporal_load(in + i + u * stride) : in[i + u * stride];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if (STNT) __builtin_nontemporal_store(v[u], out + i + u * stride);
            else out[i + u * stride] = v[u];
        }
    }
    for (; i < n16; i += stride) out[i] = in[i];
    if (clk) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
        if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
    }
}

template <bool NT>
__global__ void __launch_bounds__(256) read16_kernel(const v4u* __restrict__ in, uint32_t* __restrict__ sink, size_t n16)
{
    uint32_t acc = 0;
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * stride < n16; i += 4 * stride) {
        v4u v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) v[u] = NT ? __builtin_nontemporal_load(in + i + u * stride) : in[i + u * stride];
#pragma unroll
        for (int u = 0; u < 4; u++) acc += v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    for (; i < n16; i += stride) { v4u a = in[i]; acc += a.x ^ a.y ^ a.z ^ a.w; }
    if (acc == 0x12345678u) sink[0] = acc;
}

template <bool NT>
__global__ void __launch_bounds__(256) write16_kernel(v4u* __restrict__ out, size_t n16)
{
    v4u v;
    v.x = blockIdx.x; v.y = threadIdx.x; v.z = 3; v.w = 4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
        if (NT) __builtin_nontemporal_store(v, out + i);
        else out[i] = v;
    }
}

// LDS-DMA copy: a wave owns SLOTS x 1 KiB of LDS; per step it issues SLOTS global_load_lds_dwordx4 (1 KiB each, lane l's
// 16 bytes land at base + 16 l), waits, reads them back with ds_read_b128 and stores them.  AUX = 0 default policy, 2 = nt.
template <int SLOTS, int AUX, bool STORE>
__global__ void __launch_bounds__(256) glds_copy_kernel(const v4u* __restrict__ in, v4u* __restrict__ out, size_t n16, uint32_t* sink)
{
    extern __shared__ __attribute__((aligned(16))) v4u s_buf[];   // [waves][SLOTS][64]
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    v4u* mine = s_buf + (size_t)wave * SLOTS * 64;
    const size_t waves = (size_t)gridDim.x * 4;
    const size_t w = (size_t)blockIdx.x * 4 + wave;
    uint32_t acc = 0;
    // wave w takes chunks of SLOTS*64 vectors, strided over all waves
    for (size_t c = w * (SLOTS * 64); c + SLOTS * 64 <= n16; c += waves * (SLOTS * 64)) {
#pragma unroll
        for (int s = 0; s < SLOTS; s++)
            __builtin_amdgcn_global_load_lds((global_cvoid*)(in + c + s * 64 + lane), (lds_void*)(mine + s * 64), 16, 0, AUX);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (STORE) {
#pragma unroll
            for (int s = 0; s < SLOTS; s++) out[c + s * 64 + lane] = mine[s * 64 + lane];
        } else {
            acc += mine[lane].x;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the slots are free again
    }
    if (!STORE && acc == 0x12345678u) sink[0] = acc;
}

template <int T, int K, bool NT>
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
    for (int i = 0; i < K; i++) key[i] = NT ? __builtin_nontemporal_load(src + i * 64) : src[i * 64];
    const uint32_t per_digit = n / 256;
#pragma unroll
    for (int s = 0; s < K; s++) {
        const uint32_t q = s * T + threadIdx.x;
        const uint32_t d = q / RUN;
        out[(size_t)d * per_digit + (size_t)tile * RUN + (q % RUN)] = key[s];
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
    unsigned long long* clk;
    CHECK(hipMalloc(&in, bytes));
    CHECK(hipMalloc(&out, bytes));
    CHECK(hipMalloc(&sink, 256));
    CHECK(hipMalloc(&clk, 2 * 65536 * sizeof(unsigned long long)));
    CHECK(hipMemset(in, 1, bytes));
    CHECK(hipMemset(out, 0, bytes));
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    const int reps = 9;
    auto report = [&](const char* name, int grid, float ms, double moved) {
        printf("%-44s grid=%7d  %.4f ms  %.2f TB/s\n", name, grid, ms, moved / ms / 1e9);
        fflush(stdout);
    };
    const v4u* in16 = (const v4u*)in;
    v4u* out16 = (v4u*)out;
    for (int grid : {1024, 2048, 4096}) {
        float ms = time_ms(a, b, reps, [&] { hipLaunchKernelGGL((copy16_kernel<false, false>), dim3(grid), dim3(256), 0, 0, in16, out16, n / 4, nullptr); });
        report("copy16x4 plain", grid, ms, 2.0 * bytes);
        ms = time_ms(a, b, reps, [&] { hipLaunchKernelGGL((copy16_kernel<true, false>), dim3(grid), dim3(256), 0, 0, in16, out16, n / 4, nullptr); });
        report("copy16x4 nt loads", grid, ms, 2.0 * bytes);
        ms = time_ms(a, b, reps, [&] { hipLaunchKernelGGL((copy16_kernel<false, true>), dim3(grid), dim3(256), 0, 0, in16, out16, n / 4, nullptr); });
        report("copy16x4 nt stores", grid, ms, 2.0 * bytes);
        ms = time_ms(a, b, reps, [&] { hipLaunchKernelGGL((copy16_kernel<true, true>), dim3(grid), dim3(256), 0, 0, in16, out16, n / 4, nullptr); });
        report("copy16x4 nt loads + nt stores", grid, ms, 2.0 * bytes);
    }
    for (int grid : {2048, 4096}) {
        float ms = time_ms(a, b, reps, [&] { hipLaunchKernelGGL((read16_kernel<false>), dim3(grid), dim3(256), 0, 0, in16, sink, n / 4); });
        report("read16x4 plain", grid, ms, 1.0 * bytes);
        ms = time_ms(a, b, reps, [&] { hipLaunchKernelGGL((read16_kernel<true>), dim3(grid), dim3(256), 0, 0, in16, sink, n / 4); });
        report("read16x4 nt", grid, ms, 1.0 * bytes);
        ms = time_ms(a, b, reps, [&] { hipLaunchKernelGGL((write16_kernel<false>), dim3(grid), dim3(256), 0, 0, out16, n / 4); });
        report("write16 plain", grid, ms, 1.0 * bytes);
        ms = time_ms(a, b, reps, [&] { hipLaunchKernelGGL((write16_kernel<true>), dim3(grid), dim3(256), 0, 0, out16, n / 4); });
        report("write16 nt", grid, ms, 1.0 * bytes);
    }
    // LDS-DMA: 4 waves x SLOTS KiB of LDS per workgroup
    for (int grid : {1024, 2048, 4096}) {
        float ms;
#define GLDS(SL, AUX, ST, NAME)                                                                                        \
        ms = time_ms(a, b, reps, [&] { hipLaunchKernelGGL((glds_copy_kernel<SL, AUX, ST>), dim3(grid), dim3(256), 4 * SL * 1024, 0, in16, out16, n / 4, sink); }); \
        report(NAME, grid, ms, (ST ? 2.0 : 1.0) * bytes);
        GLDS(4, 0, true, "glds copy, 4 KiB/wave in flight")
        GLDS(4, 2, true, "glds copy nt, 4 KiB/wave")
        GLDS(8, 0, true, "glds copy, 8 KiB/wave")
        GLDS(8, 2, true, "glds copy nt, 8 KiB/wave")
        GLDS(8, 0, false, "glds read only, 8 KiB/wave")
        GLDS(8, 2, false, "glds read only nt, 8 KiB/wave")
#undef GLDS
    }
    for (uint32_t chunk : {16u}) {
        char nm[64];
        const uint32_t tiles = (uint32_t)(n / 32768);
        float ms = time_ms(a, b, reps, [&] { hipLaunchKernelGGL((tile_scatter_kernel<1024, 32, false>), dim3(tiles), dim3(1024), 0, 0, in, out, (uint32_t)n, tiles, chunk); });
        snprintf(nm, sizeof nm, "tile_scatter 1024x32 C=%u plain", chunk);
        report(nm, (int)tiles, ms, 2.0 * bytes);
        ms = time_ms(a, b, reps, [&] { hipLaunchKernelGGL((tile_scatter_kernel<1024, 32, true>), dim3(tiles), dim3(1024), 0, 0, in, out, (uint32_t)n, tiles, chunk); });
        snprintf(nm, sizeof nm, "tile_scatter 1024x32 C=%u nt loads", chunk);
        report(nm, (int)tiles, ms, 2.0 * bytes);
    }
    // the clock the chip holds while it streams: 40 copies back to back, then one stamped launch
    {
        const int grid = 2048;
        for (int i = 0; i < 40; i++) hipLaunchKernelGGL((copy16_kernel<false, false>), dim3(grid), dim3(256), 0, 0, in16, out16, n / 4, nullptr);
        hipLaunchKernelGGL((copy16_kernel<false, false>), dim3(grid), dim3(256), 0, 0, in16, out16, n / 4, clk);
        CHECK(hipDeviceSynchronize());
        std::vector<unsigned long long> h(2 * grid);
        CHECK(hipMemcpy(h.data(), clk, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        std::vector<double> mhz;
        for (int g = 0; g < grid; g++) if (h[2 * g + 1]) mhz.push_back((double)h[2 * g] / (double)h[2 * g + 1] * 100.0);
        std::sort(mhz.begin(), mhz.end());
        printf("shader clock while copying (s_memtime / s_memrealtime, %zu workgroups): min %.0f  median %.0f  max %.0f MHz\n", mhz.size(),
               mhz.front(), mhz[mhz.size() / 2], mhz.back());
    }
    // working set: what a pass whose two buffers fit the Infinity Cache would see
    for (int l2 = 22; l2 <= log2n; l2++) {
        const size_t m = (size_t)1 << l2;
        const int grid = 2048;
        float ms = time_ms(a, b, 5, [&] {
            for (int i = 0; i < 8; i++) {
                hipLaunchKernelGGL((copy16_kernel<false, false>), dim3(grid), dim3(256), 0, 0, in16, out16, m / 4, nullptr);
                hipLaunchKernelGGL((copy16_kernel<false, false>), dim3(grid), dim3(256), 0, 0, (const v4u*)out16, (v4u*)in16, m / 4, nullptr);
            }
        });
        char nm[64];
        snprintf(nm, sizeof nm, "ping-pong copy of 2^%d keys (%zu MiB x 2), 16 launches", l2, (m * 4) >> 20);
        report(nm, grid, ms / 16, 2.0 * m * 4);
    }
    CHECK(hipDeviceSynchronize());
    return 0;
}
