// lds_atomic.hip -- what an LDS atomic costs on this part, by address pattern.
//
// Stage 1 (four LDS adds per key) and the rank phase (one returning add per key) are both held to the rate at which a
// CU retires LDS atomics.  This probe times, per CU, wave-wide ds_add_u32 / ds_add_rtn_u32 instructions issued back to back
// by 16 resident waves (one 1024-thread workgroup per CU, the occupancy of both kernels) under these patterns:
//   lanebank : lane l always hits bank l % 32 (row chosen at random): no bank conflict, no shared address
//   random   : uniformly random word of a 2048-word table (what uniform keys give)
//   random/2 : the same with every other lane switched off (cost per ACTIVE lane or per instruction?)
//   same     : all 64 lanes on one word
//   G words  : G distinct words per instruction with 64/G lanes on each (i: lanes interleaved, b: blocks of lanes)
//   C copies : random word of a table replicated C times, the copy chosen by lane % C (stage 1's layout at C = 2)
// Output: clocks per wave instruction per CU (100 MHz s_memrealtime scaled by the shader clock the runtime reports).
// Build: hipcc --offload-arch=gfx950 -O3 -o lds_atomic lds_atomic.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int kIters = 256;     // rounds per wave
constexpr int kUnroll = 16;     // atomics per round (addresses precomputed in registers)
constexpr int kWords = 2048;

template <int PATTERN, bool RETURNING>
__global__ void __launch_bounds__(1024) probe(uint32_t* __restrict__ sink, uint32_t seed)
{
    __shared__ uint32_t table[kWords * 4];
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    for (uint32_t i = tid; i < kWords * 4; i += 1024) table[i] = 0;
    __syncthreads();
    uint32_t x = seed ^ (tid * 2654435761u) ^ (blockIdx.x * 40503u);
    uint32_t idx[kUnroll];
#pragma unroll
    for (int u = 0; u < kUnroll; u++) {
        x ^= x << 13; x ^= x >> 17; x ^= x << 5;
        if (PATTERN == 0) idx[u] = ((x >> 8) % (kWords / 32)) * 32 + (lane & 31u);
        else if (PATTERN == 3) idx[u] = (uint32_t)u;
        else if (PATTERN >= 10) {   // G = 2^(PATTERN-10) distinct words per instruction, 64/G lanes on each (low-entropy digits)
            constexpr uint32_t G = 1u << (PATTERN >= 10 ? PATTERN - 10 : 0);
            idx[u] = (uint32_t)u * 64u + (PATTERN & 1 ? (lane % G) : (lane / (64u / G))) * 33u;   // odd: interleaved lanes, even: lane blocks
        }
        else if (PATTERN >= 4) {   // 2^(PATTERN-3) copies of the table, chosen by lane: lanes of different classes never share a bank
            constexpr uint32_t C = 1u << (PATTERN >= 4 ? PATTERN - 3 : 0);
            idx[u] = ((x >> 8) % (kWords * 4 / C)) * C + (lane & (C - 1));
        }
        else idx[u] = (x >> 8) % kWords;
    }
    const bool on = PATTERN != 2 || (lane & 1u) == 0;
    uint32_t acc = 0;
    for (int it = 0; it < kIters; it++) {
        if (on) {
#pragma unroll
            for (int u = 0; u < kUnroll; u++) {
                if (RETURNING) acc += atomicAdd(&table[idx[u]], 1u);
                else __hip_atomic_fetch_add(&table[idx[u]], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
    }
    __syncthreads();
    acc += table[tid];
    if (acc == 0xFFFFFFFFu) sink[0] = acc;
}

// Stage 1's instruction mix: every atomic's address is made on the spot from a changing 32-bit value with V vector ALU
// operations (field extract, bank swizzle, scale) -- does the vector ALU work hide behind the LDS pipe or add to it?
template <int V>
__global__ void __launch_bounds__(1024) probe_mix(uint32_t* __restrict__ sink, uint32_t seed)
{
    __shared__ uint32_t table[kWords * 4];
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    for (uint32_t i = tid; i < kWords * 4; i += 1024) table[i] = 0;
    __syncthreads();
    uint32_t x = seed ^ (tid * 2654435761u) ^ (blockIdx.x * 40503u);
    const uint32_t copy = lane & 1u;
    for (int it = 0; it < kIters; it++) {
        x = x * 1664525u + 1013904223u;                       // a fresh "key" (not counted in V)
#pragma unroll
        for (int u = 0; u < kUnroll; u++) {
            uint32_t f;
            if (V <= 2) f = __builtin_amdgcn_ubfe(x, (uint32_t)(u + 3), 11u);                                  // 1 + scale
            else if (V <= 4) { f = __builtin_amdgcn_ubfe(x, (uint32_t)(u + 3), 11u); f ^= __builtin_amdgcn_ubfe(x, (uint32_t)(u + 8), 5u); }   // 3 + scale
            else { f = __builtin_amdgcn_ubfe(x, (uint32_t)(u + 3), 11u); f ^= __builtin_amdgcn_ubfe(x, (uint32_t)(u + 8), 5u); f = (f + (x >> 27)) & 2047u; f ^= (x >> 30); }
            __hip_atomic_fetch_add(&table[f * 2 + copy], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    __syncthreads();
    if (table[tid] == 0xFFFFFFFFu) sink[0] = 1;
}

template <int V>
static void run_mix(const char* name, int cus, double mhz, uint32_t* sink)
{
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL((probe_mix<V>), dim3(cus), dim3(1024), 0, 0, sink, 1u);
    CHECK(hipDeviceSynchronize());
    float best = 1e9f;
    for (int rep = 0; rep < 5; rep++) {
        CHECK(hipEventRecord(a));
        hipLaunchKernelGGL((probe_mix<V>), dim3(cus), dim3(1024), 0, 0, sink, 7u + rep);
        CHECK(hipEventRecord(b));
        CHECK(hipEventSynchronize(b));
        float ms;
        CHECK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
    }
    printf("%-10s %-9s %8.3f ms  %6.2f clk per wave instruction per CU\n", name, "no-return", best, best * 1e-3 * mhz * 1e6 / (16.0 * kIters * kUnroll));
}

template <int PATTERN, bool RETURNING>
static void run(const char* name, int cus, double mhz, uint32_t* sink)
{
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL((probe<PATTERN, RETURNING>), dim3(cus), dim3(1024), 0, 0, sink, 1u);
    CHECK(hipDeviceSynchronize());
    float best = 1e9f;
    for (int rep = 0; rep < 5; rep++) {
        CHECK(hipEventRecord(a));
        hipLaunchKernelGGL((probe<PATTERN, RETURNING>), dim3(cus), dim3(1024), 0, 0, sink, 7u + rep);
        CHECK(hipEventRecord(b));
        CHECK(hipEventSynchronize(b));
        float ms;
        CHECK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
    }
    const double instr_per_cu = 16.0 * kIters * kUnroll;   // wave instructions one CU retires
    const double clocks = best * 1e-3 * mhz * 1e6;
    printf("%-10s %-9s %8.3f ms  %6.2f clk per wave instruction per CU\n", name, RETURNING ? "returning" : "no-return", best, clocks / instr_per_cu);
}

int main()
{
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const double mhz = prop.clockRate / 1e3;
    printf("%s: %d CUs, %.0f MHz (launch overhead included: %d atomics per lane)\n", prop.gcnArchName, cus, mhz, kIters * kUnroll);
    uint32_t* sink;
    CHECK(hipMalloc(&sink, 4));
    run<0, false>("lanebank", cus, mhz, sink);
    run<0, true>("lanebank", cus, mhz, sink);
    run<1, false>("random", cus, mhz, sink);
    run<1, true>("random", cus, mhz, sink);
    run<2, false>("random/2", cus, mhz, sink);
    run<2, true>("random/2", cus, mhz, sink);
    run<4, false>("2 copies", cus, mhz, sink);
    run<5, false>("4 copies", cus, mhz, sink);
    run<6, false>("8 copies", cus, mhz, sink);
    run<7, false>("16 copies", cus, mhz, sink);
    run<4, true>("2 copies", cus, mhz, sink);
    run<5, true>("4 copies", cus, mhz, sink);
    run_mix<2>("mix 2 valu", cus, mhz, sink);
    run_mix<4>("mix 4 valu", cus, mhz, sink);
    run_mix<8>("mix 8 valu", cus, mhz, sink);
    run<11, false>("2 words i", cus, mhz, sink);
    run<12, false>("4 words b", cus, mhz, sink);
    run<13, false>("8 words i", cus, mhz, sink);
    run<14, false>("16 words b", cus, mhz, sink);
    run<15, false>("32 words i", cus, mhz, sink);
    run<12, true>("4 words b", cus, mhz, sink);
    run<3, false>("same", cus, mhz, sink);
    run<3, true>("same", cus, mhz, sink);
    return 0;
}
