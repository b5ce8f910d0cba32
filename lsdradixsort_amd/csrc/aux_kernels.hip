// aux_kernels.hip -- stages 1 and 2 (histograms and scans) and the launch dispatchers.
//
// Stage 1 stands in for BuildHistogramsKernel (.cu:660-702); stage 2 for the reference's
// offset construction (.cu:862-895: D2D copy, BlockPrefixSumKernel, two TransposeSMEMKernel
// launches, GPUPrefixSum + AddBlockSumsKernel).  All of it is small next to stage 3.
#include "lsd_device.hpp"
#include "lsd_kernels.hpp"

// 8-bit digits, 8 regions: FOUR copies of the 32 KiB of counters, chosen by lane % 4, in 1024-thread workgroups (128 KiB
// of LDS, one workgroup per CU).  LDS atomics of a wave instruction that meet on one word are served a lane per clock
// (tools/ceiling/lds_atomic.hip), so what the copies buy is not speed on uniform keys (1, 2 and 4 copies measure within
// 3 % of each other) but a bound on what a heavy value costs: with c copies at most 16/c lanes of a 16-lane group share a
// word.  2 -> 4 copies: stage 1 on keys that are half zeros 1.02 -> 0.63 ms, on 90 % one value 1.66 -> 0.86 ms, before
// the heavy values are counted by hand (add_field4 below).
#ifndef LSD_R8_HIST_COPIES
#define LSD_R8_HIST_COPIES 4
#endif
#define LSD_R8_HIST_COPIES_VALUE LSD_R8_HIST_COPIES

namespace lsd {

// ------------------------------------------------------------------------------------------
// Stage 1 (onesweep): every digit histogram of the array in ONE read.
//
// A pass permutes keys and never changes them, so the counts LSDRadixSortPass builds at
// .cu:30-35 for each pass can all be taken from the unsorted input.  Each workgroup keeps
// G x 2^R counters in LDS (replicated for narrow digits so 64 lanes do not serialise on two
// or sixteen words), streams keys with 16-byte loads, and flushes once with global atomics.
// ------------------------------------------------------------------------------------------
constexpr int kHistThreads = 256;
#ifndef LSD_HIST_VPT
#define LSD_HIST_VPT 4
#endif
constexpr int kHistVecPerThread = LSD_HIST_VPT;   // uint4 loads in flight per thread per iteration

template <int R, int G>
__global__ void __launch_bounds__(kHistThreads) digit_histograms_kernel(const uint32_t* __restrict__ keys, uint32_t n,
                                                                       uint32_t shift0, uint32_t* __restrict__ hist,
                                                                       uint32_t vec_chunks)
{
    constexpr int H = 1 << R;
    constexpr int C = hist_copies<R>();
    __shared__ uint32_t s_hist[G * H * C];
    const uint32_t tid = threadIdx.x;
    const uint32_t copy = tid & (C - 1);
    for (uint32_t j = tid; j < (uint32_t)(G * H * C); j += kHistThreads) s_hist[j] = 0;
    __syncthreads();

    auto count_key = [&](uint32_t k) {
#pragma unroll
        for (int g = 0; g < G; g++) {
            const uint32_t d = digit_at<R>(k, shift0 + g * R);
            uint32_t* slot = &s_hist[(g * H + d) * C + copy];
            if (R >= 6) {
                // Low-entropy digits (sorted / constant input) would serialise 64 lanes on
                // one LDS word; when the whole wave agrees, one lane adds 64.
                const uint32_t d0 = __builtin_amdgcn_readfirstlane(d);
                if (__builtin_amdgcn_read_exec() == ~0ull && __all(d == d0)) {
                    if ((tid & 63u) == 0) atomicAdd(&s_hist[(g * H + d0) * C], 64u);
                    continue;
                }
            }
            atomicAdd(slot, 1u);
        }
    };

    // body: whole uint4 chunks, grid-strided; each chunk is kHistThreads*4 keys
    const uint4* __restrict__ keys4 = reinterpret_cast<const uint4*>(keys);
    for (uint32_t c = blockIdx.x * kHistVecPerThread; c < vec_chunks; c += gridDim.x * kHistVecPerThread) {
        uint4 v[kHistVecPerThread];
#pragma unroll
        for (int u = 0; u < kHistVecPerThread; u++) {
            const uint32_t cc = c + u;
            v[u] = cc < vec_chunks ? keys4[(size_t)cc * kHistThreads + tid] : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < kHistVecPerThread; u++) {
            if (c + u < vec_chunks) {
                count_key(v[u].x);
                count_key(v[u].y);
                count_key(v[u].z);
                count_key(v[u].w);
            }
        }
    }
    // tail: the last (n mod chunk) keys -- or, for a base that is not 16-byte aligned (vec_chunks == 0: a
    // slice of a larger buffer), every key -- one per thread per step, strided over the whole grid
    {
        const uint32_t tail_begin = vec_chunks * (kHistThreads * 4);
        for (size_t i = (size_t)tail_begin + (size_t)blockIdx.x * kHistThreads + tid; i < n; i += (size_t)gridDim.x * kHistThreads) {
            const uint32_t k = keys[i];
#pragma unroll
            for (int g = 0; g < G; g++) atomicAdd(&s_hist[(g * H + digit_at<R>(k, shift0 + g * R)) * C + copy], 1u);
        }
    }
    __syncthreads();
    for (uint32_t j = tid; j < (uint32_t)(G * H); j += kHistThreads) {
        uint32_t sum = 0;
#pragma unroll
        for (int c = 0; c < C; c++) sum += s_hist[j * C + c];
        if (sum) atomicAdd(&hist[j], sum);
    }
}

template <int R, int G>
static hipError_t launch_digit_histograms_inst(uint32_t shift0, const uint32_t* keys, uint32_t n, uint32_t* hist,
                                               hipStream_t stream)
{
    // 16-byte loads need a 16-byte aligned base; otherwise everything goes through the (grid-wide) scalar loop.
    const bool aligned = (reinterpret_cast<uintptr_t>(keys) & 15u) == 0;
    const uint32_t vec_chunks = aligned ? n / (kHistThreads * 4) : 0;
    uint32_t blocks = aligned ? (vec_chunks + kHistVecPerThread - 1) / kHistVecPerThread : (n + kHistThreads * 16 - 1) / (kHistThreads * 16);
    if (blocks > 2048) blocks = 2048;   // 256 CUs x 8: enough waves to cover HBM latency
    if (blocks == 0) blocks = 1;
    hipLaunchKernelGGL((digit_histograms_kernel<R, G>), dim3(blocks), dim3(kHistThreads), 0, stream, keys, n, shift0,
                       hist, vec_chunks);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// Bucket counts for the splitter partition (multi-GPU step 1 on skewed keys): bucket of a key = the
// number of (ascending) splitters <= key.  Same structure as the digit histogram above, eight
// replicated LDS counters per bucket.
// ------------------------------------------------------------------------------------------
struct SplitterSet {
    uint32_t count;   // buckets - 1
    uint32_t live;    // splitters compared (the rest lie above every key)
    uint32_t value[7];
};

__global__ void __launch_bounds__(kHistThreads) bucket_histogram_kernel(const uint32_t* __restrict__ keys, uint32_t n,
                                                                       SplitterSet sp, uint32_t* __restrict__ hist,
                                                                       uint32_t vec_chunks)
{
    constexpr int C = 8;
    __shared__ uint32_t s_hist[8 * C];
    const uint32_t tid = threadIdx.x;
    if (tid < 8 * C) s_hist[tid] = 0;
    __syncthreads();
    auto count_key = [&](uint32_t k) {
        uint32_t b = 0;
#pragma unroll
        for (int i = 0; i < 7; i++) b += (i < (int)sp.live && k >= sp.value[i]) ? 1u : 0u;
        atomicAdd(&s_hist[b * C + (tid & (C - 1))], 1u);
    };
    const uint4* __restrict__ keys4 = reinterpret_cast<const uint4*>(keys);
    for (uint32_t c = blockIdx.x; c < vec_chunks; c += gridDim.x) {
        const uint4 v = keys4[(size_t)c * kHistThreads + tid];
        count_key(v.x);
        count_key(v.y);
        count_key(v.z);
        count_key(v.w);
    }
    for (size_t i = (size_t)vec_chunks * (kHistThreads * 4) + (size_t)blockIdx.x * kHistThreads + tid; i < n;
         i += (size_t)gridDim.x * kHistThreads)
        count_key(keys[i]);   // tail, or everything when the base is not 16-byte aligned
    __syncthreads();
    if (tid <= sp.count) {
        uint32_t sum = 0;
#pragma unroll
        for (int c = 0; c < C; c++) sum += s_hist[tid * C + c];
        if (sum) atomicAdd(&hist[tid], sum);
    }
}

hipError_t launch_bucket_histogram(int bits, const uint32_t* splitters_host, int live, const uint32_t* keys, uint32_t n,
                                   uint32_t* hist, hipStream_t stream)
{
    if (bits < 1 || bits > 3 || !splitters_host || live < 0 || live > (1 << bits) - 1) return hipErrorInvalidValue;
    SplitterSet sp{};
    sp.count = (1u << bits) - 1u;
    sp.live = (uint32_t)live;
    for (uint32_t i = 0; i < sp.live; i++) sp.value[i] = splitters_host[i];
    const bool aligned = (reinterpret_cast<uintptr_t>(keys) & 15u) == 0;
    const uint32_t vec_chunks = aligned ? n / (kHistThreads * 4) : 0;
    uint32_t blocks = aligned ? vec_chunks : (n + kHistThreads * 16 - 1) / (kHistThreads * 16);
    if (blocks > 2048) blocks = 2048;
    if (blocks == 0) blocks = 1;
    hipLaunchKernelGGL(bucket_histogram_kernel, dim3(blocks), dim3(kHistThreads), 0, stream, keys, n, sp, hist, vec_chunks);
    return hipGetLastError();
}

hipError_t launch_digit_histograms(int radix_bits, int groups, uint32_t shift0, const uint32_t* keys, uint32_t n,
                                   uint32_t* hist, hipStream_t stream)
{
    if (groups == 1) {
        switch (radix_bits) {
            case 1: return launch_digit_histograms_inst<1, 1>(shift0, keys, n, hist, stream);
            case 2: return launch_digit_histograms_inst<2, 1>(shift0, keys, n, hist, stream);
            case 3: return launch_digit_histograms_inst<3, 1>(shift0, keys, n, hist, stream);
            case 4: return launch_digit_histograms_inst<4, 1>(shift0, keys, n, hist, stream);
            case 8: return launch_digit_histograms_inst<8, 1>(shift0, keys, n, hist, stream);
            default: return hipErrorInvalidValue;
        }
    }
    if (groups * radix_bits != 32) return hipErrorInvalidValue;
    switch (radix_bits) {
        case 1: return launch_digit_histograms_inst<1, 32>(shift0, keys, n, hist, stream);
        case 2: return launch_digit_histograms_inst<2, 16>(shift0, keys, n, hist, stream);
        case 4: return launch_digit_histograms_inst<4, 8>(shift0, keys, n, hist, stream);
        case 8: return launch_digit_histograms_inst<8, 4>(shift0, keys, n, hist, stream);
        default: return hipErrorInvalidValue;
    }
}

// ------------------------------------------------------------------------------------------
// Stage 2 (onesweep): exclusive scan of each group's 2^R digit counts -- the inclusive scan
// of .cu:38-41 turned exclusive (PrefixSum, .cu:128-139).  One workgroup per group.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) scan_digit_counts_kernel(const uint32_t* __restrict__ hist,
                                                               uint32_t* __restrict__ base, int bins)
{
    __shared__ uint32_t s_wave[4];
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u, wave = tid >> 6;
    const uint32_t v = tid < (uint32_t)bins ? hist[blockIdx.x * bins + tid] : 0u;
    uint32_t incl = wave_inclusive_scan(v, lane);
    if (lane == 63u) s_wave[wave] = incl;
    __syncthreads();
    for (uint32_t w = 0; w < wave; w++) incl += s_wave[w];
    if (tid < (uint32_t)bins) base[blockIdx.x * bins + tid] = incl - v;
}

hipError_t launch_scan_digit_counts(int radix_bits, int groups, const uint32_t* hist, uint32_t* base,
                                    hipStream_t stream)
{
    if (radix_bits < 1 || radix_bits > 8) return hipErrorInvalidValue;
    hipLaunchKernelGGL(scan_digit_counts_kernel, dim3(groups), dim3(256), 0, stream, hist, base, 1 << radix_bits);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// Stage 1 (onesweep with regions): joint counts for every pass in ONE read.
//
// For pass p the rank-and-scatter kernel wants, per region x of that pass's input, the histogram
// of digit p (lsd_kernels.hpp, "Regions").  Region membership is a key field too -- the top three
// bits of digit p-1 -- so (digit p, region) is one (R+3)-bit field of the key, bits
// [R*p - 3, R*p + R), and counting it is one v_bfe_u32 and one LDS atomic per key per pass, the
// same work as a plain digit histogram with a table 8x as large.  Pass 0 has no previous digit:
// its regions are by position, uniform for a whole 1024-key chunk.
// ------------------------------------------------------------------------------------------
// Copies of every counter, chosen by lane (tid & (C-1)): small tables are replicated so that 64 lanes do not
// pile onto a few hundred words.
constexpr int joint_copies(int radix_bits, bool wide, int counters_per_table, bool dma = false)
{
    if (counters_per_table < 1024) return 4;
    if (radix_bits == 8 && !wide && counters_per_table <= 2048) return dma ? 2 : LSD_R8_HIST_COPIES_VALUE;   // 128 KiB of LDS at most
    return 1;
}

// WIDE (4-bit digits, B = 4): one LDS atomic serves TWO passes.  The field of pass p is key bits
// [4p - 4, 4p + 4); the 12-bit field W_j = bits [8j - 4, 8j + 8) contains the fields of passes 2j (its low
// 8 bits) and 2j + 1 (its high 8 bits), so counting W_0..W_3 (W_0: position region | byte 0) and summing
// 16 counters per output at flush time gives all eight tables from four atomics per key instead of eight:
// the kernel is LDS-atomic-bound, so that is what its time follows (0.49 -> 0.30 ms at 2^28 keys).
// DMA: the keys come in by LDS-DMA (global_load_lds_dwordx4, non-temporal: no VGPR destination, nothing kept in L2) into a
// per-wave ring of kDmaBuffers groups of VPT KiB and are fetched from there with one ds_read_b128 per vector.  On this
// part a read-only stream of LDS-DMA nt loads runs at 6.9-7.0 TB/s where plain 16-byte loads reach 5.2-5.4 and nt ones
// 5.7-5.9 (tools/ceiling/ceiling2.hip, profiles/r3_ceilings.txt), and the vector-memory return path and 2 x VPT x 4 VGPRs are
// free for the counting.  The ring costs LDS: 8-bit digits keep TWO lane-class copies of the counters (64 KiB) beside it.
constexpr int kDmaBuffers = 3;
template <int R, bool WIDE>
constexpr int joint_dma_vpt() { return WIDE ? 4 : 2; }   // KiB per wave per group: 16 waves x 3 x 2 KiB (r = 8), 8 waves x 3 x 4 KiB (r = 4)

template <int R, int THREADS, bool WIDE = false, bool DMA = false>
__global__ void __launch_bounds__(THREADS) joint_histograms_kernel(const uint32_t* __restrict__ keys, uint32_t n,
                                                                  uint32_t region0_keys, uint32_t* __restrict__ joint,
                                                                  uint32_t vec_chunks, const KeyTransform xf,
                                                                  uint32_t first_key, const uint32_t* __restrict__ skip)
{
    if (skip && *skip != 0u) return;   // uniform: the hybrid form took the sort (hybrid.hip)
    // `keys` may be a slice [first_key, first_key + n) of the array being sorted (the host entry counts each chunk as
    // it arrives over PCIe): pass-0 regions are by position in the WHOLE array; first_key is a multiple of the chunk.
    const uint32_t chunk_base = first_key / (uint32_t)(THREADS * 4);
    static_assert(!WIDE || R == 4, "wide fields are laid out for 4-bit digits with 4 region bits");
    constexpr int P = 32 / R;
    constexpr int B = region_bits_for_radix(R);
    constexpr int F = (1 << R) << B;          // fields per pass: (digit, region)
    constexpr int NF = WIDE ? P / 2 : P;      // LDS tables
    constexpr int FW = WIDE ? 4096 : F;       // counters per LDS table
    // Narrow digits put 64 lanes on a few hundred words per pass: replicate the table so that
    // neighbouring lanes use different words (and banks); wide digits spread by themselves.
    constexpr int C = joint_copies(R, WIDE, FW, DMA);
    static_assert(!WIDE || C == 1, "the wide flush reads one copy per counter");
    extern __shared__ __attribute__((aligned(16))) uint32_t s_joint[];   // [NF][FW][C], then (DMA) the waves' rings
    const uint32_t tid = threadIdx.x;
    const uint32_t copy = tid & (C - 1);
    for (uint32_t j = tid; j < (uint32_t)(NF * FW * C); j += THREADS) s_joint[j] = 0;
    __syncthreads();
    // Counter of a field value, bank-swizzled: the low five index bits (the LDS bank) are XORed with the
    // next five.  Few-valued digits (16 values per byte: text, small alphabets) give field values that
    // are multiples of 8 -- four banks for the whole wave without this (0.81 ms instead of 0.27).
#ifdef LSD_HIST_NO_SWIZZLE
    auto word = [&](uint32_t slot) -> uint32_t& { return s_joint[slot * C]; };
#else
    auto word = [&](uint32_t slot) -> uint32_t& { return s_joint[(slot ^ ((slot >> 5) & 31u)) * C]; };
#endif

    // Low-entropy fields (constant or sorted input, dead high digits) would serialise all 64 lanes
    // of a wave on one LDS word; when the whole wave agrees on a field, one lane adds 64 instead.
    // The agreement test is only paid by groups of keys whose FIRST key already shows it in some
    // digit (uniform random input takes the plain path with P tests per 16 keys).
    auto add_field_checked = [&](uint32_t slot) {
        const uint32_t s0 = __builtin_amdgcn_readfirstlane(slot);
        if (__builtin_amdgcn_read_exec() == ~0ull && __all(slot == s0)) {
            if ((tid & 63u) == 0) atomicAdd(&word(s0), 64u);
        } else {
            atomicAdd(&word(slot) + copy, 1u);
        }
    };
    auto count_key_checked = [&](uint32_t k, uint32_t region0) {
        if (WIDE) {
            add_field_checked((region0 << 8) | (k & 0xFFu));
#pragma unroll
            for (int j = 1; j < NF; j++) add_field_checked(j * FW + digit_at<12>(k, (uint32_t)(8 * j - 4)));
            return;
        }
        add_field_checked((region0 << R) | digit_at<R>(k, 0));   // pass 0: region-major in LDS (see flush)
#pragma unroll
        for (int p = 1; p < P; p++) add_field_checked(p * F + digit_at<R + B>(k, (uint32_t)(R * p - B)));
    };
    [[maybe_unused]] uint32_t probe_acc = 0;   // -DLSD_HIST_PROBE_NOATOMIC only
    auto count_key_plain = [&](uint32_t k, uint32_t region0) {
        if (WIDE) {
            atomicAdd(&word((region0 << 8) | (k & 0xFFu)), 1u);
#pragma unroll
            for (int j = 1; j < NF; j++) atomicAdd(&word(j * FW + digit_at<12>(k, (uint32_t)(8 * j - 4))), 1u);
            return;
        }
#ifdef LSD_HIST_PROBE_NOATOMIC   // timing probe only (wrong counts): the address arithmetic without the LDS operations
        probe_acc += (uint32_t)(uintptr_t)(&word((region0 << R) | digit_at<R>(k, 0)) + copy);
#pragma unroll
        for (int p = 1; p < P; p++)
            probe_acc ^= (uint32_t)(uintptr_t)(&word(p * F + digit_at<R + B>(k, (uint32_t)(R * p - B))) + copy);
        return;
#endif
        atomicAdd(&word((region0 << R) | digit_at<R>(k, 0)) + copy, 1u);
#pragma unroll
        for (int p = 1; p < P; p++)
            atomicAdd(&word(p * F + digit_at<R + B>(k, (uint32_t)(R * p - B))) + copy, 1u);
    };
    // Field f of a key as a table slot (the tables follow each other in LDS).
    auto slot_of = [&](int f, uint32_t k, uint32_t region0) -> uint32_t {
        if (WIDE) return f == 0 ? ((region0 << 8) | (k & 0xFFu)) : (uint32_t)(f * FW) + digit_at<12>(k, (uint32_t)(8 * f - 4));
        return f == 0 ? ((region0 << R) | digit_at<R>(k, 0)) : (uint32_t)(f * F) + digit_at<R + B>(k, (uint32_t)(R * f - B));
    };
    // HEAVY field values.  LDS atomics of one wave instruction that meet on one word are served a lane per clock
    // (tools/ceiling/lds_atomic.hip: 63 clocks for a whole 16-lane group on one word against 7 for random words), so a value
    // that a quarter, half or all of the keys carry -- zeros, a default value, constant or sorted input, dead digits -- would
    // cost stage 1 several times its uniform-key time even with the copies.  A group of VPT vectors whose first keys show
    // such a value in some field (lane 0's value, held by at least kHeavyLanes lanes) takes the careful path below: keys
    // that hold a candidate value are counted in scalar registers (a compare and a population count per wave row, no LDS
    // operation), everybody else adds for itself.  A group without one takes the plain path; both paths count every key
    // exactly, the choice is speed only.
    constexpr uint32_t kHeavyLanes = 16;
    constexpr uint32_t kNoCandidate = 0xFFFFFFFFu;   // never a slot
    // Software-pipelined with TWO register buffers that swap roles (the loop is unrolled by two): while one group of
    // 16-byte loads goes through the LDS atomics the next is in flight, and the wait in front of a group is a COUNTED one
    // (`vmcnt(VPT)`: everything but the loads just issued).  For the compiler to count, the loads and the group they overtake
    // must sit in ONE straight line: its wait-count pass merges paths conservatively, so a load behind a branch of its own
    // (an `if (chunk < end)` per load, an `if (more) load_group()` per group -- rounds 1 and 2 had both) turned the wait into
    // `vmcnt(0)`, i.e. into waiting for the loads just issued, with nothing in flight while a wave counted.  So the loop takes
    // FULL groups only and always loads: past its last group a workgroup reloads the group it already holds (an L2 hit)
    // and does not count it.  Chunks beyond the last full group go with the tail below.
    constexpr int VPT = kHistVecPerThread;
    const uint4* __restrict__ keys4 = reinterpret_cast<const uint4*>(keys);
    const uint32_t full_chunks = vec_chunks / VPT * VPT;
    auto load_group = [&](uint32_t c, uint4 (&v)[VPT]) {
#ifndef LSD_HIST_PLAIN_LOADS   // non-temporal loads: a read-only stream of them runs 8 % faster than plain ones (profiles/r3_ceilings.txt:
                             // 5.67-5.88 against 5.24-5.40 TB/s), and the keys are not read again before 2 GiB of other traffic has
                             // gone by; stage 1 0.267-0.274 -> 0.249 ms at 2^28 keys (tools/ab_bench.sh); -DLSD_HIST_PLAIN_LOADS builds the other
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        const u32x4* __restrict__ k4 = reinterpret_cast<const u32x4*>(keys);
#pragma unroll
        for (int u = 0; u < VPT; u++) {
            const u32x4 t = __builtin_nontemporal_load(k4 + (size_t)(c + u) * THREADS + tid);
            v[u] = make_uint4(t.x, t.y, t.z, t.w);
        }
#else
#pragma unroll
        for (int u = 0; u < VPT; u++) v[u] = keys4[(size_t)(c + u) * THREADS + tid];
#endif
    };
    uint32_t key1 = 0, key2 = 0;        // heavy-key candidates of this wave (uniform; kept from group to group)
    bool have1 = false, have2 = false;
    // region_of(u): pass-0 region of vector u of the group (a vector's 4 keys, and the 256 keys of the wave's row, share it)
    auto count_vectors = [&](auto region_of, uint4 (&v)[DMA ? joint_dma_vpt<R, WIDE>() : VPT]) {
        constexpr int NV = DMA ? joint_dma_vpt<R, WIDE>() : VPT;
        if (xf.on) {   // typed sorts count the "sortable" form of the keys (uniform branch); applied where the keys are used
#pragma unroll
            for (int u = 0; u < NV; u++)
                v[u] = make_uint4(to_sortable(v[u].x, xf), to_sortable(v[u].y, xf), to_sortable(v[u].z, xf), to_sortable(v[u].w, xf));
        }
        // region0_keys is a multiple of the chunk (THREADS*4 keys), so a chunk is in one region
        const uint32_t region_first = region_of(0);
        // Heavy KEYS first (zeros, a default value, two-valued keys): a key equal to a candidate is counted for ALL its fields
        // by one compare, ballot and population count -- against NF times that in the per-field form below, which such keys
        // used to take (round 2: stage 1 at 4-bit digits 1.05 ms on half-zero keys against 0.25 ms on uniform ones).  The
        // candidates are sticky across groups (a global default value stays one): the group's first keys are compared with
        // them, and only if they do not describe the group (fewer than 16 lanes) is lane 0's key, then lane 32's, tried.
        {
            const uint32_t k0 = v[0].x;
            uint32_t n1 = (uint32_t)__builtin_popcountll(__ballot(k0 == key1));
            if (!have1 || n1 < kHeavyLanes) {
                have1 = have2 = false;
                const uint32_t a = __builtin_amdgcn_readfirstlane(k0);
                unsigned long long m = __ballot(k0 == a);
                if ((uint32_t)__builtin_popcountll(m) >= kHeavyLanes) {
                    key1 = a;
                    have1 = true;
                } else {
                    const uint32_t b = (uint32_t)__builtin_amdgcn_readlane((int)k0, 32);
                    m = __ballot(k0 == b);
                    if ((uint32_t)__builtin_popcountll(m) >= kHeavyLanes) {
                        key1 = b;
                        have1 = true;
                    }
                }
                if (have1 && ~m != 0ull) {   // a second one: the first value that differs, if eight lanes hold it
                    const uint32_t other = (uint32_t)__builtin_amdgcn_readlane((int)k0, (int)__builtin_ctzll(~m));
                    if ((uint32_t)__builtin_popcountll(__ballot(k0 == other)) >= 8u) {
                        key2 = other;
                        have2 = true;
                    }
                }
            }
        }
        if (have1) {
            const uint32_t lane = tid & 63u;
            uint32_t total1 = 0, total2 = 0;   // uniform: scalar registers
#pragma unroll
            for (int u = 0; u < NV; u++) {
                const uint32_t region0 = region_of(u);
                const uint32_t k4[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
                uint32_t n1 = 0, n2 = 0;
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const bool h1 = k4[q] == key1, h2 = have2 && k4[q] == key2;
                    n1 += (uint32_t)__builtin_popcountll(__ballot(h1));
                    n2 += (uint32_t)__builtin_popcountll(__ballot(h2));
                    if (!(h1 || h2)) count_key_plain(k4[q], region0);
                }
                // field 0 carries the position region of the vector; the other fields are the key's alone
                if (lane == 0) {
                    if (n1) atomicAdd(&word(slot_of(0, key1, region0)), n1);
                    if (n2) atomicAdd(&word(slot_of(0, key2, region0)), n2);
                }
                total1 += n1;
                total2 += n2;
            }
            if (lane == 0) {
#pragma unroll
                for (int f = 1; f < NF; f++) {
                    if (total1) atomicAdd(&word(slot_of(f, key1, 0u)), total1);
                    if (total2) atomicAdd(&word(slot_of(f, key2, 0u)), total2);
                }
            }
            return;
        }
        bool any = false;
#pragma unroll
        for (int f = 0; f < NF; f++) {
            const uint32_t s0 = slot_of(f, v[0].x, region_first);
            any = any || (uint32_t)__builtin_popcountll(__ballot(s0 == __builtin_amdgcn_readfirstlane(s0))) >= kHeavyLanes;
        }
        if (!any) {
#pragma unroll
            for (int u = 0; u < NV; u++) {
                const uint32_t region0 = region_of(u);
                count_key_plain(v[u].x, region0);
                count_key_plain(v[u].y, region0);
                count_key_plain(v[u].z, region0);
                count_key_plain(v[u].w, region0);
            }
            return;
        }
        // Per field: up to two candidate values c1, c2 whose holders are counted in scalar registers and added by one lane
        // when the candidates change or the group ends.  The candidates are kept while they describe the vector at hand (16
        // lanes or more of its first keys hold one of them: global heavy values never change) and are picked again from
        // the vector's own first keys otherwise (sorted input: every vector has its own leading value and, where a digit
        // boundary falls inside the wave's 256 keys, a trailing one = the first value that differs).
        const uint32_t lane = tid & 63u;
#pragma unroll
        for (int f = 0; f < NF; f++) {
            uint32_t c1 = kNoCandidate, c2 = kNoCandidate, held1 = 0, held2 = 0;   // uniform: scalar registers; picked at the first vector
            auto flush = [&]() {
                if (lane == 0) {
                    if (held1) atomicAdd(&word(c1), held1);
                    if (held2) atomicAdd(&word(c2), held2);
                }
                held1 = held2 = 0;
            };
#pragma unroll
            for (int u = 0; u < NV; u++) {
                const uint32_t region0 = region_of(u);
                const uint32_t s4[4] = {slot_of(f, v[u].x, region0), slot_of(f, v[u].y, region0), slot_of(f, v[u].z, region0),
                                        slot_of(f, v[u].w, region0)};
                const uint32_t a = s4[0];
                if ((uint32_t)__builtin_popcountll(__ballot(a == c1 || a == c2)) < kHeavyLanes) {
                    flush();
                    const uint32_t first = __builtin_amdgcn_readfirstlane(a);
                    const unsigned long long mf = __ballot(a == first);
                    c1 = c2 = kNoCandidate;
                    if ((uint32_t)__builtin_popcountll(mf) >= kHeavyLanes) {
                        c1 = first;
                        const unsigned long long rest = ~mf;
                        if (rest) {
                            const uint32_t other = (uint32_t)__builtin_amdgcn_readlane((int)a, (int)__builtin_ctzll(rest));
                            if ((uint32_t)__builtin_popcountll(__ballot(a == other)) >= 8u) c2 = other;
                        }
                    }
                }
                if (c1 == kNoCandidate) {
                    atomicAdd(&word(s4[0]) + copy, 1u);
                    atomicAdd(&word(s4[1]) + copy, 1u);
                    atomicAdd(&word(s4[2]) + copy, 1u);
                    atomicAdd(&word(s4[3]) + copy, 1u);
                    continue;
                }
                const bool all4 = (s4[0] == c1) & (s4[1] == c1) & (s4[2] == c1) & (s4[3] == c1);
                if (__all(all4)) {   // the wave's 256 keys agree (constant or sorted input, dead digits)
                    held1 += 256u;
                    continue;
                }
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const bool h1 = s4[q] == c1, h2 = s4[q] == c2;
                    held1 += (uint32_t)__builtin_popcountll(__ballot(h1));
                    held2 += (uint32_t)__builtin_popcountll(__ballot(h2));
                    if (!(h1 || h2)) atomicAdd(&word(s4[q]) + copy, 1u);
                }
            }
            flush();
        }
    };
    uint32_t tail_begin;   // first key the loops below leave to the tail
    if constexpr (!DMA) {
        auto count_group = [&](uint32_t c, uint4 (&v)[VPT]) {
            count_vectors([&](int u) { return ((chunk_base + c + (uint32_t)u) * (uint32_t)(THREADS * 4)) / region0_keys; }, v);
        };
        const uint32_t stride = gridDim.x * VPT;
        uint32_t c = blockIdx.x * VPT;
        if (c < full_chunks) {
            uint4 buf_a[VPT], buf_b[VPT];
            load_group(c, buf_a);
            for (;;) {
                const uint32_t c1 = c + stride;
                const bool more1 = c1 < full_chunks;
                load_group(more1 ? c1 : c, buf_b);
                count_group(c, buf_a);
                if (!more1) break;
                const uint32_t c2 = c1 + stride;
                const bool more2 = c2 < full_chunks;
                load_group(more2 ? c2 : c1, buf_a);
                count_group(c1, buf_b);
                if (!more2) break;
                c = c2;
            }
        }
        tail_begin = full_chunks * (THREADS * 4);
    } else {
        // A wave streams GROUPS of DV consecutive pieces (a piece = 64 vectors = 256 keys = 1 KiB: what one LDS-DMA wave
        // instruction moves; lane l's 16 bytes land at base + 16 l), strided over all waves of the grid.  kDmaBuffers groups
        // are in flight per wave; the wait in front of a group is counted (the DMA completes in issue order), and a wave past
        // its last group re-requests its first one so that the count stays the same (an L2 hit, not counted twice).
        constexpr int DV = joint_dma_vpt<R, WIDE>();
        constexpr int WAVES = THREADS / kWave;
        typedef __attribute__((address_space(3))) void lds_void_t;
        typedef __attribute__((address_space(1))) const void global_cvoid_t;
        const uint32_t lane = tid & 63u, wave = tid >> 6;
        uint4* const ring = reinterpret_cast<uint4*>(s_joint + NF * FW * C) + (size_t)wave * (kDmaBuffers * DV * 64);
        const uint32_t groups = vec_chunks == 0 ? 0u : (n / 256u) / (uint32_t)DV;      // vec_chunks == 0: base not 16-byte aligned
        const uint32_t stride = gridDim.x * (uint32_t)WAVES;
        const uint32_t first = blockIdx.x * (uint32_t)WAVES + wave;
        auto issue = [&](uint32_t g, int buf) {
#pragma unroll
            for (int u = 0; u < DV; u++)
                __builtin_amdgcn_global_load_lds((global_cvoid_t*)(keys4 + ((size_t)g * DV + u) * 64 + lane),
                                                 (lds_void_t*)(ring + (buf * DV + u) * 64), 16, 0, 2 /* nt */);
        };
        if (first < groups) {
#pragma unroll
            for (int b = 0; b < kDmaBuffers; b++) {
                const uint32_t g = first + (uint32_t)b * stride;
                issue(g < groups ? g : first, b);
            }
            int buf = 0;
            for (uint32_t g = first; g < groups; g += stride) {
                asm volatile("s_waitcnt vmcnt(%0)" : : "n"((kDmaBuffers - 1) * DV) : "memory");
                uint4 v[DV];
#pragma unroll
                for (int u = 0; u < DV; u++) v[u] = ring[(buf * DV + u) * 64 + lane];
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the buffer is in registers: it may be refilled
                const uint32_t ahead = g + (uint32_t)kDmaBuffers * stride;
                issue(ahead < groups ? ahead : first, buf);
                count_vectors([&](int u) { return (uint32_t)(((size_t)first_key + ((size_t)g * DV + (size_t)u) * 256u) / region0_keys); }, v);
                buf = buf + 1 == kDmaBuffers ? 0 : buf + 1;
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // nothing may still be landing in LDS when the workgroup leaves
        }
        tail_begin = groups * (uint32_t)(DV * 256);
    }
    {
        // tail: the chunks past the last full group and the keys past the last chunk -- or every key when the base is not
        // 16-byte aligned (vec_chunks == 0) -- strided over the grid; a step's keys are consecutive, so a wave stays inside
        // one pass-0 region except at a boundary
        for (size_t i = (size_t)tail_begin + (size_t)blockIdx.x * THREADS + tid; i < n; i += (size_t)gridDim.x * THREADS)
            count_key_checked(xf.on ? to_sortable(keys[i], xf) : keys[i], (uint32_t)((first_key + i) / region0_keys));
    }
#ifdef LSD_HIST_PROBE_NOATOMIC
    if (probe_acc == 0x12345u) s_joint[tid] = probe_acc;
#endif
    __syncthreads();
    // Flush.  Pass 0's fields sit region-major in LDS: all 64 lanes of a wave share their position
    // region, so with the region in the low index bits they would share four LDS banks; the global
    // table is digit-major for every pass.
    for (uint32_t j = tid; j < (uint32_t)(P * F); j += THREADS) {
        uint32_t cnt = 0;
        if (WIDE) {
            // global entry j = pass p, digit d, region x (digit-major); sum the 16 wide counters that agree
            const uint32_t p = j / (uint32_t)F, d = (j >> B) & 15u, x = j & 15u;
#pragma unroll
            for (uint32_t o = 0; o < 16; o++) {
                uint32_t slot;
                if (p == 0) slot = (x << 8) | (o << 4) | d;            // W_0 = region0 | digit 1 | digit 0: sum over digit 1
                else if (p == 1) slot = (o << 8) | (d << 4) | x;       // region = digit 0: sum over the position region
                else if ((p & 1) == 0) slot = (o << 8) | (d << 4) | x; // W_j = digit 2j+1 | digit 2j | digit 2j-1: sum over the top
                else slot = (d << 8) | (x << 4) | o;                   // pass 2j+1: region = digit 2j: sum over the bottom
                cnt += word((p / 2) * FW + slot);
            }
        } else {
            uint32_t src = j;
            if (j < (uint32_t)F) src = ((j & (uint32_t)((1 << B) - 1)) << R) | (j >> B);
#pragma unroll
            for (int q = 0; q < C; q++) cnt += (&word(src))[q];
        }
        if (cnt) atomicAdd(&joint[j], cnt);
    }
}

#ifndef LSD_R8_HIST_THREADS3
#define LSD_R8_HIST_THREADS3 1024
#endif
#ifndef LSD_R4_HIST_THREADS
#define LSD_R4_HIST_THREADS 512
#endif
#ifndef LSD_R8_HIST_THREADS
#define LSD_R8_HIST_THREADS 512
#endif

#ifndef LSD_HIST_DMA
#define LSD_HIST_DMA 0   // 1 builds the LDS-DMA form (measured round 3: 0.32 ms against 0.27 ms -- the kernel is bound by the LDS pipe, and the DMA's LDS writes and the ds_read_b128 fetches are 10 % more work for it)
#endif

template <int R, int THREADS, bool WIDE = false, bool DMA = (LSD_HIST_DMA != 0)>
static hipError_t launch_joint_inst(const uint32_t* keys, uint32_t n, uint32_t region0_keys, uint32_t* joint,
                                    hipStream_t stream, const KeyTransform& xf, uint32_t first_key, const uint32_t* skip)
{
    constexpr int P = 32 / R;
    constexpr int F = (1 << R) << region_bits_for_radix(R);
    constexpr int NF = WIDE ? P / 2 : P;
    constexpr int FW = WIDE ? 4096 : F;
    constexpr int C = joint_copies(R, WIDE, FW, DMA);
    constexpr size_t ring_bytes = DMA ? (size_t)(THREADS / kWave) * kDmaBuffers * joint_dma_vpt<R, WIDE>() * 1024 : 0;
    constexpr size_t lds_bytes = (size_t)NF * FW * C * sizeof(uint32_t) + ring_bytes;
    static_assert(lds_bytes <= 160 * 1024, "counters and rings must fit one CU's LDS");
    auto kernel = joint_histograms_kernel<R, THREADS, WIDE, DMA>;
    if (lds_bytes > 64 * 1024) {
        static std::atomic<uint64_t> told{0};
        const hipError_t attr = allow_dynamic_lds(reinterpret_cast<const void*>(kernel), lds_bytes, told);
        if (attr != hipSuccess) return attr;
    }
    if (region0_keys == 0 || region0_keys % (THREADS * 4) != 0 || first_key % (THREADS * 4) != 0) return hipErrorInvalidValue;
    const bool aligned = (reinterpret_cast<uintptr_t>(keys) & 15u) == 0;
    const uint32_t vec_chunks = aligned ? n / (THREADS * 4) : 0;
    uint32_t blocks = aligned ? (vec_chunks + kHistVecPerThread - 1) / kHistVecPerThread : (n + THREADS * 16 - 1) / (THREADS * 16);
#ifndef LSD_HIST_GRID_WAVES
#define LSD_HIST_GRID_WAVES (2048 * 4)
#endif
    const uint32_t cap = (uint32_t)(LSD_HIST_GRID_WAVES * 64 / THREADS);   // enough waves to cover HBM latency (512 workgroups of 1024 threads)
    if (blocks > cap) blocks = cap;
    if (blocks == 0) blocks = 1;
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(THREADS), lds_bytes, stream, keys, n, region0_keys, joint, vec_chunks, xf, first_key, skip);
    return hipGetLastError();
}

hipError_t launch_joint_histograms(int radix_bits, const uint32_t* keys, uint32_t n, uint32_t region0_keys,
                                   uint32_t* joint, hipStream_t stream, const KeyTransform& xf, uint32_t first_key, const uint32_t* skip)
{
    switch (radix_bits) {
#ifdef LSD_R4_NARROW_HIST
        case 4: return launch_joint_inst<4, 256>(keys, n, region0_keys, joint, stream, xf, first_key, skip);
#else
        case 4: return launch_joint_inst<4, LSD_R4_HIST_THREADS, true>(keys, n, region0_keys, joint, stream, xf, first_key, skip);   // 64 KiB of counters per workgroup
#endif
        case 8: return launch_joint_inst<8, (LSD_R8_REGION_BITS == 3 ? LSD_R8_HIST_THREADS3 : LSD_R8_HIST_THREADS)>(keys, n, region0_keys, joint, stream, xf, first_key, skip);   // 32 / 64 / 128 KiB of counters per workgroup at 3 / 4 / 5 bits
        default: return hipErrorInvalidValue;
    }
}

// ------------------------------------------------------------------------------------------
// Stage 2 (onesweep): every pass's region table from the counts.  One workgroup per pass:
//   digit_base[d]      = exclusive scan over d of the digit totals          (.cu:38-41 / PrefixSum)
//   base[x][d]         = digit_base[d] + counts of digit d in regions before x
//   extents of pass p+1 = [digit_base[x*H/8], digit_base[(x+1)*H/8])   (regions = top bits of digit p)
//   extents of pass 0   = [x*R0, (x+1)*R0) clipped to n
// ------------------------------------------------------------------------------------------
template <int REG>
__global__ void __launch_bounds__(256) scan_regions_kernel(const uint32_t* __restrict__ counts, int bins, uint32_t n,
                                                          uint32_t tile_keys, uint32_t region0_keys, int passes,
                                                          uint32_t* __restrict__ tables, uint32_t table_words,
                                                          uint32_t* __restrict__ plan, uint32_t* __restrict__ fault,
                                                          const uint32_t* __restrict__ hybrid_ok, uint32_t skip_dead_passes)
{
    __shared__ uint32_t s_wave[4];
    __shared__ uint32_t s_base[257];
    __shared__ uint32_t s_const[kPlanWords];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const int pass = blockIdx.x;
    if (hybrid_ok && *hybrid_ok != 0u) {   // uniform: the hybrid form runs; these passes leave at once and touch nothing (2)
        if (plan && tid == 0) {
            plan[2 * pass] = 2u;
            plan[2 * pass + 1] = 0u;
            if (pass + 1 == passes) plan[2 * passes] = 0u;   // the local stage leaves the keys in the caller's buffer
        }
        return;
    }
    // This kernel is latency, not work: a small sort spends 5 of its 50 us here (rocprofv3, 2^20 keys, round 3).  So every
    // global load it needs is requested up front, in one window: this pass's counts first ...
    const uint32_t* c = counts + (size_t)pass * bins * REG;
    uint32_t* table = tables + (size_t)pass * table_words;
    uint32_t per_region[REG];
    uint32_t total = 0;
    if (tid < (uint32_t)bins) {
#pragma unroll
        for (int x = 0; x < REG; x++) per_region[x] = c[tid * REG + x];
    }
    if (plan) {
        // ... then the pass plan (PassParams::plan): a digit that is the same for every key (one bin holds all n) makes its
        // pass the identity.  This workgroup looks at its own pass and at the ones before it, whose number of REAL passes says
        // which buffer its keys are in: (pass + 1) * bins (pass, digit) cells, dealt over the threads.
        if (tid < (uint32_t)kPlanWords) s_const[tid] = 0;
        __syncthreads();
        const uint32_t cells = (uint32_t)(pass + 1) * (uint32_t)bins;
#pragma unroll 4
        for (uint32_t cell = tid; cell < cells; cell += 256u) {
            uint32_t t = 0;
#pragma unroll
            for (int x = 0; x < REG; x++) t += counts[(size_t)cell * REG + x];
            if (t == n && skip_dead_passes) s_const[cell / (uint32_t)bins] = 1;
        }
        __syncthreads();
        if (tid == 0) {
            uint32_t moved = 0;
            for (int q = 0; q < pass; q++) moved += s_const[q] ? 0u : 1u;
            plan[2 * pass] = s_const[pass];
            plan[2 * pass + 1] = moved & 1u;
            if (pass + 1 == passes) plan[2 * passes] = (moved + (s_const[pass] ? 0u : 1u)) & 1u;
        }
    }
    if (tid < (uint32_t)bins) {
#pragma unroll
        for (int x = 0; x < REG; x++) total += per_region[x];
    }
    uint32_t incl = wave_inclusive_scan(total, lane);
    if (lane == 63u) s_wave[wave] = incl;
    __syncthreads();
    for (uint32_t w = 0; w < wave; w++) incl += s_wave[w];
    const uint32_t digit_base = incl - total;
    // Every key has exactly one digit: a pass's counts sum to n.  Counts that do not (a miscounting stage-1 variant:
    // DESIGN.md section 4.5.2) would give the pass bases and extents that do not describe its input; say so in the fault
    // word here, once, before any pass runs on them (the passes' destination guard keeps their stores in bounds).
    if (fault && tid == (uint32_t)bins - 1u && incl != n) atomicOr(fault, 4u);
    if (tid < (uint32_t)bins) {
        s_base[tid] = digit_base < n ? digit_base : n;   // extents below stay inside [0, n] whatever the counts say
        uint32_t run = digit_base;
#pragma unroll
        for (int x = 0; x < REG; x++) {
            table[kRegionHeaderWords + x * bins + tid] = run;
            run += per_region[x];
        }
    }
    if (tid == 0) s_base[bins] = n;
    __syncthreads();
    // Extents: lane x of the first wave owns region x; the regions' first status rows are an exclusive scan of their tile
    // counts across those lanes.
    if (wave == 0) {
        auto write_extents = [&](uint32_t* t, uint32_t lo, uint32_t hi) {
            lo = lo < n ? lo : n;
            hi = hi < n ? hi : n;
            const uint32_t len = hi > lo ? hi - lo : 0u;
            const uint32_t tiles = lane < (uint32_t)REG ? (len + tile_keys - 1) / tile_keys : 0u;
            const uint32_t upto = wave_inclusive_scan(tiles, lane);
            if (lane < (uint32_t)REG) {
                t[lane] = lo;
                t[kMaxRegions + lane] = len;
                t[2 * kMaxRegions + lane] = tiles;
                t[3 * kMaxRegions + lane] = upto - tiles;
            } else if (lane < (uint32_t)kMaxRegions) {
                // a kernel compiled for more regions than this table has (the 4-bit kernels partitioning by one region for
                // the multi-GPU step) must find the others empty
                t[2 * kMaxRegions + lane] = 0;
            }
        };
        const uint32_t x = lane < (uint32_t)REG ? lane : 0u;
        if (pass == 0) {
            const unsigned long long e0 = (unsigned long long)x * region0_keys, e1 = e0 + region0_keys;
            if (REG == 1) write_extents(table, 0u, n);
            else write_extents(table, e0 < n ? (uint32_t)e0 : n, e1 < n ? (uint32_t)e1 : n);
        }
        if (pass + 1 < passes) {
            uint32_t* next = tables + (size_t)(pass + 1) * table_words;
            const int per = bins / REG;   // digits per region
            if (REG == 1) write_extents(next, 0u, n);
            else write_extents(next, s_base[x * per], s_base[(x + 1) * per]);
        }
    }
}

// The keys (and payloads) back into the caller's buffer when the plan left them in the other one.
__global__ void __launch_bounds__(1024) finish_plan_kernel(const uint32_t* __restrict__ plan_final, uint32_t* __restrict__ keys,
                                                           const uint32_t* __restrict__ alt_keys, uint32_t* __restrict__ vals,
                                                           const uint32_t* __restrict__ alt_vals, uint32_t n)
{
    if (*plan_final == 0) return;   // uniform: the usual case
    for (size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x; i < n; i += (size_t)gridDim.x * 1024) {
        keys[i] = alt_keys[i];
        if (vals) vals[i] = alt_vals[i];
    }
}

hipError_t launch_finish_plan(const uint32_t* plan_final, uint32_t* keys, const uint32_t* alt_keys, uint32_t* vals,
                              const uint32_t* alt_vals, uint32_t n, hipStream_t stream)
{
    if (!plan_final || !keys || !alt_keys || (vals && !alt_vals)) return hipErrorInvalidValue;
    uint32_t blocks = (n + 4 * 1024 - 1) / (4 * 1024);
    if (blocks > 512) blocks = 512;   // two workgroups per CU copy at full rate; the usual launch returns at once
    if (blocks == 0) blocks = 1;
    hipLaunchKernelGGL(finish_plan_kernel, dim3(blocks), dim3(1024), 0, stream, plan_final, keys, alt_keys, vals, alt_vals, n);
    return hipGetLastError();
}

hipError_t launch_scan_regions(int radix_bits, int passes, int regions, const uint32_t* counts, uint32_t n,
                               uint32_t tile_keys, uint32_t region0_keys, uint32_t* tables, hipStream_t stream, uint32_t* plan,
                               uint32_t* fault, const uint32_t* hybrid_ok, bool skip_dead_passes)
{
    if (radix_bits < 1 || radix_bits > 8 || (regions != 1 && regions != regions_for_radix(radix_bits))) return hipErrorInvalidValue;
    const uint32_t sdp = skip_dead_passes ? 1u : 0u;
    if (plan && 2 * passes + 1 > kPlanWords) return hipErrorInvalidValue;
    const int bins = 1 << radix_bits;
    const uint32_t words = (uint32_t)region_table_words(radix_bits);
    if (regions == 1)
        hipLaunchKernelGGL((scan_regions_kernel<1>), dim3(passes), dim3(256), 0, stream, counts, bins, n, tile_keys,
                           region0_keys, passes, tables, words, plan, fault, hybrid_ok, sdp);
    else if (regions == 8)
        hipLaunchKernelGGL((scan_regions_kernel<8>), dim3(passes), dim3(256), 0, stream, counts, bins, n, tile_keys,
                           region0_keys, passes, tables, words, plan, fault, hybrid_ok, sdp);
    else if (regions == 16)
        hipLaunchKernelGGL((scan_regions_kernel<16>), dim3(passes), dim3(256), 0, stream, counts, bins, n, tile_keys,
                           region0_keys, passes, tables, words, plan, fault, hybrid_ok, sdp);
    else
        hipLaunchKernelGGL((scan_regions_kernel<32>), dim3(passes), dim3(256), 0, stream, counts, bins, n, tile_keys,
                           region0_keys, passes, tables, words, plan, fault, hybrid_ok, sdp);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// Stage 1 (staged): per-tile digit counts h[tile][digit], BuildHistogramsKernel .cu:660-702.
// One workgroup per tile; counters in LDS, one coalesced row written per tile.
// ------------------------------------------------------------------------------------------
template <int R, int T>
__global__ void __launch_bounds__(T) tile_histograms_kernel(const uint32_t* __restrict__ keys, uint32_t n,
                                                           uint32_t shift, uint32_t tile_keys,
                                                           uint32_t* __restrict__ hist)
{
    constexpr int H = 1 << R;
    constexpr int C = hist_copies<R>();
    __shared__ uint32_t s_hist[H * C];
    const uint32_t tid = threadIdx.x;
    const uint32_t copy = tid & (C - 1);
    for (uint32_t j = tid; j < (uint32_t)(H * C); j += T) s_hist[j] = 0;
    __syncthreads();
    const uint32_t begin = blockIdx.x * tile_keys;
    const uint32_t end = (n - begin < tile_keys) ? n : begin + tile_keys;
    // 16-byte loads over the tile's body when the tile starts on a 16-byte boundary (tile sizes are multiples of four keys,
    // so it does whenever the array does); the few keys behind the last whole vector, or everything otherwise, one by one
    uint32_t scalar_from = begin;
    if ((reinterpret_cast<uintptr_t>(keys + begin) & 15u) == 0) {
        const uint4* __restrict__ v4 = reinterpret_cast<const uint4*>(keys + begin);
        const uint32_t vecs = (end - begin) / 4u;
        for (uint32_t v = tid; v < vecs; v += T) {
            const uint4 k = v4[v];
            atomicAdd(&s_hist[digit_at<R>(k.x, shift) * C + copy], 1u);
            atomicAdd(&s_hist[digit_at<R>(k.y, shift) * C + copy], 1u);
            atomicAdd(&s_hist[digit_at<R>(k.z, shift) * C + copy], 1u);
            atomicAdd(&s_hist[digit_at<R>(k.w, shift) * C + copy], 1u);
        }
        scalar_from = begin + vecs * 4u;
    }
    for (uint32_t i = scalar_from + tid; i < end; i += T) atomicAdd(&s_hist[digit_at<R>(keys[i], shift) * C + copy], 1u);
    __syncthreads();
    for (uint32_t d = tid; d < (uint32_t)H; d += T) {
        uint32_t sum = 0;
#pragma unroll
        for (int c = 0; c < C; c++) sum += s_hist[d * C + c];
        hist[(size_t)blockIdx.x * H + d] = sum;
    }
}

hipError_t launch_tile_histograms(int radix_bits, const TileShape& shape, const uint32_t* keys, uint32_t n,
                                  uint32_t shift, uint32_t* hist, hipStream_t stream)
{
    const uint32_t tile_keys = (uint32_t)shape.tile();
    const uint32_t tiles = (n + tile_keys - 1) / tile_keys;
    if (tiles == 0) return hipSuccess;
    switch (radix_bits) {
#define LSD_CASE(RB)                                                                                              \
    case RB:                                                                                                      \
        hipLaunchKernelGGL((tile_histograms_kernel<RB, 256>), dim3(tiles), dim3(256), 0, stream, keys, n, shift, \
                           tile_keys, hist);                                                                      \
        break;
        LSD_CASE(1) LSD_CASE(2) LSD_CASE(3) LSD_CASE(4) LSD_CASE(8)
#undef LSD_CASE
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// Stage 2 (staged): offset tables from h[tile][digit].
//   local[t][d]  = exclusive scan over d within tile t                       (.cu:869)
//   global[t][d] = keys with digit < d anywhere + keys with digit d in tiles < t   (.cu:877-895)
// The reference reaches the second by transposing to digit-major and scanning flat; here the
// table stays block-major and the digit-major order is walked directly:
//   (1) column sums over strips of kStrip tiles        -> strip_sum[strip][d]
//   (2) one workgroup scans strip_sum in digit-major order (d outer, strip inner), exclusive
//   (3) each (strip, d) thread replays its strip from that base and writes global[t][d].
// Threads are laid out digit-fastest so every access to a [.][d] row is coalesced.
// ------------------------------------------------------------------------------------------
constexpr uint32_t kStrip = 64;

__global__ void __launch_bounds__(256) local_offsets_kernel(const uint32_t* __restrict__ hist,
                                                           uint32_t* __restrict__ local, uint32_t tiles, int bins_log2)
{
    // 256 / bins rows per workgroup; Hillis-Steele inside each row through LDS
    __shared__ uint32_t s[2][256];
    const uint32_t bins = 1u << bins_log2;
    const uint32_t rows_per_block = 256u >> bins_log2;
    const uint32_t tid = threadIdx.x;
    const uint32_t d = tid & (bins - 1);
    const uint32_t row = blockIdx.x * rows_per_block + (tid >> bins_log2);
    const bool live = row < tiles;
    const uint32_t v = live ? hist[(size_t)row * bins + d] : 0u;
    int cur = 0;
    s[0][tid] = v;
    __syncthreads();
    for (uint32_t off = 1; off < bins; off <<= 1) {
        uint32_t x = s[cur][tid];
        if (d >= off) x += s[cur][tid - off];
        s[cur ^ 1][tid] = x;
        cur ^= 1;
        __syncthreads();
    }
    if (live) local[(size_t)row * bins + d] = s[cur][tid] - v;
}

__global__ void __launch_bounds__(256) strip_sums_kernel(const uint32_t* __restrict__ hist,
                                                        uint32_t* __restrict__ strip_sum, uint32_t tiles,
                                                        uint32_t strips, int bins_log2)
{
    const uint32_t bins = 1u << bins_log2;
    const uint32_t gid = blockIdx.x * 256u + threadIdx.x;   // (strip, digit), digit fastest
    const uint32_t strip = gid >> bins_log2, d = gid & (bins - 1);
    if (strip >= strips) return;
    const uint32_t t0 = strip * kStrip;
    const uint32_t t1 = (tiles - t0 < kStrip) ? tiles : t0 + kStrip;
    uint32_t sum = 0;
    for (uint32_t t = t0; t < t1; t++) sum += hist[(size_t)t * bins + d];
    strip_sum[(size_t)strip * bins + d] = sum;
}

// One workgroup; walks bins*strips entries in digit-major order with a running carry.
__global__ void __launch_bounds__(1024) scan_strip_sums_kernel(uint32_t* __restrict__ strip_sum, uint32_t strips,
                                                              int bins_log2)
{
    __shared__ uint32_t s_wave[16];
    __shared__ uint32_t s_carry;
    const uint32_t bins = 1u << bins_log2;
    const uint32_t total = strips << bins_log2;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    if (tid == 0) s_carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < total; base += 1024u) {
        const uint32_t e = base + tid;                 // digit-major linear index
        const uint32_t d = e / strips, strip = e - d * strips;
        const bool live = e < total;
        const uint32_t v = live ? strip_sum[(size_t)strip * bins + d] : 0u;
        uint32_t incl = wave_inclusive_scan(v, lane);
        if (lane == 63u) s_wave[wave] = incl;
        __syncthreads();
        uint32_t carry = s_carry;
        for (uint32_t w = 0; w < wave; w++) carry += s_wave[w];
        incl += carry;
        if (live) strip_sum[(size_t)strip * bins + d] = incl - v;
        __syncthreads();
        if (tid == 1023u) s_carry = incl;
        __syncthreads();
    }
}

__global__ void __launch_bounds__(256) global_offsets_kernel(const uint32_t* __restrict__ hist,
                                                            const uint32_t* __restrict__ strip_base,
                                                            uint32_t* __restrict__ global, uint32_t tiles,
                                                            uint32_t strips, int bins_log2)
{
    const uint32_t bins = 1u << bins_log2;
    const uint32_t gid = blockIdx.x * 256u + threadIdx.x;
    const uint32_t strip = gid >> bins_log2, d = gid & (bins - 1);
    if (strip >= strips) return;
    const uint32_t t0 = strip * kStrip;
    const uint32_t t1 = (tiles - t0 < kStrip) ? tiles : t0 + kStrip;
    uint32_t running = strip_base[(size_t)strip * bins + d];
    for (uint32_t t = t0; t < t1; t++) {
        const uint32_t c = hist[(size_t)t * bins + d];
        global[(size_t)t * bins + d] = running;
        running += c;
    }
}

size_t tile_offsets_scratch_words(size_t tiles, int radix_bits)
{
    const size_t strips = (tiles + kStrip - 1) / kStrip;
    return strips << radix_bits;
}

hipError_t launch_tile_offsets(int radix_bits, const uint32_t* hist, uint32_t* local, uint32_t* global,
                               uint32_t tiles, uint32_t* scratch, hipStream_t stream)
{
    if (radix_bits < 1 || radix_bits > 8) return hipErrorInvalidValue;
    if (tiles == 0) return hipSuccess;
    const uint32_t bins = 1u << radix_bits;
    if (global) {
        const uint32_t strips = (tiles + kStrip - 1) / kStrip;
        const uint32_t threads = strips * bins;
        const uint32_t blocks = (threads + 255u) / 256u;
        hipLaunchKernelGGL(strip_sums_kernel, dim3(blocks), dim3(256), 0, stream, hist, scratch, tiles, strips,
                           radix_bits);
        hipLaunchKernelGGL(scan_strip_sums_kernel, dim3(1), dim3(1024), 0, stream, scratch, strips, radix_bits);
        hipLaunchKernelGGL(global_offsets_kernel, dim3(blocks), dim3(256), 0, stream, hist, scratch, global, tiles,
                           strips, radix_bits);
    }
    if (local) {
        // after `global`: local may alias hist (in-place, like the reference's h[0,GH))
        const uint32_t rows_per_block = 256u >> radix_bits;
        const uint32_t blocks = (tiles + rows_per_block - 1) / rows_per_block;
        hipLaunchKernelGGL(local_offsets_kernel, dim3(blocks), dim3(256), 0, stream, hist, local, tiles, radix_bits);
    }
    return hipGetLastError();
}

__global__ void widen_counts_kernel(const uint32_t* __restrict__ in, uint64_t* __restrict__ out, int bins)
{
    const int i = threadIdx.x;
    if (i < bins) out[i] = in[i];
}

hipError_t launch_widen_counts(const uint32_t* hist32, uint64_t* counts64, int bins, hipStream_t stream)
{
    hipLaunchKernelGGL(widen_counts_kernel, dim3(1), dim3(256), 0, stream, hist32, counts64, bins);
    return hipGetLastError();
}

__global__ void keep_fault_kernel(uint32_t* sticky, const uint32_t* fault)
{
    if (*fault) *sticky |= *fault;
}

hipError_t launch_keep_fault(uint32_t* sticky, const uint32_t* fault, hipStream_t stream)
{
    hipLaunchKernelGGL(keep_fault_kernel, dim3(1), dim3(1), 0, stream, sticky, fault);
    return hipGetLastError();
}

__global__ void store_u64_kernel(uint64_t* out, uint64_t value) { *out = value; }

hipError_t launch_store_u64(uint64_t* out, uint64_t value, hipStream_t stream)
{
    hipLaunchKernelGGL(store_u64_kernel, dim3(1), dim3(1), 0, stream, out, value);
    return hipGetLastError();
}

// A regular sample of a shard for the splitter choice (sharded.hip): out[0] = m = min(samples, n),
// out[1 + i] = keys[i * n / m] for i < m; the slots behind stay as they are.
__global__ void __launch_bounds__(256) sample_keys_kernel(const uint32_t* __restrict__ keys, uint32_t n, uint32_t samples,
                                                          uint32_t* __restrict__ out)
{
    const uint32_t m = n < samples ? n : samples;
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i == 0) out[0] = m;
    if (i < m) out[1 + i] = keys[(size_t)(((uint64_t)i * n) / m)];
}

hipError_t launch_sample_keys(const uint32_t* keys, uint32_t n, uint32_t samples, uint32_t* out, hipStream_t stream)
{
    if (samples == 0 || !out || (n && !keys)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(sample_keys_kernel, dim3((samples + 255) / 256), dim3(256), 0, stream, keys, n, samples, out);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// Probe for kRankLdsAdd: does a returning LDS add, issued by the 64 lanes of one wave
// instruction onto colliding addresses, return its old values in lane order?  Each wave
// compares ds_add_rtn_u32 against the ballot-derived stable rank over collision patterns from
// "none" to "all 64 lanes on one word", with every CU busy.  Any disagreement clears *ok.
// ------------------------------------------------------------------------------------------
// Run in the occupancy shapes of the kernels that rely on the property: 1024-thread workgroups holding 128 KiB of LDS
// (one per CU, sixteen waves contending for the LDS pipe: the default 32768-key tile) and 512-thread workgroups
// holding 74 KiB (two per CU); the tables sit at the front of the dynamic allocation, the rest only claims the space.
__global__ void __launch_bounds__(1024) probe_lds_add_kernel(uint32_t iters, uint32_t* mismatches)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t s_probe_raw[];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t waves = blockDim.x >> 6;
    volatile lds_u32* cnt = (volatile lds_u32*)s_probe_raw + wave * 256;
    volatile lds_u32* ref = (volatile lds_u32*)s_probe_raw + (waves + wave) * 256;
    for (uint32_t j = lane; j < 256; j += 64) {
        cnt[j] = 0;
        ref[j] = 0;
    }
    uint32_t bad = 0;
    for (uint32_t it = 0; it < iters; it++) {
        uint32_t h = (it * 0x9E3779B9u) ^ (blockIdx.x * 0x85EBCA6Bu) ^ (tid * 0xC2B2AE35u);
        h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16;
        uint32_t d;
        switch ((it + blockIdx.x) % 6u) {
            case 0: d = h & 0xFFu; break;
            case 1: d = h & 0x0Fu; break;
            case 2: d = h & 0x01u; break;
            case 3: d = 7u; break;
            case 4: d = (lane >> 2) & 0xFFu; break;
            default: d = (h & 0xFFu) * ((h >> 8) & 1u); break;
        }
        const uint64_t peers = match_ballot<8>(d);
        const uint32_t before = ref[d];
        const uint32_t expect = mbcnt_add(peers, before);
        const uint32_t old = __hip_atomic_fetch_add((lds_u32*)&cnt[d], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        if (old != expect) bad++;
        if (expect == before) ref[d] = popc64_add(peers, before);
    }
    if (bad) atomicAdd(mismatches, bad);
}

hipError_t probe_lds_add_lane_order(bool* ok, hipStream_t stream)
{
    *ok = false;
    uint32_t* d_bad = nullptr;
    hipError_t e = hipMalloc(&d_bad, sizeof(uint32_t));
    if (e != hipSuccess) return e;
    uint32_t h_bad = 1;
    e = hipMemsetAsync(d_bad, 0, sizeof(uint32_t), stream);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(probe_lds_add_kernel),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(probe_lds_add_kernel, dim3(256 * 3), dim3(1024), 128 * 1024, stream, 600u, d_bad);
        e = hipGetLastError();
    }
    if (e == hipSuccess) {
        hipLaunchKernelGGL(probe_lds_add_kernel, dim3(256 * 3), dim3(512), 74 * 1024, stream, 600u, d_bad);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(&h_bad, d_bad, sizeof(uint32_t), hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    (void)hipFree(d_bad);
    if (e == hipSuccess) *ok = (h_bad == 0);
    return e;
}

// ------------------------------------------------------------------------------------------
// rank-and-scatter dispatch: per-radix translation units hold the instantiations.
// ------------------------------------------------------------------------------------------
hipError_t launch_rank_scatter_r8(int shape_id, int rank_method, bool chained, const PassParams& p, hipStream_t stream);
hipError_t launch_rank_scatter_r4(int shape_id, int rank_method, bool chained, const PassParams& p, hipStream_t stream);
hipError_t launch_rank_scatter_small(int radix_bits, int shape_id, int rank_method, bool chained, const PassParams& p, hipStream_t stream);

// Slot 0 is the default; the others stay compiled for tools/tune.py (DESIGN.md has the sweep).
static const TileShape kShapesR8[] = {{512, 32}, {1024, 16}, {1024, 32}, {512, 16}, {1024, 32}, {256, 16}};
static const TileShape kShapesR4[] = {{512, 32}, {512, 16}, {256, 16}, {1024, 32}, {1024, 32}, {1024, 16}};
static const TileShape kShapesSmall[] = {{256, 16}, {512, 32}, {1024, 32}};

bool single_round_shape(int radix_bits, int id)
{
    // the CAP arguments of rank_scatter_r8.hip / _r4.hip / _small.hip: r8 shapes 1 (1024 x 16, CAP 8192) and 2 (1024 x 32,
    // CAP 16384) and r4 shape 3 (1024 x 32, CAP 16384) reorder in two rounds
    if (radix_bits == 8) return id != 1 && id != 2;
    if (radix_bits == 4) return id != 3;
    return true;
}

int tile_shapes(int radix_bits, const TileShape** out)
{
    switch (radix_bits) {
        case 8: *out = kShapesR8; return (int)(sizeof(kShapesR8) / sizeof(TileShape));
        case 4: *out = kShapesR4; return (int)(sizeof(kShapesR4) / sizeof(TileShape));
        case 1: case 2: case 3: *out = kShapesSmall; return (int)(sizeof(kShapesSmall) / sizeof(TileShape));
        default: *out = nullptr; return 0;
    }
}

hipError_t launch_rank_scatter(int radix_bits, const TileShape& shape, int rank_method, bool chained,
                               const PassParams& p, hipStream_t stream)
{
    const TileShape* shapes = nullptr;
    const int count = tile_shapes(radix_bits, &shapes);
    const int id = (int)(&shape - shapes);   // shapes are identified by their table slot
    if (id < 0 || id >= count) return hipErrorInvalidValue;
    if (p.num_tiles == 0) return hipSuccess;
    switch (radix_bits) {
        case 8: return launch_rank_scatter_r8(id, rank_method, chained, p, stream);
        case 4: return launch_rank_scatter_r4(id, rank_method, chained, p, stream);
        default: return launch_rank_scatter_small(radix_bits, id, rank_method, chained, p, stream);
    }
}

}  // namespace lsd
