// hybrid.hip -- stage 1 and the planner of the HYBRID form (2^25 .. 9.6e8 items; 8- or 4-bit digits; keys, pairs, records; typed keys).
// Written below for its first shape -- 8-bit digits, 2^15 buckets; what differs at 4-bit digits, with 2^14 buckets, for typed keys
// and for shards that share a key prefix is said at the kernels.
//
// The reference moves every key through global memory once per digit (GPULSDRadixSort, LSDRadixSort.cu:844-905: four passes at
// 8-bit digits), and so does this library's chained form: 4 + 4 x 8 = 36 B/key.  An MI355X CU has 160 KiB of LDS: a bucket of
// 16384 keys fits it, and the two LOW digits of such a bucket can be sorted without leaving the CU.  The hybrid form therefore
// runs the two HIGH digits first, as ordinary chained passes (bits 16-23, then 24-31: LSD order, so the array ends up sorted by
// its top 16 bits), and finishes the 2^15 buckets of equal top-15-bit value in LDS (local_sort.hip: bits 0-8, then 9-16):
//
//     one read (counts) + 2 global passes + 1 local stage = 4 + 8 + 8 + 8 = 28 B/key, and the local stage has no chained scan.
//
// It is only valid if EVERY bucket fits the local stage, which depends on the keys -- so it is decided on the device, exactly,
// before anything is moved: the upfront read counts the top-15-bit buckets (32768 counters, 128 KiB of LDS per workgroup)
// next to the first pass's (position region, digit) field, and the planner (one workgroup) takes the hybrid form iff the
// largest bucket holds at most kLocalSortCap keys.  Otherwise every hybrid kernel returns at once and the ordinary form runs
// (its own upfront read included: skewed keys pay 0.2 ms for the attempt).  Either way the result is the sorted array; which
// form ran is visible in the workspace (lsdsort_timing.hybrid).
//
// Same stable rank, same kernels for the global passes; nothing here is derived from the reference's kernels.
#include "lsd_device.hpp"
#include "lsd_kernels.hpp"

namespace lsd {

constexpr int kHybridHistThreads = 1024;
constexpr int kHybridCopiesA = 2;          // lane-class copies of the first pass's 2048 (region, digit) counters
constexpr int kHybridVpt = 4;              // 16-byte vectors per thread per group

// Upfront read of the hybrid form.  fieldA[(digit of bits 16-23) * 8 + position region] and bucket[key >> bucket_shift] (global, zero on
// entry) receive the counts.  Grid-stride over chunks of 4096 keys, two register buffers, non-temporal loads (as stage 1 of the
// ordinary form, aux_kernels.hip).  Heavy values are handled as there: keys equal to a sticky candidate value (zeros, a default
// value) are counted by ballot for both fields at once, and per field the holders of the first lane's counter are counted by
// ballot when sixteen lanes or more share it (sorted or constant input, dead digits, small ranges).
// The sample: 65536 keys at a regular stride, 1024 per workgroup over 64 workgroups, each workgroup counting ITS samples by bucket
// in LDS (one workgroup doing all of them waits 0.12 ms for its own 65536 cache lines; global counters melt on constant keys:
// 65536 returning atomics on one word took 0.6 ms).  A bucket's share of a workgroup's 1024 samples is 1/32 .. 1/16 of a key: eight in one
// bucket -- 0.8 % of all keys, two hundred times a bucket's share -- raise the flag.
__global__ void __launch_bounds__(1024) hybrid_sample_kernel(const uint32_t* __restrict__ keys, uint32_t n, uint32_t bucket_shift,
                                                             uint32_t* __restrict__ words)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t s_cnt[];   // [2^15 at most: with 2^16 buckets two neighbours share a counter]
    __shared__ uint32_t s_differ;
    constexpr uint32_t kSamples = 65536;
    const unsigned long long step = n / kSamples;   // n >= 2^26: at least 1024
    const uint32_t tid = threadIdx.x;
    // interleaved: workgroup w takes samples w, w + 64, ... so each sees the WHOLE array at 64 x step -- a contiguous run of one
    // bucket's keys (sorted input, a bucket of 11000 keys stored together) is spread over the workgroups instead of filling one
    const uint32_t k = keys[(size_t)((unsigned long long)(tid * 64u + blockIdx.x) * step)];
    const uint32_t fold = bucket_shift < 17u ? 17u - bucket_shift : 0u;
    const uint32_t counters = 1u << (32u - bucket_shift - fold);
    for (uint32_t j = tid; j < counters; j += 1024) s_cnt[j] = 0;
    if (tid == 0) s_differ = 0;
    __syncthreads();
    // the key prefix: bits in which no sampled key differs from the first key of the array.  All 64 workgroups OR into the plan
    // word (the upfront read takes its leading zeros, and checks them against every key); for ITS look at the buckets a workgroup
    // uses what its own 1024 samples say -- they span the whole array, and this look is a heuristic.
    uint32_t d = k ^ keys[0];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) d |= __shfl_xor(d, off, kWave);
    if ((tid & 63u) == 0u) atomicOr(&s_differ, d);
    __syncthreads();
    const uint32_t differ = s_differ;
    if (tid == 0) atomicOr(&words[kHybridWordDiffer], differ);
    uint32_t prefix = hybrid_prefix_of(differ);
    if (prefix > kHybridMaxPrefix) prefix = 0;   // constant keys, a dead top byte: the upfront read will not run anyway
    const uint32_t mask = (1u << (32u - bucket_shift)) - 1u;
    if (atomicAdd(&s_cnt[((k >> (bucket_shift - prefix)) & mask) >> fold], 1u) + 1u >= 8u) words[kHybridWordHopeless] = 1u;
}

hipError_t launch_hybrid_sample(const uint32_t* keys, uint32_t n, int bucket_bits, uint32_t* words, hipStream_t stream)
{
    if (n < 65536u * 64u || bucket_bits < 11 || (1 << bucket_bits) > kHybridBuckets || !words) return hipErrorInvalidValue;
    const uint32_t bucket_shift = 32u - (uint32_t)bucket_bits;
    constexpr size_t lds_bytes = (size_t)32768 * sizeof(uint32_t);
    static std::atomic<uint64_t> told{0};
        const hipError_t attr = allow_dynamic_lds(reinterpret_cast<const void*>(hybrid_sample_kernel), lds_bytes, told);
    if (attr != hipSuccess) return attr;
    hipLaunchKernelGGL(hybrid_sample_kernel, dim3(64), dim3(1024), lds_bytes, stream, keys, n, bucket_shift, words);
    return hipGetLastError();
}

// R: digit width of the global passes, 8 or 4.  XF: typed keys, counted as to_sortable(key, xf).  The key prefix (0 .. 7 bits, from
// the sample's plan word): buckets and digits are taken that many bits lower, and every key is checked against the first.
// B16: 2^16 buckets, counted in 16-bit halves of the same 32768 LDS words.  A half that overflows (65536 keys of one bucket in one
// workgroup's share: nothing the local stage could take anyway) wraps or carries into its neighbour; either way the counts then
// sum to LESS than n (every such event loses 65535 or 65536), which the planner's sum check refuses.
template <int R, bool XF, bool B16>
__global__ void __launch_bounds__(kHybridHistThreads) hybrid_histograms_kernel(const uint32_t* __restrict__ keys, uint32_t n,
                                                                               uint32_t region0_keys, uint32_t* __restrict__ field_a,
                                                                               uint32_t* __restrict__ bucket, uint32_t vec_chunks,
                                                                               uint32_t* __restrict__ words, uint32_t bucket_shift,
                                                                               const KeyTransform xf)
{
    // uniform: the sample has ruled the hybrid form out, or found a constant top byte (the ordinary form then skips a pass and
    // moves no more bytes than this one would): nothing is counted, and the planner, seeing no counts, says no
    if (words[kHybridWordHopeless] != 0u) return;
    const uint32_t prefix = hybrid_prefix_of(words[kHybridWordDiffer]);
    if (prefix > kHybridMaxPrefix) return;
    // 8-bit digits: the first pass's field, [8 position regions][256 digits], two lane-class copies.  4-bit digits: the JOINT field
    // of the first two passes, [16 position regions][bits 16-23] in one copy -- the same 4096 words and the same two LDS adds per
    // key; the planner sums it to pass A's [digit][region] and pass B's [digit][A's digit].
    constexpr int T = kHybridHistThreads, CA = R == 8 ? kHybridCopiesA : 1, VPT = kHybridVpt;
    constexpr uint32_t FA = R == 8 ? 2048 : 4096;
    const uint32_t NB = 1u << (32u - bucket_shift);   // at most kHybridBuckets (the LDS is sized for that)
    extern __shared__ __attribute__((aligned(16))) uint32_t s_mem[];
    uint32_t* const s_a = s_mem;                 // [region][digit][CA], region-major: a wave's lanes share the region
    uint32_t* const s_b = s_mem + FA * CA;       // [NB] words, or (B16) [NB / 2] words of two 16-bit counters
    const uint32_t tid = threadIdx.x, lane = tid & 63u, copy = tid & (CA - 1);
    for (uint32_t j = tid; j < FA * CA + (B16 ? NB / 2u : NB); j += T) s_mem[j] = 0;
    __syncthreads();

    const uint32_t a_shift = 16u - prefix, b_shift = bucket_shift - prefix, b_mask = NB - 1u;
    auto slot_a = [&](uint32_t k, uint32_t region0) -> uint32_t { return ((region0 << 8) | ((k >> a_shift) & 0xFFu)) * CA; };
    auto slot_b = [&](uint32_t k) -> uint32_t { return (k >> b_shift) & b_mask; };
    const uint32_t kref = n ? (XF ? to_sortable(keys[0], xf) : keys[0]) : 0u;
    uint32_t differs = 0;   // OR of (key ^ first key) over this thread's keys -- the prefix check
    auto add_b = [&](uint32_t b, uint32_t count) {   // bucket b += count
        if (B16) atomicAdd(&s_b[b >> 1], count << ((b & 1u) << 4));
        else atomicAdd(&s_b[b], count);
    };
    auto count_plain = [&](uint32_t k, uint32_t region0) {
        atomicAdd(&s_a[slot_a(k, region0) + copy], 1u);
        add_b(slot_b(k), 1u);
    };
    uint32_t key1 = 0, key2 = 0;   // sticky heavy-key candidates (uniform)
    bool have1 = false, have2 = false;
    auto count_group = [&](uint32_t c, uint4 (&v)[VPT]) {
        // heavy keys: see joint_histograms_kernel (aux_kernels.hip)
        {
            const uint32_t k0 = v[0].x;
            const uint32_t n1 = (uint32_t)__builtin_popcountll(__ballot(k0 == key1));
            if (!have1 || n1 < 16u) {
                have1 = have2 = false;
                const uint32_t a = __builtin_amdgcn_readfirstlane(k0);
                unsigned long long m = __ballot(k0 == a);
                if ((uint32_t)__builtin_popcountll(m) >= 16u) {
                    key1 = a;
                    have1 = true;
                } else {
                    const uint32_t b = (uint32_t)__builtin_amdgcn_readlane((int)k0, 32);
                    m = __ballot(k0 == b);
                    if ((uint32_t)__builtin_popcountll(m) >= 16u) {
                        key1 = b;
                        have1 = true;
                    }
                }
                if (have1 && ~m != 0ull) {
                    const uint32_t other = (uint32_t)__builtin_amdgcn_readlane((int)k0, (int)__builtin_ctzll(~m));
                    if ((uint32_t)__builtin_popcountll(__ballot(k0 == other)) >= 8u) {
                        key2 = other;
                        have2 = true;
                    }
                }
            }
        }
#pragma unroll
        for (int u = 0; u < VPT; u++) {
            const uint32_t region0 = ((c + (uint32_t)u) * (uint32_t)(T * 4)) / region0_keys;   // a chunk lies in one region
            const uint32_t k4[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
            differs |= (k4[0] ^ kref) | (k4[1] ^ kref) | (k4[2] ^ kref) | (k4[3] ^ kref);
            if (have1) {
                uint32_t n1 = 0, n2 = 0;
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const bool h1 = k4[q] == key1, h2 = have2 && k4[q] == key2;
                    n1 += (uint32_t)__builtin_popcountll(__ballot(h1));
                    n2 += (uint32_t)__builtin_popcountll(__ballot(h2));
                    if (!(h1 || h2)) count_plain(k4[q], region0);
                }
                if (lane == 0) {
                    if (n1) {
                        atomicAdd(&s_a[slot_a(key1, region0)], n1);
                        add_b(slot_b(key1), n1);
                    }
                    if (n2) {
                        atomicAdd(&s_a[slot_a(key2, region0)], n2);
                        add_b(slot_b(key2), n2);
                    }
                }
                continue;
            }
            // Heavy FIELD values (sorted or constant input, dead digits, small key ranges: many lanes on one counter, which
            // the LDS serves a lane per clock): per field, if sixteen lanes or more of the vector's first keys share the first
            // lane's counter, its holders among all four keys are counted by ballot and added once; everybody else adds for
            // itself.  Uniform keys pay the two looks (a few scalar instructions per vector).
            {
                const uint32_t a0 = slot_a(k4[0], region0), a_first = (uint32_t)__builtin_amdgcn_readfirstlane(a0);
                if ((uint32_t)__builtin_popcountll(__ballot(a0 == a_first)) >= 16u) {
                    uint32_t held = 0;
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const uint32_t sa = slot_a(k4[q], region0);
                        const bool h = sa == a_first;
                        held += (uint32_t)__builtin_popcountll(__ballot(h));
                        if (!h) atomicAdd(&s_a[sa + copy], 1u);
                    }
                    if (lane == 0) atomicAdd(&s_a[a_first], held);
                } else {
#pragma unroll
                    for (int q = 0; q < 4; q++) atomicAdd(&s_a[slot_a(k4[q], region0) + copy], 1u);
                }
                const uint32_t b0 = slot_b(k4[0]), b_first = (uint32_t)__builtin_amdgcn_readfirstlane(b0);
                if ((uint32_t)__builtin_popcountll(__ballot(b0 == b_first)) >= 16u) {
                    uint32_t held = 0;
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const uint32_t sb = slot_b(k4[q]);
                        const bool h = sb == b_first;
                        held += (uint32_t)__builtin_popcountll(__ballot(h));
                        if (!h) add_b(sb, 1u);
                    }
                    if (lane == 0) add_b(b_first, held);
                } else {
#pragma unroll
                    for (int q = 0; q < 4; q++) add_b(slot_b(k4[q]), 1u);
                }
            }
        }
    };
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const u32x4* __restrict__ k4p = reinterpret_cast<const u32x4*>(keys);
    auto load_group = [&](uint32_t c, uint4 (&v)[VPT]) {
#pragma unroll
        for (int u = 0; u < VPT; u++) {
            const u32x4 t = __builtin_nontemporal_load(k4p + (size_t)(c + u) * T + tid);
            v[u] = XF ? make_uint4(to_sortable(t.x, xf), to_sortable(t.y, xf), to_sortable(t.z, xf), to_sortable(t.w, xf))
                      : make_uint4(t.x, t.y, t.z, t.w);
        }
    };
    const uint32_t full_chunks = vec_chunks / VPT * VPT;
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
    // tail (and everything, for a base that is not 16-byte aligned: vec_chunks == 0): one key per thread per step
    for (size_t i = (size_t)full_chunks * (T * 4) + (size_t)blockIdx.x * T + tid; i < n; i += (size_t)gridDim.x * T)
    {
        const uint32_t k = XF ? to_sortable(keys[i], xf) : keys[i];
        differs |= k ^ kref;
        count_plain(k, (uint32_t)(i / region0_keys));
    }
    if (prefix && (differs >> (32u - prefix)) != 0u) words[kHybridWordViolated] = 1u;   // benign race: everybody writes 1
    __syncthreads();
    for (uint32_t j = tid; j < FA; j += T) {
        uint32_t cnt = 0;
        if (R == 8) {   // global layout [digit][region], LDS layout [region][digit][copy]
            const uint32_t d = j >> 3, x = j & 7u;
#pragma unroll
            for (int q = 0; q < CA; q++) cnt += s_a[((x << 8) | d) * CA + q];
        } else {        // the joint field as it lies
            cnt = s_a[j];
        }
        if (cnt) atomicAdd(&field_a[j], cnt);
    }
    for (uint32_t j = tid; j < NB; j += T) {
        const uint32_t cnt = B16 ? (s_b[j >> 1] >> ((j & 1u) << 4)) & 0xFFFFu : s_b[j];
        if (cnt) atomicAdd(&bucket[j], cnt);
    }
}

hipError_t launch_hybrid_histograms(int radix_bits, const uint32_t* keys, uint32_t n, uint32_t region0_keys, uint32_t* field_a, uint32_t* bucket,
                                    int bucket_bits, uint32_t* words, hipStream_t stream, const KeyTransform& xf)
{
    if (!words) return hipErrorInvalidValue;
    if (bucket_bits < 11 || (1 << bucket_bits) > kHybridBuckets || (radix_bits != 8 && radix_bits != 4)) return hipErrorInvalidValue;
    constexpr int T = kHybridHistThreads;
    constexpr size_t lds_bytes = (size_t)(4096 + 32768) * sizeof(uint32_t);   // 2^15 bucket counters of 32 bits or 2^16 of 16
    static_assert(2048 * kHybridCopiesA == 4096, "both digit widths keep 4096 field counters");
    static_assert(lds_bytes <= 160 * 1024, "one workgroup per CU");
    if (region0_keys == 0 || region0_keys % (T * 4) != 0) return hipErrorInvalidValue;
    static std::atomic<uint64_t> told{0};
    hipError_t attr = hipSuccess;
    {
        int dev = 0;
        attr = hipGetDevice(&dev);
        if (attr == hipSuccess && !(told.load(std::memory_order_acquire) & (1ull << (dev & 63)))) {
#define LSD_K(R, XF, B16) reinterpret_cast<const void*>(hybrid_histograms_kernel<R, XF, B16>)
            const void* kernels[8] = {LSD_K(8, false, false), LSD_K(8, true, false), LSD_K(4, false, false), LSD_K(4, true, false),
                                      LSD_K(8, false, true),  LSD_K(8, true, true),  LSD_K(4, false, true),  LSD_K(4, true, true)};
#undef LSD_K
            for (const void* k : kernels) {
                attr = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
                if (attr != hipSuccess) break;
            }
            if (attr == hipSuccess) told.fetch_or(1ull << (dev & 63), std::memory_order_release);
        }
    }
    if (attr != hipSuccess) return attr;
    const bool aligned = (reinterpret_cast<uintptr_t>(keys) & 15u) == 0;
    const uint32_t vec_chunks = aligned ? n / (T * 4) : 0;
    // one resident workgroup per CU, each flushing 34816 counters once: more workgroups would only flush more
    uint32_t blocks = aligned ? (vec_chunks + kHybridVpt - 1) / kHybridVpt : (n + T * 16 - 1) / (T * 16);
    if (blocks > 256) blocks = 256;
    if (blocks == 0) blocks = 1;
#define LSD_HYB_HIST(R, XF, B16) hipLaunchKernelGGL((hybrid_histograms_kernel<R, XF, B16>), dim3(blocks), dim3(T), lds_bytes, stream, keys, n, region0_keys, field_a, bucket, vec_chunks, words, 32u - (uint32_t)bucket_bits, xf)
    const bool general = xf.on != 0, b16 = bucket_bits == 16;
    if (radix_bits == 8) {
        if (b16) { if (general) LSD_HYB_HIST(8, true, true); else LSD_HYB_HIST(8, false, true); }
        else     { if (general) LSD_HYB_HIST(8, true, false); else LSD_HYB_HIST(8, false, false); }
    } else {
        if (b16) { if (general) LSD_HYB_HIST(4, true, true); else LSD_HYB_HIST(4, false, true); }
        else     { if (general) LSD_HYB_HIST(4, true, false); else LSD_HYB_HIST(4, false, false); }
    }
#undef LSD_HYB_HIST
    return hipGetLastError();
}

// The planner: one workgroup.  From the bucket counts: the verdict (largest bucket <= kLocalSortCap and the counts sum to n),
// the buckets' bases (exclusive scan, 2^bucket_bits + 1 words), the plan words the other kernels read, and the global passes'
// (digit, region) count fields that the upfront read has not written itself:
//   8-bit digits: the second pass's field B -- region = top three bits of the first pass's digit, i.e. PER / 2 consecutive buckets
//                 per cell;
//   4-bit digits: all four ([pass][digit][region], region = the previous pass's digit): D (bits 28-31 by 24-27) and C (bits 24-27
//                 by 20-23) are sums of buckets, B (bits 20-23 by 16-19) and A (bits 16-19 by position region) sums of the joint
//                 field [position region][bits 16-23] of the upfront read.
template <int R, int PER>   // PER = consecutive buckets per thread: 64 / 32 / 16 for 2^16 / 2^15 / 2^14 buckets (a thread = the top ten bits)
__global__ void __launch_bounds__(1024) hybrid_plan_kernel(const uint32_t* __restrict__ bucket, uint32_t n, uint32_t* __restrict__ bases,
                                                           uint32_t* __restrict__ fields_out, const uint32_t* __restrict__ joint,
                                                           uint32_t* __restrict__ words, uint32_t* __restrict__ large_list, uint32_t small_cap)
{
    __shared__ uint32_t s_wave[16], s_max[16], s_large, s_c[256], s_d[256];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    if (tid == 0) s_large = 0;
    if (R == 4 && tid < 256) s_c[tid] = s_d[tid] = 0;
    __syncthreads();
    uint32_t cnt[PER];
    const uint4* src = reinterpret_cast<const uint4*>(bucket + (size_t)tid * PER);
#pragma unroll
    for (int j = 0; j < (int)PER / 4; j++) {
        const uint4 t = src[j];
        cnt[4 * j] = t.x; cnt[4 * j + 1] = t.y; cnt[4 * j + 2] = t.z; cnt[4 * j + 3] = t.w;
    }
    uint32_t sum = 0, mx = 0, half0 = 0;
    uint32_t quarter[4] = {0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < (int)PER; j++) {
        if (j == (int)PER / 2) half0 = sum;
        sum += cnt[j];
        quarter[j / (PER / 4)] += cnt[j];
        mx = cnt[j] > mx ? cnt[j] : mx;
        // the launch over all buckets takes those of up to small_cap keys (the three-per-CU variant's capacity); the others
        // go on a list that a second, small launch walks with the large variant
        if (cnt[j] > small_cap) large_list[atomicAdd(&s_large, 1u)] = tid * PER + (uint32_t)j;
    }
    if (R == 8) {
        fields_out[2 * tid] = half0;             // cell (digit, region) = buckets [PER t, PER t + PER / 2): the layout of a pass's count table
        fields_out[2 * tid + 1] = sum - half0;
    } else {
        // tid = bits 22-31 of the key, a thread's quarter q = bits 20-21: cell D = [bits 28-31][bits 24-27] = tid >> 2,
        // cell C = [bits 24-27][bits 20-23] = (tid & 63) << 2 | q
        atomicAdd(&s_d[tid >> 2], sum);
#pragma unroll
        for (int q = 0; q < 4; q++) atomicAdd(&s_c[((tid & 63u) << 2) | (uint32_t)q], quarter[q]);
        if (tid < 256) {          // A[digit d4][region x] = sum over d5 of joint[x][d5 << 4 | d4]
            const uint32_t d4 = tid >> 4, x = tid & 15u;
            uint32_t a = 0;
#pragma unroll
            for (uint32_t d5 = 0; d5 < 16; d5++) a += joint[x * 256u + ((d5 << 4) | d4)];
            fields_out[tid] = a;
        } else if (tid < 512) {   // B[digit d5][region d4] = sum over x of joint[x][d5 << 4 | d4]: index = bits 16-23 as they are
            const uint32_t byte = tid - 256u;
            uint32_t b = 0;
#pragma unroll
            for (uint32_t x = 0; x < 16; x++) b += joint[x * 256u + byte];
            fields_out[256u + byte] = b;
        }
    }
    uint32_t incl = wave_inclusive_scan(sum, lane);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const uint32_t other = __shfl_xor(mx, off, kWave);
        mx = other > mx ? other : mx;
    }
    if (lane == 63u) s_wave[wave] = incl;
    if (lane == 0u) s_max[wave] = mx;
    __syncthreads();
    if (R == 4 && tid < 256) {
        fields_out[512u + tid] = s_c[tid];
        fields_out[768u + tid] = s_d[tid];
    }
    uint32_t carry = 0, total = 0, largest = 0;
#pragma unroll
    for (int w = 0; w < 16; w++) {
        carry += (uint32_t)w < wave ? s_wave[w] : 0u;
        total += s_wave[w];
        largest = s_max[w] > largest ? s_max[w] : largest;
    }
    uint32_t run = carry + incl - sum;
#pragma unroll
    for (int j = 0; j < (int)PER; j++) {
        bases[(size_t)tid * PER + j] = run;
        run += cnt[j];
    }
    if (tid == 0) {
        bases[1024 * PER] = n;
        const uint32_t ok = (largest <= (uint32_t)kLocalSortCap && total == n && words[kHybridWordViolated] == 0u) ? 1u : 0u;
        // what follows from the key prefix (the upfront read counted below it; if it did not run, nothing here is used)
        uint32_t prefix = hybrid_prefix_of(words[kHybridWordDiffer]);
        if (prefix > kHybridMaxPrefix) prefix = 0;
        words[kHybridWordPrefix] = prefix;
#pragma unroll
        for (int g = 0; g < 4; g++) words[kHybridWordShift + g] = 16u - prefix + (uint32_t)(g * R);
        words[kHybridWordLowBits] = 32u - prefix - (PER == 64 ? 16u : PER == 32 ? 15u : 14u);
        words[kHybridWordOk] = ok;            // the ordinary form's kernels return at once when this is set
        words[kHybridWordSkipLocal] = ok ^ 1u;
        words[kHybridWordLargeCount] = s_large;
#pragma unroll
        for (int g = 0; g < 4; g++) {          // PassParams::plan of the g-th global pass: skip?, roles swapped?
            words[kHybridWordPlan + 2 * g] = ok ^ 1u;
            words[kHybridWordPlan + 2 * g + 1] = (uint32_t)(g & 1);   // an odd pass reads what the one before it wrote
        }
        words[kHybridWordLargest] = largest;
    }
}

hipError_t launch_hybrid_plan(int radix_bits, const uint32_t* bucket, uint32_t n, int bucket_bits, uint32_t* bases, uint32_t* fields_out,
                              const uint32_t* joint, uint32_t* words, uint32_t* large_list, uint32_t small_cap, hipStream_t stream)
{
    if ((radix_bits != 8 && radix_bits != 4) || (radix_bits == 4 && !joint)) return hipErrorInvalidValue;
#define LSD_PLAN(R, PER) hipLaunchKernelGGL((hybrid_plan_kernel<R, PER>), dim3(1), dim3(1024), 0, stream, bucket, n, bases, fields_out, joint, words, large_list, small_cap)
    if (bucket_bits == 16) {
        if (radix_bits == 8) LSD_PLAN(8, 64); else LSD_PLAN(4, 64);
    } else if (bucket_bits == 15) {
        if (radix_bits == 8) LSD_PLAN(8, 32); else LSD_PLAN(4, 32);
    } else if (bucket_bits == 14) {
        if (radix_bits == 8) LSD_PLAN(8, 16); else LSD_PLAN(4, 16);
    } else {
        return hipErrorInvalidValue;
    }
#undef LSD_PLAN
    return hipGetLastError();
}

}  // namespace lsd
