// local_sort.hip -- the last stage of the hybrid form: buckets that fit the LDS, one workgroup each.
//
// No counterpart in the reference (its every pass goes through global memory, LSDRadixSort.cu:839-910).  After the hybrid form's two
// global passes on the key's two high bytes the array is sorted by its top 16 bits, i.e. cut into 2^15 buckets of equal top-15-bit
// value whose sizes the upfront read has counted exactly (hybrid.hip).  A bucket of at most kLocalCap keys is then finished
// where it lies: one 512-thread workgroup loads it, runs the remaining digit passes -- the same stable rank (one returning LDS add
// per key against wave-private counters) and the same LDS reorder as a global pass, but from LDS to LDS -- and stores it back
// in place.  8 B/key of HBM traffic for the low 17 bits instead of 16 B/key for two more global passes, and no chained scan:
// buckets are independent.
//
// Buckets are the top 15 or the top 14 bits (lsd_kernels.hpp hybrid_bucket_bits), below a key prefix where the caller has named one;
// 4-bit-digit sorts come here after four global passes instead of two.
//
// Per workgroup (T = 512 threads, up to K = 32 keys each, two workgroups per CU):
//   load, wave-striped: wave w owns positions [w * rows * 64, (w + 1) * rows * 64) of the bucket, lane l's i-th register holds
//                       position w * rows * 64 + i * 64 + l, rows = ceil(size / T); positions past the bucket hold 0xFFFFFFFF
//                       (the highest digit in every pass and the highest positions: they stay at the end and are never stored)
//   per digit pass    : zero the wave's counters | rank = returning add | barrier | one thread per digit: wave bases + exclusive
//                       scan over digits | barrier | keys -> LDS at (base + rank) | barrier | read back in position order
//   store             : from LDS, linear.
// Counters are 16 bits wide, two to a word (a wave holds at most 2048 keys): 8 waves x 512 digits in 8 KiB, so that two
// workgroups share a CU (72 KiB each) and one's loads and barriers hide under the other's LDS work.
#include "lsd_device.hpp"
#include "lsd_kernels.hpp"

namespace lsd {

typedef __attribute__((address_space(3))) uint16_t lds_u16;

constexpr int kLocalThreads = 512;
constexpr int kLocalWaves = kLocalThreads / kWave;
constexpr int kLocalMaxBins = 512;
static_assert(kLocalThreads * 32 == kLocalSortCap, "the capacity the planner checks buckets against");
static_assert(kLocalThreads * 20 == kLocalSortCapSmall && kLocalThreads * 16 == kLocalSortCapSmallPairs, "the capacities of the three-per-CU variants");
template <int K>
constexpr size_t local_lds_words() { return (size_t)kLocalThreads * K + kLocalWaves * (kLocalMaxBins / 2) + 64; }

// K = 32: buckets of up to 16384 keys, 72 KiB of LDS, two workgroups per CU.  K = 20: up to 10240 keys, 48 KiB, THREE per CU
// (and at most 80 registers): the stage is bound by LDS work that one workgroup's barriers and loads leave idle, so the third
// resident workgroup is worth about a fifth of its time.  The planner knows the largest bucket and picks (LocalSortParams::skip
// of the other launch); uniform keys at 2^28 have buckets of 8192 +- 300.
// PAIRS: a payload word follows each key (LocalSortParams::vals).  It takes the key's LDS slot in a second round of every pass, as
// in the global pass kernel: keys to LDS, keys back, payloads to the same slots, payloads back -- two more barriers per pass.
// WHOLE: the bucket is the whole array [0, p.num_buckets) and a fourth digit pass may follow (launch_small_sort)
template <int K, bool PAIRS, bool XOUT, bool WHOLE = false>
__device__ __forceinline__ void sort_bucket(const LocalSortParams& p, const uint32_t b)
{
    constexpr int T = kLocalThreads, W = kLocalWaves, HW = kLocalMaxBins / 2;   // HW: counter words per wave
    constexpr int CAP = T * K;
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    lds_u32* const s_keys = (lds_u32*)smem;                               // [CAP]
    volatile lds_u32* const s_cnt = (volatile lds_u32*)(s_keys + CAP);   // [W][HW] words = [W][512] 16-bit counters, then bases
    volatile lds_u16* const s_cnt16 = (volatile lds_u16*)s_cnt;
    lds_u32* const s_misc = (lds_u32*)(s_cnt + W * HW);

    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t lo = WHOLE ? 0u : p.bases[b], hi = WHOLE ? p.num_buckets : p.bases[b + 1];
    const uint32_t size = hi - lo;
    if (size == 0u || hi < lo) return;
    if (size > (uint32_t)CAP) {
        // the small variant leaves larger buckets to the listed launch; above the large capacity the planner promised
        // otherwise: say so, touch nothing
        if (!p.larger_elsewhere && tid == 0 && p.fault) atomicOr(p.fault, 8u);
        return;
    }
    uint32_t* const bucket = p.keys + lo;
    const uint32_t rows = (size + (uint32_t)T - 1u) / (uint32_t)T;   // uniform, 1 .. K
    const uint32_t wbase = wave * rows * 64u + lane;

    uint32_t* const bucket_vals = PAIRS ? p.vals + lo : nullptr;
    uint32_t key[K], rank[K], val[PAIRS ? K : 1];
#pragma unroll
    for (int i = 0; i < K; i++) {
        key[i] = 0xFFFFFFFFu;   // rows past the bucket's last are never ranked or stored; the paired scatter below looks at one
        if ((uint32_t)i < rows) {
            const uint32_t pos = wbase + (uint32_t)i * 64u;
            if (pos < size) key[i] = bucket[pos];
            if (PAIRS) val[i] = pos < size ? bucket_vals[pos] : 0u;
        }
    }

    auto digit_pass = [&](uint32_t shift, uint32_t width) {
        const uint32_t bins = 1u << width, mask = bins - 1u;
        // this wave's counters start at zero (its own words only: LDS operations of a wave are served in order, and nobody
        // else reads them before the barrier below)
#pragma unroll
        for (int j = 0; j < HW / kWave; j++) s_cnt[wave * HW + j * kWave + lane] = 0;
#pragma unroll
        for (int i = 0; i < K; i++) {
            if ((uint32_t)i < rows) {
                const uint32_t d = (key[i] >> shift) & mask;
                const uint32_t sh = (d & 1u) * 16u;
                const uint32_t old = __hip_atomic_fetch_add((lds_u32*)&s_cnt[wave * HW + (d >> 1)], 1u << sh, __ATOMIC_RELAXED,
                                                            __HIP_MEMORY_SCOPE_WAVEFRONT);
                rank[i] = (old >> sh) & 0xFFFFu;
            }
        }
        __syncthreads();
        // one thread per digit: the waves' counts become their bases inside the digit's range, the digits' totals an exclusive scan
        uint32_t total = 0;
        uint32_t wave_excl[W];
        if (tid < bins) {
#pragma unroll
            for (int w = 0; w < W; w++) {
                wave_excl[w] = total;
                total += s_cnt16[w * kLocalMaxBins + tid];
            }
        }
        uint32_t incl = wave_inclusive_scan(tid < bins ? total : 0u, lane);
        if (lane == 63u) s_misc[wave] = incl;
        __syncthreads();
        uint32_t part[W];
#pragma unroll
        for (int w = 0; w < W; w++) part[w] = s_misc[w];
        uint32_t carry = 0;
#pragma unroll
        for (int w = 0; w < W; w++) carry += (uint32_t)w < wave ? part[w] : 0u;
        const uint32_t local_off = incl + carry - total;
        if (tid < bins) {
#pragma unroll
            for (int w = 0; w < W; w++) s_cnt16[w * kLocalMaxBins + tid] = (uint16_t)(local_off + wave_excl[w]);
        }
        __syncthreads();
        // two rows per uniform branch: both base reads are in flight before the first write waits for its own (0.590 -> 0.584 ms per
        // 2^28 keys).  A row past the bucket's last reads a counter it never uses -- its key register holds whatever it holds,
        // masked into the table.
        static_assert(K % 2 == 0, "rows are scattered in pairs");
#pragma unroll
        for (int i = 0; i < K; i += 2) {
            if ((uint32_t)i < rows) {
                const uint32_t b0 = s_cnt16[wave * kLocalMaxBins + ((key[i] >> shift) & mask)];
                const uint32_t b1 = s_cnt16[wave * kLocalMaxBins + ((key[i + 1] >> shift) & mask)];
                const uint32_t pos0 = b0 + rank[i];
                if (PAIRS) rank[i] = pos0;   // the payload's slot
                s_keys[pos0] = key[i];
                if ((uint32_t)(i + 1) < rows) {
                    const uint32_t pos1 = b1 + rank[i + 1];
                    if (PAIRS) rank[i + 1] = pos1;
                    s_keys[pos1] = key[i + 1];
                }
            }
        }
        __syncthreads();
    };
    // after a pass the keys are in LDS in their new order.  `last`: they leave for global memory (linear store); otherwise they
    // come back into registers in position order (the next pass's first barrier keeps its LDS writes behind these reads).  With
    // payloads: the same for them, through the same slots, once the keys have been taken out.
    auto take_out = [&](bool last) {
        if (last) {
            // XOUT: a typed sort's keys leave as what they were (int32, float32, descending order).  A kernel of its own: the
            // five instructions per key cost the uint32 sort 0.025 ms of 0.59 when they sat in the one store loop
            for (uint32_t q = tid; q < size; q += (uint32_t)T) bucket[q] = XOUT ? from_sortable(s_keys[q], p.xout) : s_keys[q];
        } else {
#pragma unroll
            for (int i = 0; i < K; i++)
                if ((uint32_t)i < rows) key[i] = s_keys[wbase + (uint32_t)i * 64u];
        }
        if (PAIRS) {
            __syncthreads();   // every key has been taken out
#pragma unroll
            for (int i = 0; i < K; i++)
                if ((uint32_t)i < rows) s_keys[rank[i]] = val[i];
            __syncthreads();
            if (last) {
                for (uint32_t q = tid; q < size; q += (uint32_t)T) bucket_vals[q] = s_keys[q];
            } else {
#pragma unroll
                for (int i = 0; i < K; i++)
                    if ((uint32_t)i < rows) val[i] = s_keys[wbase + (uint32_t)i * 64u];
            }
        }
    };

    // the digits: as given, or (the hybrid form, planned on the device) bits [0, low) in one or two passes: nine bits, then the rest
    uint32_t sh0 = p.shift[0], wd0 = p.width[0], sh1 = p.shift[1], wd1 = p.width[1], wd2 = p.width[2];
    if (p.low_bits_word) {
        const uint32_t low = *p.low_bits_word;   // uniform
        sh0 = 0u; wd0 = low < 9u ? low : 9u;
        sh1 = wd0; wd1 = low - wd0;
        wd2 = 0u;
    }
    bool dead0 = false, dead1 = false;
    if (p.low_bits_word && wd1) {
        // Two digits (the hybrid form).  A digit that is the same for every key of the bucket -- dead low bits: keys that are
        // multiples of 512 -- is no pass at all, and as a pass it is the worst one: every lane on one counter, which the LDS
        // serves a lane per clock (2.47 instead of 1.85 ms per 2^28 such keys).  One OR over the bucket says so: key ^ first key,
        // per thread, per wave, then across the waves through LDS.  One of the two passes always runs.
        const uint32_t ref = bucket[0];
        uint32_t diff = 0;
#pragma unroll
        for (int i = 0; i < K; i++)
            if ((uint32_t)i < rows) diff |= wbase + (uint32_t)i * 64u < size ? key[i] ^ ref : 0u;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) diff |= __shfl_xor(diff, off, kWave);
        if (lane == 0u) s_misc[32 + wave] = diff;
        __syncthreads();
#pragma unroll
        for (int w = 0; w < W; w++) diff |= s_misc[32 + w];
        dead0 = ((diff >> sh0) & ((1u << wd0) - 1u)) == 0u;                  // uniform
        dead1 = !dead0 && ((diff >> sh1) & ((1u << wd1) - 1u)) == 0u;
    }
    if (!dead0) {
        digit_pass(sh0, wd0);
        take_out(wd1 == 0u || dead1);
    }
    if (wd1 && !dead1) {
        digit_pass(sh1, wd1);
        take_out(wd2 == 0u);
    }
    if (wd2) {
        digit_pass(p.shift[2], wd2);
        take_out(!WHOLE || p.width[3] == 0u);
    }
    if (WHOLE && wd2 && p.width[3]) {
        digit_pass(p.shift[3], p.width[3]);
        take_out(true);
    }
}

// Several payload arrays (records: lsdsort_multi_u32_device): carrying each of them through every digit pass, as PAIRS does with
// its one, would cost two LDS round trips per array and pass and a register array each.  Instead the KEYS are sorted first (as in
// the keys-only kernel, remembering each element's slot after the first pass), the second pass's slots are left in LDS indexed by
// position, and one random LDS read per element composes the two: final[i] = slot2[slot1[i]].  Each payload array then takes ONE
// trip: loaded in position order, written to its elements' final slots, stored linearly -- in place, the bucket being this
// workgroup's alone (every load of an array is in registers before the barrier that precedes its first store).  At most two
// digit passes (the hybrid form's local stage has two).
template <int K>
__device__ __forceinline__ void sort_bucket_multi(const LocalSortParams& p, const uint32_t b)
{
    constexpr int T = kLocalThreads, W = kLocalWaves, HW = kLocalMaxBins / 2;
    constexpr int CAP = T * K;
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    lds_u32* const s_keys = (lds_u32*)smem;
    volatile lds_u32* const s_cnt = (volatile lds_u32*)(s_keys + CAP);
    volatile lds_u16* const s_cnt16 = (volatile lds_u16*)s_cnt;
    lds_u32* const s_misc = (lds_u32*)(s_cnt + W * HW);

    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t lo = p.bases[b], hi = p.bases[b + 1];
    const uint32_t size = hi - lo;
    if (size == 0u || hi < lo) return;
    if (size > (uint32_t)CAP) {
        if (!p.larger_elsewhere && tid == 0 && p.fault) atomicOr(p.fault, 8u);
        return;
    }
    uint32_t* const bucket = p.keys + lo;
    const uint32_t rows = (size + (uint32_t)T - 1u) / (uint32_t)T;
    const uint32_t wbase = wave * rows * 64u + lane;

    uint32_t key[K], slot[K], first_slot[K];
#pragma unroll
    for (int i = 0; i < K; i++) {
        if ((uint32_t)i < rows) {
            const uint32_t pos = wbase + (uint32_t)i * 64u;
            key[i] = pos < size ? bucket[pos] : 0xFFFFFFFFu;
        }
    }
    // one digit pass over key[] (position order): afterwards the keys lie in LDS in their new order and slot[i] says where key[i] went
    auto digit_pass = [&](uint32_t shift, uint32_t width) {
        const uint32_t bins = 1u << width, mask = bins - 1u;
#pragma unroll
        for (int j = 0; j < HW / kWave; j++) s_cnt[wave * HW + j * kWave + lane] = 0;
#pragma unroll
        for (int i = 0; i < K; i++) {
            if ((uint32_t)i < rows) {
                const uint32_t d = (key[i] >> shift) & mask;
                const uint32_t sh = (d & 1u) * 16u;
                const uint32_t old = __hip_atomic_fetch_add((lds_u32*)&s_cnt[wave * HW + (d >> 1)], 1u << sh, __ATOMIC_RELAXED,
                                                            __HIP_MEMORY_SCOPE_WAVEFRONT);
                slot[i] = (old >> sh) & 0xFFFFu;
            }
        }
        __syncthreads();
        uint32_t total = 0;
        uint32_t wave_excl[W];
        if (tid < bins) {
#pragma unroll
            for (int w = 0; w < W; w++) {
                wave_excl[w] = total;
                total += s_cnt16[w * kLocalMaxBins + tid];
            }
        }
        uint32_t incl = wave_inclusive_scan(tid < bins ? total : 0u, lane);
        if (lane == 63u) s_misc[wave] = incl;
        __syncthreads();
        uint32_t carry = 0;
#pragma unroll
        for (int w = 0; w < W; w++) carry += (uint32_t)w < wave ? s_misc[w] : 0u;
        const uint32_t local_off = incl + carry - total;
        if (tid < bins) {
#pragma unroll
            for (int w = 0; w < W; w++) s_cnt16[w * kLocalMaxBins + tid] = (uint16_t)(local_off + wave_excl[w]);
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < K; i++) {
            if ((uint32_t)i < rows) {
                const uint32_t d = (key[i] >> shift) & mask;
                slot[i] += (uint32_t)s_cnt16[wave * kLocalMaxBins + d];
                s_keys[slot[i]] = key[i];
            }
        }
        __syncthreads();
    };
    uint32_t sh0 = p.shift[0], wd0 = p.width[0], sh1 = p.shift[1], wd1 = p.width[1];
    if (p.low_bits_word) {   // the hybrid form: bits [0, low), nine first (see sort_bucket)
        const uint32_t low = *p.low_bits_word;
        sh0 = 0u; wd0 = low < 9u ? low : 9u;
        sh1 = wd0; wd1 = low - wd0;
    }
    digit_pass(sh0, wd0);
    if (wd1) {
#pragma unroll
        for (int i = 0; i < K; i++)
            if ((uint32_t)i < rows) {
                first_slot[i] = slot[i];
                key[i] = s_keys[wbase + (uint32_t)i * 64u];
            }
        digit_pass(sh1, wd1);
    }
    for (uint32_t q = tid; q < size; q += (uint32_t)T) bucket[q] = s_keys[q];   // the keys are done
    if (wd1) {
        __syncthreads();   // the keys have left the LDS: it now holds the second pass's slots by position ...
#pragma unroll
        for (int i = 0; i < K; i++)
            if ((uint32_t)i < rows) s_keys[wbase + (uint32_t)i * 64u] = slot[i];
        __syncthreads();
#pragma unroll
        for (int i = 0; i < K; i++)   // ... and the element that started at position wbase + 64 i ends in slot2[slot1]
            if ((uint32_t)i < rows) slot[i] = s_keys[first_slot[i]];
    }
    for (uint32_t e = 0; e < p.num_payloads; e++) {
        uint32_t* const pay = (e == 0 ? p.vals : p.more[e - 1]) + lo;
        __syncthreads();   // the LDS is free again (slots read, or the previous array stored)
#pragma unroll
        for (int i = 0; i < K; i++) {
            if ((uint32_t)i < rows) {
                const uint32_t pos = wbase + (uint32_t)i * 64u;
                key[i] = pos < size ? pay[pos] : 0u;
            }
        }
#pragma unroll
        for (int i = 0; i < K; i++)
            if ((uint32_t)i < rows) s_keys[slot[i]] = key[i];
        __syncthreads();
        for (uint32_t q = tid; q < size; q += (uint32_t)T) pay[q] = s_keys[q];
    }
}

template <int K>
__global__ void __launch_bounds__(kLocalThreads, (K <= 16 ? 6 : 2)) local_sort_multi_kernel(const LocalSortParams p)
{
    if (p.skip && *p.skip != 0u) return;
    sort_bucket_multi<K>(p, blockIdx.x);
}

// the planner's list, walked by a small grid (a kernel of its own: with both uses in one kernel the body is inlined twice and spills)
template <int K>
__global__ void __launch_bounds__(kLocalThreads, (K <= 16 ? 6 : 2)) local_sort_multi_list_kernel(const LocalSortParams p)
{
    if (p.skip && *p.skip != 0u) return;
    const uint32_t listed = *p.list_count;
    for (uint32_t item = blockIdx.x; item < listed; item += gridDim.x) {
        sort_bucket_multi<K>(p, p.list[item]);
        __syncthreads();
    }
}

template <int K>
static hipError_t launch_local_multi(const LocalSortParams& p, hipStream_t stream)
{
    constexpr size_t lds_bytes = local_lds_words<K>() * sizeof(uint32_t);
    if (p.list) {
        if constexpr (K == 32) {   // the list is the large variant's (three register arrays of 32: one workgroup per CU, as the pairs')
            auto kernel = local_sort_multi_list_kernel<K>;
            static std::atomic<uint64_t> told{0};
        const hipError_t attr = allow_dynamic_lds(reinterpret_cast<const void*>(kernel), lds_bytes, told);
            if (attr != hipSuccess) return attr;
            hipLaunchKernelGGL(kernel, dim3(256), dim3(kLocalThreads), lds_bytes, stream, p);
            return hipGetLastError();
        }
        return hipErrorInvalidValue;
    }
    auto kernel = local_sort_multi_kernel<K>;
    static std::atomic<uint64_t> told{0};
        const hipError_t attr = allow_dynamic_lds(reinterpret_cast<const void*>(kernel), lds_bytes, told);
    if (attr != hipSuccess) return attr;
    hipLaunchKernelGGL(kernel, dim3(p.num_buckets), dim3(kLocalThreads), lds_bytes, stream, p);
    return hipGetLastError();
}

// registers: three workgroups per CU need 80 or fewer (keys K <= 20; pairs K <= 16), two 128; the 16384-pair variant keeps three
// arrays of 32 and gets 256 (one workgroup per CU: it only ever sees the planner's list of outsized buckets)
template <int K, bool PAIRS>
constexpr int local_waves_per_simd() { return K <= 10 ? 8 : (PAIRS ? K <= 16 : K <= 20) ? 6 : (PAIRS ? 2 : 4); }

// one bucket per workgroup
template <int K, bool PAIRS, bool XOUT>
__global__ void __launch_bounds__(kLocalThreads, (local_waves_per_simd<K, PAIRS>())) local_sort_kernel(const LocalSortParams p)
{
    if (p.skip && *p.skip != 0u) return;   // uniform: the plan took the other form
    sort_bucket<K, PAIRS, XOUT>(p, blockIdx.x);
}

// The buckets of the planner's list (those above the small variant's capacity), dealt over a small grid: uniform keys leave
// the list empty, and a launch of 32768 workgroups that each find nothing to do costs 17 us.
template <int K, bool PAIRS, bool XOUT>
__global__ void __launch_bounds__(kLocalThreads, (local_waves_per_simd<K, PAIRS>())) local_sort_list_kernel(const LocalSortParams p)
{
    if (p.skip && *p.skip != 0u) return;
    const uint32_t listed = *p.list_count;
    for (uint32_t item = blockIdx.x; item < listed; item += gridDim.x) {
        sort_bucket<K, PAIRS, XOUT>(p, p.list[item]);
        __syncthreads();   // the next bucket reuses the LDS
    }
}

template <int K, bool PAIRS, bool XOUT>
static hipError_t launch_local_inst_x(const LocalSortParams& p, hipStream_t stream);
template <int K, bool PAIRS>
static hipError_t launch_local_inst(const LocalSortParams& p, hipStream_t stream)
{
    return p.xout.on ? launch_local_inst_x<K, PAIRS, true>(p, stream) : launch_local_inst_x<K, PAIRS, false>(p, stream);
}
template <int K, bool PAIRS, bool XOUT>
static hipError_t launch_local_inst_x(const LocalSortParams& p, hipStream_t stream)
{
    constexpr size_t lds_bytes = local_lds_words<K>() * sizeof(uint32_t);
    if (p.list) {
        if constexpr (K == 32) {   // the list is the large variant's
            auto kernel = local_sort_list_kernel<K, PAIRS, XOUT>;
            static std::atomic<uint64_t> told{0};
        const hipError_t attr = allow_dynamic_lds(reinterpret_cast<const void*>(kernel), lds_bytes, told);
            if (attr != hipSuccess) return attr;
            hipLaunchKernelGGL(kernel, dim3(512), dim3(kLocalThreads), lds_bytes, stream, p);   // two workgroups per CU walk the list
            return hipGetLastError();
        }
        return hipErrorInvalidValue;
    }
    auto kernel = local_sort_kernel<K, PAIRS, XOUT>;
    if (lds_bytes > 64 * 1024) {
        static std::atomic<uint64_t> told{0};
        const hipError_t attr = allow_dynamic_lds(reinterpret_cast<const void*>(kernel), lds_bytes, told);
        if (attr != hipSuccess) return attr;
    }
    hipLaunchKernelGGL(kernel, dim3(p.num_buckets), dim3(kLocalThreads), lds_bytes, stream, p);
    return hipGetLastError();
}

// A sort of up to 16384 items is one workgroup's work: one launch instead of the eight of the chained form (whose kernels are
// all latency at this size: 39 us for 2^14 keys).
template <bool PAIRS>
__global__ void __launch_bounds__(kLocalThreads, (PAIRS ? 2 : 4)) small_sort_kernel(const LocalSortParams p, uint32_t* clear0, uint32_t* clear1)
{
    if (threadIdx.x == 0) {
        if (clear0) *clear0 = 0u;
        if (clear1) *clear1 = 0u;
    }
    sort_bucket<32, PAIRS, false, true>(p, 0u);
}

hipError_t launch_small_sort(uint32_t* keys, uint32_t* vals, uint32_t n, uint32_t* clear0, uint32_t* clear1, hipStream_t stream)
{
    if (!keys || n == 0 || n > (uint32_t)kLocalSortCap) return hipErrorInvalidValue;
    LocalSortParams p{};
    p.keys = keys;
    p.vals = vals;
    p.num_buckets = n;                                  // WHOLE: the array's length
    for (int i = 0; i < 4; i++) {
        p.shift[i] = 8u * (uint32_t)i;
        p.width[i] = 8u;
    }
    constexpr size_t lds_bytes = local_lds_words<32>() * sizeof(uint32_t);
    if (vals) {
        static std::atomic<uint64_t> told{0};
        const hipError_t attr = allow_dynamic_lds(reinterpret_cast<const void*>(small_sort_kernel<true>), lds_bytes, told);
        if (attr != hipSuccess) return attr;
        hipLaunchKernelGGL(small_sort_kernel<true>, dim3(1), dim3(kLocalThreads), lds_bytes, stream, p, clear0, clear1);
    } else {
        static std::atomic<uint64_t> told{0};
        const hipError_t attr = allow_dynamic_lds(reinterpret_cast<const void*>(small_sort_kernel<false>), lds_bytes, told);
        if (attr != hipSuccess) return attr;
        hipLaunchKernelGGL(small_sort_kernel<false>, dim3(1), dim3(kLocalThreads), lds_bytes, stream, p, clear0, clear1);
    }
    return hipGetLastError();
}

hipError_t launch_local_sort(const LocalSortParams& p, hipStream_t stream)
{
    if (p.num_buckets == 0) return hipSuccess;
    if (!p.keys || !p.bases || (p.width[0] == 0 && !p.low_bits_word)) return hipErrorInvalidValue;
    for (int i = 0; i < 3; i++)
        if (p.width[i] > 9 || (p.width[i] && p.shift[i] + p.width[i] > 32)) return hipErrorInvalidValue;
    if (p.width[3]) return hipErrorInvalidValue;   // a fourth pass is launch_small_sort's
    if ((p.list == nullptr) != (p.list_count == nullptr)) return hipErrorInvalidValue;
    if (p.small_variant && p.list) return hipErrorInvalidValue;   // the list is the large variant's
    if (p.num_payloads > 3 || (p.num_payloads > 0 && !p.vals)) return hipErrorInvalidValue;
    if (p.num_payloads > 1) {   // records: several payload arrays
        if (p.width[2] || p.xout.on || !p.more[0] || (p.num_payloads > 2 && !p.more[1])) return hipErrorInvalidValue;
        return p.small_variant ? launch_local_multi<16>(p, stream) : launch_local_multi<32>(p, stream);
    }
    if (p.small_variant == 2) {
        if (p.vals) return launch_local_inst<kLocalSortCapTiny / kLocalThreads, true>(p, stream);
        return launch_local_inst<kLocalSortCapTiny / kLocalThreads, false>(p, stream);
    }
    if (p.vals) {
        if (p.small_variant) return launch_local_inst<kLocalSortCapSmallPairs / kLocalThreads, true>(p, stream);
        return launch_local_inst<32, true>(p, stream);
    }
    if (p.small_variant) return launch_local_inst<kLocalSortCapSmall / kLocalThreads, false>(p, stream);
    return launch_local_inst<32, false>(p, stream);
}

}  // namespace lsd
