// rank_scatter.hpp -- stage 3 of a pass: stable rank inside a tile, then scatter.
//
// Stands in for LSDRadixSortKernel (.cu:795-837) + SMEMLSDBinaryRadixSort (.cu:373-402).
// The reference sorts the tile with r one-bit Blelloch splits (r*(5+2*log2 B) barriers) and
// then looks dst up from two offset tables.  Here, per tile of T threads x K keys:
//
//   1. coalesced load, wave-striped: wave w owns keys [w*64K, (w+1)*64K) of the tile and lane
//      l's i-th register holds key w*64K + i*64 + l, so (register row, lane) order == key order.
//   2. intra-wave stable rank, one register row at a time.  For each row every lane needs the
//      set of lanes holding the same digit (its peers):
//        R <= 4 : R wave-wide ballots (match_ballot)
//        R  > 4 : each lane ORs its lane bit into a wave-private LDS word table[digit]
//                 (ds_or_b64, commutative => order-free and deterministic) and reads it back.
//      rank = (same-digit keys in earlier rows of this wave, a wave-private LDS counter)
//           + (peers in lower lanes, v_mbcnt on the peer mask).
//      The lowest peer advances the counter by the peer count and clears the table word.
//      Waves never touch each other's tables, so this phase has no workgroup barrier.
//   3. one thread per digit sums the W wave counters: per-wave bases, tile digit totals,
//      exclusive scan over digits = the tile's local offsets (BlockPrefixSumKernel as launched
//      at .cu:869).
//   4. tile base per digit ("global offsets", .cu:885-894):
//        chained: publish the tile's digit totals, look back over earlier tiles' status words
//                 until an inclusive prefix is met, publish our inclusive prefix;
//        staged : read global_off[tile][digit].
//   5. keys go to LDS at their tile-sorted position (local offset + wave base + rank), are
//      read back in linear order and stored to  dst = pos - local[d] + global[d]  (.cu:833):
//      every digit's keys of the tile form one contiguous run in global memory.
//   6. key/value: the payload takes the same LDS slot and the same dst.
//
// Tail tile: missing keys are 0xFFFFFFFF; they carry the highest digit and the highest
// positions, so they sort to the end of the tile and are neither counted nor stored.
#pragma once
#include "lsd_device.hpp"
#include "lsd_kernels.hpp"

namespace lsd {

template <int R>
constexpr bool use_lds_match() { return R > 4; }

template <int R, int T, int K>
constexpr int rank_scatter_lds_words()
{
    constexpr int H = 1 << R;
    constexpr int W = T / kWave;
    constexpr int keys_words = T * K;
    constexpr int tab_words = use_lds_match<R>() ? W * H * 2 : 0;
    constexpr int buf = keys_words > tab_words ? keys_words : tab_words;
    return buf + W * H + H + 32;
}

template <int R, int T, int K, bool PAIRS, bool CHAINED>
__global__ void __launch_bounds__(T) rank_scatter_kernel(const PassParams p)
{
    constexpr int H = 1 << R;
    constexpr int W = T / kWave;
    constexpr int TILE = T * K;
    constexpr bool LDS_MATCH = use_lds_match<R>();
    constexpr int KEYS_WORDS = TILE;
    constexpr int TAB_WORDS = LDS_MATCH ? W * H * 2 : 0;
    constexpr int BUF_WORDS = KEYS_WORDS > TAB_WORDS ? KEYS_WORDS : TAB_WORDS;
    static_assert(T % kWave == 0 && H <= T, "one thread per digit in the tile scan");
    static_assert(TILE <= 65536, "tile positions must fit the LDS budget");

    // Explicit LDS (address space 3) pointers: the volatile accesses below would otherwise be
    // lowered to flat_* instructions (address-space inference skips volatile operations).
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    lds_u32* const s_base = (lds_u32*)smem;
    lds_u32* const s_keys = s_base;                                   // [TILE]   (phase 5/6)
    volatile lds_u64* const s_tab = (volatile lds_u64*)smem;          // [W][H]   (phase 2, overlays s_keys)
    volatile lds_u32* const s_cnt = (volatile lds_u32*)(s_base + BUF_WORDS);  // [W][H] counters, then wave bases
    lds_u32* const s_gdelta = s_base + BUF_WORDS + W * H;             // [H] global base - local offset
    lds_u32* const s_misc = s_gdelta + H;                             // [0] tile id, [1..] wave totals

    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u;
    const uint32_t wave = tid >> 6;

    // wave-private tables start at zero
#pragma unroll
    for (int j = 0; j < (H + kWave - 1) / kWave; j++) {
        const uint32_t d = j * kWave + lane;
        if (H >= kWave || d < H) {
            s_cnt[wave * H + d] = 0;
            if (LDS_MATCH) s_tab[wave * H + d] = 0;
        }
    }

    uint32_t tile;
    if (CHAINED) {
        // Tile ids are handed out in arrival order, so a tile only ever waits on tiles whose
        // workgroups have already started: the look-back cannot deadlock whatever the
        // dispatch order or residency (MI355X guide: never assume either).
        if (tid == 0) s_misc[0] = atomicAdd(p.tile_counter, 1u);
        __syncthreads();
        tile = __builtin_amdgcn_readfirstlane(s_misc[0]);
        if (tile >= p.num_tiles) return;   // uniform; cannot happen with grid == num_tiles
    } else {
        tile = blockIdx.x;
    }

    const uint32_t tile_base = tile * (uint32_t)TILE;
    const uint32_t remaining = p.n - tile_base;
    const uint32_t valid = remaining < (uint32_t)TILE ? remaining : (uint32_t)TILE;
    const uint32_t shift = p.shift;

    // ---- 1. load ------------------------------------------------------------------------
    uint32_t key[K];
    const uint32_t first = tile_base + wave * (uint32_t)(kWave * K) + lane;
    if (valid == (uint32_t)TILE) {
#pragma unroll
        for (int i = 0; i < K; i++) key[i] = p.in[first + i * kWave];
    } else {
#pragma unroll
        for (int i = 0; i < K; i++) {
            const uint32_t idx = first + i * kWave;
            key[i] = idx < p.n ? p.in[idx] : 0xFFFFFFFFu;
        }
    }

    // ---- 2. intra-wave stable rank ----------------------------------------------------------
    uint32_t rank[K];
    const uint64_t lane_bit = 1ull << lane;
#pragma unroll
    for (int i = 0; i < K; i++) {
        const uint32_t d = digit_at<R>(key[i], shift);
        uint64_t peers;
        if (LDS_MATCH) {
            __hip_atomic_fetch_or((lds_u64*)&s_tab[wave * H + d], lane_bit, __ATOMIC_RELAXED,
                                  __HIP_MEMORY_SCOPE_WAVEFRONT);
            peers = s_tab[wave * H + d];
        } else {
            peers = match_ballot<R>(d);
        }
        const uint32_t before = s_cnt[wave * H + d];
        const uint32_t r = mbcnt_add(peers, before);
        rank[i] = r;
        if (r == before) {   // lowest peer
            s_cnt[wave * H + d] = popc64_add(peers, before);
            if (LDS_MATCH) s_tab[wave * H + d] = 0;
        }
    }
    __syncthreads();

    // ---- 3. per-wave bases, tile digit totals, local offsets -------------------------------
    uint32_t total = 0;
    uint32_t wave_excl[W];
    if (tid < (uint32_t)H) {
#pragma unroll
        for (int w = 0; w < W; w++) {
            wave_excl[w] = total;
            total += s_cnt[w * H + tid];
        }
    }
    // digit totals that other tiles may see: the tail's padding is not data
    uint32_t pub_total = total;
    if (tid == (uint32_t)(H - 1)) pub_total -= (uint32_t)TILE - valid;

    uint32_t* const my_status = CHAINED ? p.status + (size_t)tile * H + tid : nullptr;
    const uint32_t parity = p.parity;
    if (CHAINED && tid < (uint32_t)H) {
        // publish as early as possible: successors can already add this tile's counts
        const uint32_t code = tile == 0 ? code_prefix(parity) : code_aggregate(parity);
        store_status(my_status, (pub_total << 2) | code);
    }

    uint32_t incl = wave_inclusive_scan(tid < (uint32_t)H ? total : 0u, lane);
    if (H > kWave) {
        if (lane == 63u) s_misc[1 + wave] = incl;
        __syncthreads();
        uint32_t carry = 0;
#pragma unroll
        for (int w = 0; w < H / kWave; w++)
            if ((uint32_t)w < wave) carry += s_misc[1 + w];
        incl += carry;
    }
    const uint32_t local_off = incl - total;   // exclusive scan over digits
    if (tid < (uint32_t)H) {
#pragma unroll
        for (int w = 0; w < W; w++) s_cnt[w * H + tid] = local_off + wave_excl[w];
    }

    // ---- 4. tile base per digit ----------------------------------------------------------------
    if (tid < (uint32_t)H) {
        uint32_t gbase;
        if (CHAINED) {
            uint32_t excl = 0;
            if (tile > 0) {
                const uint32_t c_stale = code_stale(parity);
                const uint32_t c_prefix = code_prefix(parity);
                uint32_t j = tile - 1;
                uint32_t spins = 0;
                for (;;) {
                    const uint32_t s = load_status(p.status + (size_t)j * H + tid);
                    const uint32_t code = s & 3u;
                    if (code == c_stale) {
                        if (++spins > kSpinLimit) {
                            atomicOr(p.fault, 1u);
                            break;
                        }
                        __builtin_amdgcn_s_sleep(2);
                        continue;
                    }
                    excl += s >> 2;
                    if (code == c_prefix || j == 0) break;
                    j--;
                }
                store_status(my_status, ((excl + pub_total) << 2) | c_prefix);
            }
            gbase = p.digit_base[tid] + excl;
        } else {
            gbase = p.global_off[(size_t)tile * H + tid];
        }
        s_gdelta[tid] = gbase - local_off;
    }
    __syncthreads();

    // ---- 5. tile-local reorder through LDS, then run-contiguous global stores -----------------
    uint32_t pos[K];
#pragma unroll
    for (int i = 0; i < K; i++) {
        const uint32_t d = digit_at<R>(key[i], shift);
        pos[i] = s_cnt[wave * H + d] + rank[i];
        s_keys[pos[i]] = key[i];
    }
    __syncthreads();

    uint32_t dst[K];
#pragma unroll
    for (int i = 0; i < K; i++) {
        const uint32_t q = i * T + tid;
        const uint32_t k = s_keys[q];
        const uint32_t d = digit_at<R>(k, shift);
        dst[i] = s_gdelta[d] + q;
        if (q < valid) p.out[dst[i]] = k;
    }

    // ---- 6. payloads follow their keys --------------------------------------------------------
    if (PAIRS) {
        uint32_t val[K];
        if (valid == (uint32_t)TILE) {
#pragma unroll
            for (int i = 0; i < K; i++) val[i] = p.vals_in[first + i * kWave];
        } else {
#pragma unroll
            for (int i = 0; i < K; i++) {
                const uint32_t idx = first + i * kWave;
                val[i] = idx < p.n ? p.vals_in[idx] : 0u;
            }
        }
        __syncthreads();   // every key has been read back
#pragma unroll
        for (int i = 0; i < K; i++) s_keys[pos[i]] = val[i];
        __syncthreads();
#pragma unroll
        for (int i = 0; i < K; i++) {
            const uint32_t q = i * T + tid;
            if (q < valid) p.vals_out[dst[i]] = s_keys[q];
        }
    }
}

// Launch one instantiation.  LDS above 64 KiB needs the attribute raised once per function.
template <int R, int T, int K, bool PAIRS, bool CHAINED>
hipError_t launch_rank_scatter_inst(const PassParams& p, hipStream_t stream)
{
    constexpr size_t lds_bytes = (size_t)rank_scatter_lds_words<R, T, K>() * sizeof(uint32_t);
    auto kernel = rank_scatter_kernel<R, T, K, PAIRS, CHAINED>;
    if (lds_bytes > 64 * 1024) {
        static hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (attr != hipSuccess) return attr;
    }
    hipLaunchKernelGGL(kernel, dim3(p.num_tiles), dim3(T), lds_bytes, stream, p);
    return hipGetLastError();
}

template <int R, int T, int K>
hipError_t launch_rank_scatter_shape(bool chained, const PassParams& p, hipStream_t stream)
{
    const bool pairs = p.vals_in != nullptr;
    if (chained)
        return pairs ? launch_rank_scatter_inst<R, T, K, true, true>(p, stream)
                     : launch_rank_scatter_inst<R, T, K, false, true>(p, stream);
    return pairs ? launch_rank_scatter_inst<R, T, K, true, false>(p, stream)
                 : launch_rank_scatter_inst<R, T, K, false, false>(p, stream);
}

}  // namespace lsd
