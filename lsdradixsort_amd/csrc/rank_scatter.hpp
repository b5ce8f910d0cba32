// rank_scatter.hpp -- stage 3 of a pass: stable rank inside a tile, then scatter.
//
// Stands in for LSDRadixSortKernel (.cu:795-837) + SMEMLSDBinaryRadixSort (.cu:373-402).
// The reference sorts the tile with r one-bit Blelloch splits (r*(5+2*log2 B) barriers) and
// then looks dst up from two offset tables.  Here, per tile of T threads x K keys:
//
//   1. coalesced load, wave-striped: wave w owns keys [w*64K, (w+1)*64K) of the tile and lane
//      l's i-th register holds key w*64K + i*64 + l, so (register row, lane) order == key order.
//   2. intra-wave stable rank, one register row at a time, against wave-private LDS counters
//      (waves never touch each other's, so this phase has no workgroup barrier).  Three ways
//      to learn a lane's rank among the lanes holding the same digit (its peers), RANK =
//        kRankBallot : R wave-wide ballots give the peer mask; rank = counter + v_mbcnt(mask);
//                      the lowest peer advances the counter by popcount(mask).
//        kRankLdsOr  : each lane ORs its lane bit into a wave-private LDS word table[digit]
//                      (ds_or_b64: commutative, so order-free) and reads the mask back; rest as
//                      above.  Cheaper than 8 ballots for 8-bit digits.
//        kRankLdsAdd : one returning ds_add_rtn_u32 on counter[digit] per key IS the rank:
//                      gfx950 serves the colliding lanes of one wave instruction in lane order
//                      (tools/experiments/lds_atomic_order.hip: 4e10 lane-ops, 0 exceptions;
//                      the library re-checks this on the device before first use and falls
//                      back to the mask forms if it ever fails).  One LDS op per key.
//   3. one thread per digit sums the W wave counters: per-wave bases, tile digit totals,
//      exclusive scan over digits = the tile's local offsets (BlockPrefixSumKernel as launched
//      at .cu:869).
//   4. tile base per digit ("global offsets", .cu:885-894):
//        chained: publish the tile's digit totals, look back over earlier tiles' status words
//                 (a window of kLookback predecessors per step, loads issued before the LDS
//                 reorder and consumed after it) until an inclusive prefix is met, publish ours;
//        staged : read global_off[tile][digit].
//   5. keys go to LDS at their tile-sorted position (local offset + wave base + rank), are
//      read back in linear order and stored to  dst = pos - local[d] + global[d]  (.cu:833):
//      every digit's keys of the tile form one contiguous run in global memory.  The LDS
//      buffer holds CAP positions; a tile larger than that is reordered in TILE/CAP rounds by
//      position, which lengthens the runs (what HBM write efficiency depends on) without
//      growing the LDS footprint.
//   6. key/value: the payload takes the same LDS slot and the same dst.
//
// Tail tile: missing keys are 0xFFFFFFFF; they carry the highest digit and the highest
// positions, so they sort to the end of the tile and are neither counted nor stored.
#pragma once
#include "lsd_device.hpp"
#include "lsd_kernels.hpp"
#include <cstdio>

namespace lsd {

#ifndef LSD_PERSIST
#define LSD_PERSIST 0
#endif

template <int R, int T, int K, int CAP, int RANK>
constexpr int rank_scatter_lds_words()
{
    constexpr int H = 1 << R;
    constexpr int W = T / kWave;
    constexpr int keys_words = CAP;
    constexpr int tab_words = RANK == kRankLdsOr ? W * H * 2 : 0;
    constexpr int buf = keys_words > tab_words ? keys_words : tab_words;
    return buf + W * H + H + 32;
}

// LDS-only workgroup barrier: waits for this wave's LDS traffic, not for its global loads, so
// look-back loads issued before it stay in flight across it.
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// Diagnostic build only (make STATS=1 -> liblsdsort_stats.so, never the product): wave 0 of every
// tile stamps s_memrealtime (100 MHz) at phase boundaries and adds the differences to
// a per-tile record p.stats[tile][0..6]; [7] look-back refills, [8] empty polls (thread 0's digit).
#ifdef LSD_PHASE_STATS
#define LSD_STAMP(idx)                                                                      \
    do {                                                                                    \
        const unsigned long long now__ = __builtin_amdgcn_s_memrealtime();                  \
        if (tid == 0 && p.stats) p.stats[(size_t)stat_row__ * 10 + (idx)] = now__ - stamp__; \
        stamp__ = now__;                                                                    \
    } while (0)
#define LSD_COUNT(idx, v)                                                                   \
    do {                                                                                    \
        if (p.stats) p.stats[(size_t)stat_row__ * 10 + (idx)] += (unsigned long long)(v);   \
    } while (0)
#else
#define LSD_STAMP(idx) do { } while (0)
#define LSD_COUNT(idx, v) do { } while (0)
#endif

// Register budget: a K=16 tile at 73 VGPRs lands exactly on the 6-waves-per-SIMD step, where a
// third 512-thread workgroup only fits a CU when every SIMD happens to have two free slots; in
// practice two were resident (measured: 480 tiles in flight instead of 768).  Asking for one
// wave more per SIMD than the workgroup count needs keeps the allocation off that edge.
template <int T, int K>
constexpr int min_waves_per_simd()
{
    return K <= 16 ? (T <= 512 ? 7 : 8) : (K <= 24 ? (T <= 512 ? 6 : 4) : (K <= 32 ? 4 : 2));
}

template <int R, int T, int K, int CAP, int RANK, bool PAIRS, bool CHAINED>
__global__ void __launch_bounds__(T, (min_waves_per_simd<T, K>())) rank_scatter_kernel(const PassParams p)
{
#ifdef LSD_PHASE_STATS
    unsigned long long stamp__ = __builtin_amdgcn_s_memrealtime();
    uint32_t stat_row__ = 0;   // status row of the tile being stamped
#endif
    constexpr int H = 1 << R;
    constexpr int W = T / kWave;
    constexpr int TILE = T * K;
    constexpr int ROUNDS = TILE / CAP;        // the LDS reorder buffer holds CAP keys at a time
    constexpr int SLOTS = CAP / T;            // read-back slots per thread per round
    constexpr int KEYS_WORDS = CAP;
    constexpr int TAB_WORDS = RANK == kRankLdsOr ? W * H * 2 : 0;
    constexpr int BUF_WORDS = KEYS_WORDS > TAB_WORDS ? KEYS_WORDS : TAB_WORDS;
#ifndef LSD_LOOKBACK_WINDOW
#define LSD_LOOKBACK_WINDOW (H >= 64 ? 4 : 8)
#endif
    constexpr int LB = LSD_LOOKBACK_WINDOW;   // predecessors inspected per look-back step
#ifndef LSD_PERSIST
#define LSD_PERSIST 0
#endif
    constexpr bool PERSIST = CHAINED && (LSD_PERSIST != 0);   // workgroups loop over tiles, prefetching the next
    static_assert(T % kWave == 0 && H <= T, "one thread per digit in the tile scan");
    static_assert(TILE % CAP == 0 && CAP % T == 0 && (CAP & (CAP - 1)) == 0, "rounds must tile the tile");

    // Explicit LDS (address space 3) pointers: the volatile accesses below would otherwise be
    // lowered to flat_* instructions (address-space inference skips volatile operations).
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    lds_u32* const s_base = (lds_u32*)smem;
    lds_u32* const s_keys = s_base;                                   // [CAP]    (phase 5/6, one round at a time)
    volatile lds_u64* const s_tab = (volatile lds_u64*)smem;          // [W][H]   (phase 2, kRankLdsOr, overlays s_keys)
    volatile lds_u32* const s_cnt = (volatile lds_u32*)(s_base + BUF_WORDS);  // [W][H] counters, then wave bases
    lds_u32* const s_gdelta = s_base + BUF_WORDS + W * H;             // [H] global base - local offset
    lds_u32* const s_misc = s_gdelta + H;                             // [1..17] wave totals, [28..29] claimed tile

    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u;
    const uint32_t wave = tid >> 6;
    const uint32_t shift = p.shift;

    // ---- tile acquisition --------------------------------------------------------------------
    // Chained form: the workgroup is persistent and takes tiles from ticket dispensers, one per
    // region.  A tile only ever waits on earlier tickets of the SAME dispenser, i.e. on tiles some
    // workgroup has already taken (as its current tile, or as the one it will start right after its
    // current, lower-numbered one): the lowest unfinished tile is always somebody's current tile,
    // so the look-back cannot deadlock whatever the dispatch order, placement or residency (the
    // MI355X guide: never assume any of them).  A workgroup serves the region of its own XCD first
    // (hardware XCC_ID) and the others when that one is used up: placement is for speed only
    // (neighbouring runs meet in one L2; each XCD walks its own short chain).
    auto claim = [&]() {   // thread 0: take the next ticket, leave (region, ticket) in LDS
        uint32_t xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        uint32_t got = 0xFFFFFFFFu, ticket = 0;
        for (uint32_t a = 0; a < (uint32_t)kRegions; a++) {
            const uint32_t x = (xcc + a) & (uint32_t)(kRegions - 1);
            const uint32_t region_tiles = p.regions[16 + x];
            if (region_tiles == 0) continue;
            ticket = atomicAdd(p.tickets + x, 1u);
            if (ticket < region_tiles) {
                got = x;
                break;
            }
        }
        s_misc[28] = got;
        s_misc[29] = ticket;
    };

    uint32_t tile = 0;        // row of the tile in the status array
    uint32_t chain_pos = 0;   // its position in its region's chain
    uint32_t tile_base = 0;   // index of its first key
    uint32_t range_end = 0;   // one past the last key it may touch
    uint32_t region = 0;
    uint32_t chain_row0 = 0;
    auto adopt = [&]() -> bool {   // all threads, after a barrier that follows claim()
        region = __builtin_amdgcn_readfirstlane(s_misc[28]);
        if (region == 0xFFFFFFFFu) return false;
        chain_pos = __builtin_amdgcn_readfirstlane(s_misc[29]);
        const uint32_t r_start = p.regions[region];
        chain_row0 = p.regions[24 + region];
        tile = chain_row0 + chain_pos;
        tile_base = r_start + chain_pos * (uint32_t)TILE;
        range_end = r_start + p.regions[8 + region];
        return true;
    };

    if (CHAINED) {
#ifdef LSD_STAGGER
        // de-synchronise the persistent workgroups of an XCD: a one-off delay spread over roughly
        // one tile period, so that tiles of one chain reach their look-back one after another
        if (PERSIST) {
            const uint32_t slot = (blockIdx.x >> 3) & 127u;
            for (uint32_t z = 0; z < slot * (uint32_t)LSD_STAGGER; z++) __builtin_amdgcn_s_sleep(16);
        }
#endif
        if (tid == 0) claim();
        __syncthreads();
        if (!adopt()) return;   // uniform: nothing left for this workgroup
#ifdef LSD_PHASE_STATS
        stat_row__ = tile;
        if (tid == 0 && p.stats) p.stats[(size_t)stat_row__ * 10 + 9] = stamp__;
#endif
        LSD_STAMP(0);   // ticket
    } else {
        tile = blockIdx.x;
        // Affinity for the table-driven form, where tiles are independent: blocks with equal
        // blockIdx mod 8 share an XCD under round-robin placement, so give each residue class
        // chunks of C consecutive tiles.
        const uint32_t C = p.xcd_chunk;
        if (C) {
            const uint32_t group = 8u * C;
            const uint32_t g0 = (tile / group) * group;
            if (g0 + group <= p.num_tiles) {
                const uint32_t k = tile - g0;
                tile = g0 + (k % 8u) * C + (k / 8u);
            }
        }
        tile_base = tile * (uint32_t)TILE;
        range_end = p.n;
    }

    // ---- 1. load: wave-striped, so (register row, lane) order == key order --------------------
    uint32_t key[K];
    auto load_keys = [&]() {
        const uint32_t first = tile_base + wave * (uint32_t)(kWave * K) + lane;
        if (range_end - tile_base >= (uint32_t)TILE) {
#pragma unroll
            for (int i = 0; i < K; i++) key[i] = p.in[first + i * kWave];
        } else {
#pragma unroll
            for (int i = 0; i < K; i++) {
                const uint32_t idx = first + i * kWave;
                key[i] = idx < range_end ? p.in[idx] : 0xFFFFFFFFu;
            }
        }
    };
    load_keys();

    for (;;) {
        const uint32_t remaining = range_end - tile_base;
        const uint32_t valid = remaining < (uint32_t)TILE ? remaining : (uint32_t)TILE;
        const bool full = valid == (uint32_t)TILE;
        const uint32_t first = tile_base + wave * (uint32_t)(kWave * K) + lane;
        const uint32_t cur_tile = tile, cur_chain_pos = chain_pos, cur_region = region, cur_end = range_end;
        // this thread's digit column of its region's chain
        const uint32_t* const status_col = CHAINED ? p.status + (size_t)chain_row0 * H + tid : nullptr;
#ifdef LSD_PHASE_STATS
        stat_row__ = cur_tile;
        if (!CHAINED && tid == 0 && p.stats) p.stats[(size_t)stat_row__ * 10 + 9] = stamp__;
#endif

        // wave-private tables start at zero (the previous tile's users are behind a barrier)
#pragma unroll
        for (int j = 0; j < (H + kWave - 1) / kWave; j++) {
            const uint32_t d = j * kWave + lane;
            if (H >= kWave || d < H) {
                s_cnt[wave * H + d] = 0;
                if (RANK == kRankLdsOr) s_tab[wave * H + d] = 0;
            }
        }

        // ---- 2. intra-wave stable rank ------------------------------------------------------
#ifdef LSD_PHASE_STATS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        LSD_STAMP(1);   // key load
#endif
        uint32_t rank[K];
        if (RANK == kRankLdsAdd) {
            // The returned old value is (same-digit keys in earlier rows) + (peers in lower
            // lanes): the K atomics are independent, so they issue back to back.
#pragma unroll
            for (int i = 0; i < K; i++) {
                const uint32_t d = digit_at<R>(key[i], shift);
                rank[i] = __hip_atomic_fetch_add((lds_u32*)&s_cnt[wave * H + d], 1u, __ATOMIC_RELAXED,
                                                 __HIP_MEMORY_SCOPE_WAVEFRONT);
            }
        } else {
            const uint64_t lane_bit = 1ull << lane;
#pragma unroll
            for (int i = 0; i < K; i++) {
                const uint32_t d = digit_at<R>(key[i], shift);
                uint64_t peers;
                if (RANK == kRankLdsOr) {
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
                    if (RANK == kRankLdsOr) s_tab[wave * H + d] = 0;
                }
            }
        }
        __syncthreads();
        LSD_STAMP(2);   // rank + barrier

        // ---- 3. per-wave bases, tile digit totals, local offsets ---------------------------
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

        const uint32_t parity = p.parity;
        const uint32_t c_stale = code_stale(parity);
        const uint32_t c_prefix = code_prefix(parity);
        uint32_t window[LB];
        int32_t j = (int32_t)cur_chain_pos - 1;   // nearest predecessor in the chain not yet consumed
        if (CHAINED && tid < (uint32_t)H) {
            // publish as early as possible: successors can already add this tile's counts
            const uint32_t code = cur_chain_pos == 0 ? c_prefix : code_aggregate(parity);
            store_status(p.status + (size_t)cur_tile * H + tid, (pub_total << 2) | code);
            // first look-back window: issued now, consumed after the LDS reorder below
#pragma unroll
            for (int l = 0; l < LB; l++)
                window[l] = (j - l >= 0) ? load_status(status_col + (size_t)(j - l) * H) : c_stale;
        }

        uint32_t incl = wave_inclusive_scan(tid < (uint32_t)H ? total : 0u, lane);
        if (H > kWave) {
            if (lane == 63u) s_misc[1 + wave] = incl;
            lds_barrier();
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
        lds_barrier();
        LSD_STAMP(3);   // totals, publish, scan, bases

        // payload loads go out now; they land while the keys are reordered
        uint32_t val[PAIRS ? K : 1];
        if (PAIRS) {
            if (full) {
#pragma unroll
                for (int i = 0; i < K; i++) val[i] = p.vals_in[first + i * kWave];
            } else {
#pragma unroll
                for (int i = 0; i < K; i++) {
                    const uint32_t idx = first + i * kWave;
                    val[i] = idx < cur_end ? p.vals_in[idx] : 0u;
                }
            }
        }

        // ---- 5. reorder through LDS in ROUNDS rounds of CAP tile positions each --------------
        // A key whose tile-sorted position is q belongs to round q / CAP, slot q % CAP.  Rounds are
        // by position, not by digit, so their size never depends on the key distribution; the tile
        // (and with it the length of every digit's run in global memory) can exceed the LDS buffer.
        uint32_t pos[K];
#pragma unroll
        for (int i = 0; i < K; i++) {
            const uint32_t d = digit_at<R>(key[i], shift);
            pos[i] = s_cnt[wave * H + d] + rank[i];
        }

        bool more = false;   // another tile has been claimed for this workgroup
#pragma unroll
        for (int round = 0; round < ROUNDS; round++) {
            if (round > 0) lds_barrier();   // the previous round has been read back
#pragma unroll
            for (int i = 0; i < K; i++) {
                if (ROUNDS == 1 || (pos[i] / (uint32_t)CAP) == (uint32_t)round) s_keys[pos[i] % (uint32_t)CAP] = key[i];
            }
            // With the last round written the key registers are free: take the next ticket now, so
            // its latency and the next tile's loads hide behind this tile's look-back and stores.
            if (PERSIST && !PAIRS && round == ROUNDS - 1 && tid == 0) claim();

            if (round == 0) {
                LSD_STAMP(4);   // first round's LDS writes
                // ---- 4. tile base per digit (overlapped with the first round's LDS writes) --------
                if (tid < (uint32_t)H) {
                    uint32_t gbase;
                    if (CHAINED) {
                        uint32_t excl = 0;
                        if (cur_chain_pos > 0) {
                            uint32_t spins = 0;
                            for (;;) {
                                int consumed = 0;
                                bool found = false;
#pragma unroll
                                for (int l = 0; l < LB; l++) {
                                    const uint32_t code = window[l] & 3u;
                                    if (!found && consumed == l && code != c_stale) {
                                        excl += window[l] >> 2;
                                        consumed = l + 1;
                                        found = (code == c_prefix) || (j - l == 0);
                                    }
                                }
                                if (found) break;
                                if (consumed == 0) {
                                    if (++spins > kSpinLimit) {
                                        atomicOr(p.fault, 1u);
                                        break;
                                    }
                                    __builtin_amdgcn_s_sleep(1);
                                    if (tid == 0) LSD_COUNT(8, 1);
                                }
                                if (tid == 0) LSD_COUNT(7, 1);
                                j -= consumed;
#pragma unroll
                                for (int l = 0; l < LB; l++)
                                    window[l] = (j - l >= 0) ? load_status(status_col + (size_t)(j - l) * H) : c_stale;
                            }
                            store_status(p.status + (size_t)cur_tile * H + tid, ((excl + pub_total) << 2) | c_prefix);
                        }
                        gbase = p.regions[kRegionHeaderWords + cur_region * H + tid] + excl;
                    } else {
                        gbase = p.global_off[(size_t)cur_tile * H + tid];
                    }
                    s_gdelta[tid] = gbase - local_off;
                }
            }
            lds_barrier();
            if (round == 0) LSD_STAMP(5);   // look-back (wave 0's digits) + barrier
            if (PERSIST && !PAIRS && round == ROUNDS - 1) {
                more = adopt();
                if (more) load_keys();   // next tile's keys stream in under the stores below
            }

            // linear read-back: consecutive threads hold consecutive tile positions, so each digit's
            // keys leave as one contiguous run
            uint32_t dst[SLOTS];
#pragma unroll
            for (int s2 = 0; s2 < SLOTS; s2++) {
                const uint32_t slot = s2 * T + tid;
                const uint32_t q = round * CAP + slot;
                const uint32_t k = s_keys[slot];
                const uint32_t d = digit_at<R>(k, shift);
                dst[s2] = s_gdelta[d] + q;
                if (full || q < valid) p.out[dst[s2]] = k;
            }

            // ---- 6. payloads follow their keys through the same slots -------------------------
            if (PAIRS) {
                lds_barrier();   // every key of this round has been read back
#pragma unroll
                for (int i = 0; i < K; i++) {
                    if (ROUNDS == 1 || (pos[i] / (uint32_t)CAP) == (uint32_t)round) s_keys[pos[i] % (uint32_t)CAP] = val[i];
                }
                lds_barrier();
#pragma unroll
                for (int s2 = 0; s2 < SLOTS; s2++) {
                    const uint32_t slot = s2 * T + tid;
                    const uint32_t q = round * CAP + slot;
                    if (full || q < valid) p.vals_out[dst[s2]] = s_keys[slot];
                }
            }
        }
#ifdef LSD_PHASE_STATS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        LSD_STAMP(6);   // read-back + stores drained
        if (tid == 0 && p.stats && more) p.stats[(size_t)tile * 10 + 9] = stamp__;   // next tile's start
#endif
        if (!PERSIST) break;
        if (PAIRS) {
            // key/value tiles take their next ticket here (the payload registers stay live to the end)
            if (tid == 0) claim();
            lds_barrier();
            more = adopt();
            if (more) load_keys();
        }
        if (!more) break;
        lds_barrier();   // nobody still reads this tile's LDS when the next one starts writing it
    }
}

// Launch one instantiation.  LDS above 64 KiB needs the attribute raised once per function.
template <int R, int T, int K, int CAP, int RANK, bool PAIRS, bool CHAINED>
hipError_t launch_rank_scatter_inst(const PassParams& p, hipStream_t stream)
{
    constexpr size_t lds_bytes = (size_t)rank_scatter_lds_words<R, T, K, CAP, RANK>() * sizeof(uint32_t);
    auto kernel = rank_scatter_kernel<R, T, K, CAP, RANK, PAIRS, CHAINED>;
    if (lds_bytes > 64 * 1024) {
        static hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (attr != hipSuccess) return attr;
    }
#ifdef LSD_PHASE_STATS
    {
        static bool printed = false;
        if (!printed) {
            printed = true;
            int blocks = -1;
            (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, reinterpret_cast<const void*>(kernel), T, lds_bytes);
            fprintf(stderr, "[lsdsort] rank_scatter<R=%d,T=%d,K=%d,CAP=%d,RANK=%d,PAIRS=%d,CHAINED=%d> lds=%zu occupancy=%d blocks/CU\n",
                    R, T, K, CAP, RANK, (int)PAIRS, (int)CHAINED, lds_bytes, blocks);
        }
    }
#endif
    uint32_t grid = p.num_tiles;
    if (CHAINED && (LSD_PERSIST != 0)) {
        // persistent workgroups: as many as the device holds at once, never more than there are tiles
        static int resident = 0;
        if (resident == 0) {
            int per_cu = 0, dev = 0, cus = 0;
            hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(kernel), T, lds_bytes);
            if (e == hipSuccess) e = hipGetDevice(&dev);
            if (e == hipSuccess) e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
            if (e != hipSuccess) return e;
            resident = (per_cu > 0 ? per_cu : 1) * (cus > 0 ? cus : 1);
        }
        if (grid > (uint32_t)resident) grid = (uint32_t)resident;
    }
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(T), lds_bytes, stream, p);
    return hipGetLastError();
}

template <int R, int T, int K, int CAP, int RANK>
hipError_t launch_rank_scatter_rank(bool chained, const PassParams& p, hipStream_t stream)
{
    const bool pairs = p.vals_in != nullptr;
    if (chained)
        return pairs ? launch_rank_scatter_inst<R, T, K, CAP, RANK, true, true>(p, stream)
                     : launch_rank_scatter_inst<R, T, K, CAP, RANK, false, true>(p, stream);
    return pairs ? launch_rank_scatter_inst<R, T, K, CAP, RANK, true, false>(p, stream)
                 : launch_rank_scatter_inst<R, T, K, CAP, RANK, false, false>(p, stream);
}

// rank_method: kRankLdsAdd, or anything else for the mask form suited to the digit width.
template <int R, int T, int K, int CAP = T * K>
hipError_t launch_rank_scatter_shape(int rank_method, bool chained, const PassParams& p, hipStream_t stream)
{
    if (rank_method == kRankLdsAdd) return launch_rank_scatter_rank<R, T, K, CAP, kRankLdsAdd>(chained, p, stream);
    return launch_rank_scatter_rank<R, T, K, CAP, (R > 4 ? kRankLdsOr : kRankBallot)>(chained, p, stream);
}

}  // namespace lsd
