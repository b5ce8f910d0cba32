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

namespace lsd {

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
#define LSD_STAMP(idx)                                                          \
    do {                                                                        \
        const unsigned long long now__ = __builtin_amdgcn_s_memrealtime();      \
        rec__[(idx)] = now__ - stamp__;                                         \
        stamp__ = now__;                                                        \
    } while (0)
#define LSD_COUNT(idx, v) do { rec__[(idx)] += (unsigned long long)(v); } while (0)
#else
#define LSD_STAMP(idx) do { } while (0)
#define LSD_COUNT(idx, v) do { } while (0)
#endif

template <int R, int T, int K, int CAP, int RANK, bool PAIRS, bool CHAINED>
__global__ void __launch_bounds__(T) rank_scatter_kernel(const PassParams p)
{
#ifdef LSD_PHASE_STATS
    unsigned long long stamp__ = __builtin_amdgcn_s_memrealtime();
    unsigned long long rec__[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};   // per-tile record, stored once at the end
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
            if (RANK == kRankLdsOr) s_tab[wave * H + d] = 0;
        }
    }

    uint32_t tile;
    if (CHAINED) {
        // Tile ids come from a ticket taken on arrival, so a tile only ever waits on tiles whose
        // workgroups have already started (or are among the next few to start): the look-back
        // cannot deadlock whatever the dispatch order, placement or residency (the MI355X guide:
        // never assume any of them).
        //
        // XCD affinity (speed only).  Neighbouring tiles write neighbouring runs; when they run on
        // different XCDs each 64-byte block shared by two runs leaves two L2s as two partial
        // writes, which is what bounds this kernel at 8-bit digits (profiles/: one extra partial
        // HBM write per run).  So the ticket only fixes the GROUP of 8*C consecutive tiles a
        // workgroup works in; inside the group it claims the next tile of the C-tile chunk that
        // belongs to its own XCD (hardware XCC_ID), and moves on to the other chunks of the same
        // group if that one is used up.  A group has exactly as many tickets as tiles and a
        // failed claim only ever hits an exhausted chunk, so every workgroup finds a tile; at most
        // 8*C-1 workgroups can be waiting on tiles nobody has claimed yet, far below residency.
        if (tid == 0) {
            const uint32_t ticket = atomicAdd(p.tile_counter, 1u);
            const uint32_t C = p.xcd_chunk;
            uint32_t t = ticket;
            if (C && ticket < p.num_tiles) {
                const uint32_t group = ticket / (8u * C);
                const uint32_t g0 = group * 8u * C;
                const uint32_t group_tiles = (p.num_tiles - g0 < 8u * C) ? p.num_tiles - g0 : 8u * C;
                uint32_t xcc;
                asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
                for (uint32_t a = 0; a < 8u; a++) {
                    const uint32_t q = (xcc + a) & 7u;
                    const uint32_t q0 = q * C;
                    if (q0 >= group_tiles) continue;
                    const uint32_t cap = (group_tiles - q0 < C) ? group_tiles - q0 : C;
                    const uint32_t jj = atomicAdd(p.chunk_counters + (size_t)group * 8u + q, 1u);
                    if (jj < cap) {
                        t = g0 + q0 + jj;
                        break;
                    }
                }
            }
            s_misc[0] = t;
        }
        __syncthreads();
        tile = __builtin_amdgcn_readfirstlane(s_misc[0]);
        if (tile >= p.num_tiles) return;   // uniform; cannot happen with grid == num_tiles
        LSD_STAMP(0);   // ticket + claim
    } else {
        tile = blockIdx.x;
        // Same affinity for the table-driven form, where tiles are independent: blocks with equal
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
    }

    const uint32_t tile_base = tile * (uint32_t)TILE;
    const uint32_t remaining = p.n - tile_base;
    const uint32_t valid = remaining < (uint32_t)TILE ? remaining : (uint32_t)TILE;
    const bool full = valid == (uint32_t)TILE;
    const uint32_t shift = p.shift;

    // ---- 1. load ------------------------------------------------------------------------
    uint32_t key[K];
    const uint32_t first = tile_base + wave * (uint32_t)(kWave * K) + lane;
    if (full) {
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
#ifdef LSD_PHASE_STATS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    LSD_STAMP(1);   // key load
#endif
    uint32_t rank[K];
    if (RANK == kRankLdsAdd) {
        // The returned old value is (same-digit keys in earlier rows) + (peers in lower lanes):
        // the K atomics are independent, so they issue back to back.
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

    const uint32_t parity = p.parity;
    const uint32_t c_stale = code_stale(parity);
    const uint32_t c_prefix = code_prefix(parity);
    const uint32_t* const status_col = CHAINED ? p.status + tid : nullptr;   // this thread's digit column
    uint32_t window[LB];
    int32_t j = (int32_t)tile - 1;   // nearest predecessor not yet consumed
    if (CHAINED && tid < (uint32_t)H) {
        // publish as early as possible: successors can already add this tile's counts
        const uint32_t code = tile == 0 ? c_prefix : code_aggregate(parity);
        store_status(p.status + (size_t)tile * H + tid, (pub_total << 2) | code);
        // first look-back window: issued now, consumed after the LDS reorder below
#pragma unroll
        for (int l = 0; l < LB; l++) window[l] = (j - l >= 0) ? load_status(status_col + (size_t)(j - l) * H) : c_stale;
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
                val[i] = idx < p.n ? p.vals_in[idx] : 0u;
            }
        }
    }

    // ---- 5. reorder through LDS in ROUNDS rounds of CAP tile positions each ------------------
    // A key whose tile-sorted position is q belongs to round q / CAP, slot q % CAP.  Rounds are
    // by position, not by digit, so their size never depends on the key distribution; the tile
    // (and with it the length of every digit's run in global memory) can exceed the LDS buffer.
    uint32_t pos[K];
#pragma unroll
    for (int i = 0; i < K; i++) {
        const uint32_t d = digit_at<R>(key[i], shift);
        pos[i] = s_cnt[wave * H + d] + rank[i];
    }

#pragma unroll
    for (int round = 0; round < ROUNDS; round++) {
        if (round > 0) lds_barrier();   // the previous round has been read back
#pragma unroll
        for (int i = 0; i < K; i++) {
            if (ROUNDS == 1 || (pos[i] / (uint32_t)CAP) == (uint32_t)round) s_keys[pos[i] % (uint32_t)CAP] = key[i];
        }

        if (round == 0) {
            LSD_STAMP(4);   // first round's LDS writes
            // ---- 4. tile base per digit (overlapped with the first round's LDS writes) ------------
            if (tid < (uint32_t)H) {
                uint32_t gbase;
                if (CHAINED) {
                    uint32_t excl = 0;
                    if (tile > 0) {
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
                        store_status(p.status + (size_t)tile * H + tid, ((excl + pub_total) << 2) | c_prefix);
                    }
                    gbase = p.digit_base[tid] + excl;
                } else {
                    gbase = p.global_off[(size_t)tile * H + tid];
                }
                s_gdelta[tid] = gbase - local_off;
            }
        }
        lds_barrier();
        if (round == 0) LSD_STAMP(5);   // look-back (wave 0's digits) + barrier

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
#if defined(LSD_STORE_KIND) && LSD_STORE_KIND == 1
            if (full || q < valid) __builtin_nontemporal_store(k, p.out + dst[s2]);
#elif defined(LSD_STORE_KIND) && LSD_STORE_KIND == 2
            if (full || q < valid) __hip_atomic_store(p.out + dst[s2], k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
            if (full || q < valid) p.out[dst[s2]] = k;
#endif
        }

        // ---- 6. payloads follow their keys through the same slots -----------------------------
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
    if (tid == 0 && p.stats) {
        for (int i = 0; i < 10; i++) p.stats[(size_t)tile * 10 + i] = rec__[i];
    }
#endif
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
    hipLaunchKernelGGL(kernel, dim3(p.num_tiles), dim3(T), lds_bytes, stream, p);
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
