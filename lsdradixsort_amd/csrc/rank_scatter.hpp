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
//                 (Lookback::LB rows per step; the first step's loads are issued just before the
//                 LDS writes of phase 5 and consumed after them) until an inclusive prefix is met,
//                 publish ours;
//        staged : read global_off[tile][digit].
//   5. keys go to LDS at their tile-sorted position (local offset + wave base + rank), are
//      read back in linear order and stored to  dst = pos - local[d] + global[d]  (.cu:833):
//      every digit's keys of the tile form one contiguous run in global memory.  The LDS
//      buffer holds CAP positions; a tile larger than that is reordered in TILE/CAP rounds by
//      position, which lengthens the runs (what HBM write efficiency depends on) without
//      growing the LDS footprint.
//   6. key/value: the payload takes the same LDS slot and the same dst.
//
// Tail tile: missing keys are 0xFFFFFFFF; they carry the highest digit any key of the tile can have (H - 1 for a bit
// field, the number of live splitters under the splitter partition) and the highest positions, so they sort to the end of
// the tile and are neither counted nor stored.
#pragma once
#include "lsd_device.hpp"
#include "lsd_kernels.hpp"
#include <cstdio>
#include <type_traits>
#include <hip/hip_ext.h>

namespace lsd {

// LDS-only workgroup barrier: waits for this wave's LDS traffic, not for its global loads, so
// look-back loads issued before it stay in flight across it.
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// Diagnostic build only (make STATS=1 -> liblsdsort_stats.so, never the product): wave 0 of every
// tile stamps s_memrealtime (100 MHz) at phase boundaries and adds the differences to
// a per-tile record p.stats[tile][0..6]; [7] look-back refills, [8] empty polls (thread 0's digit).
constexpr int kStatsStride = 16;   // [0..6] phases, [7] refills, [8] empty polls, [9] start, [10] rows walked,
                                   // [11] t(prefix stored), [12] t(prefix met), [13] chain pos it was met at, [14] t(walk start), [15] t(first step consumed)
#ifdef LSD_PHASE_STATS
#define LSD_SET(idx, v)                                                                     \
    do {                                                                                    \
        if (p.stats) p.stats[(size_t)stat_row__ * kStatsStride + (idx)] = (unsigned long long)(v); \
    } while (0)
#define LSD_STAMP(idx)                                                                      \
    do {                                                                                    \
        const unsigned long long now__ = __builtin_amdgcn_s_memrealtime();                  \
        if (tid == 0 && p.stats) p.stats[(size_t)stat_row__ * kStatsStride + (idx)] = now__ - stamp__; \
        stamp__ = now__;                                                                    \
    } while (0)
#define LSD_COUNT(idx, v)                                                                   \
    do {                                                                                    \
        if (p.stats) p.stats[(size_t)stat_row__ * kStatsStride + (idx)] += (unsigned long long)(v);   \
    } while (0)
#else
#define LSD_STAMP(idx) do { } while (0)
#define LSD_COUNT(idx, v) do { } while (0)
#define LSD_SET(idx, v) do { } while (0)
#endif

// Register budget: a K=16 tile at 73 VGPRs lands exactly on the 6-waves-per-SIMD step, where a
// third 512-thread workgroup only fits a CU when every SIMD happens to have two free slots; in
// practice two were resident (measured: 480 tiles in flight instead of 768).  Asking for one
// wave more per SIMD than the workgroup count needs keeps the allocation off that edge.
template <int T, int K>
constexpr int min_waves_per_simd()
{
    return K <= 16 ? (T <= 512 ? 7 : 8) : (K <= 32 ? 4 : 2);
}

// Look-back geometry: LB status rows per thread per step; the first step is taken by up to four
// "slots" of threads at once (thread t: digit t % H, slot t / H), so it covers SLOTS*LB predecessors
// with one round trip.
template <int R, int T, int K = 32>
struct Lookback {
    static constexpr int H = 1 << R;
    // rows per step, measured on 2^28 keys (tools/ab_bench.sh): 2 -> 0.534, 4 -> 0.551, 8 -> 0.553 ms/pass at
    // 8-bit digits: a step costs a round trip whatever its width, but every row is 1 KiB of status reads
    // Small sorts (the 16-keys-per-thread shapes, n < 2^21): every tile of a chain starts at the same moment, nobody has a
    // prefix to offer early, and a tile walks its whole chain of aggregates -- there the width of a step is what counts
    // (LSD_LB_SMALL, tools/size_sweep.py).
#ifndef LSD_LB_SMALL
#define LSD_LB_SMALL 2
#endif
#ifdef LSD_LB   // experiment builds (make variant / stats DEFS=-DLSD_LB=n)
    static constexpr int LB = H >= 64 ? (K <= 16 && T <= 512 ? LSD_LB_SMALL : LSD_LB) : 8;
#else
    static constexpr int LB = H >= 64 ? (K <= 16 && T <= 512 ? LSD_LB_SMALL : 2) : 8;
#endif
    static constexpr int SLOTS = 1;   // measured: helper slots (2 or 4) buy nothing here, the extra barrier costs a little
    static constexpr int LDS_WORDS = (SLOTS - 1) * LB * H;
};

template <int R, int T, int K, int CAP, int RANK>
constexpr int rank_scatter_lds_words()
{
    constexpr int H = 1 << R;
    constexpr int W = T / kWave;
    constexpr int keys_words = CAP;
    constexpr int tab_words = RANK == kRankLdsOr ? W * H * 2 : 0;
    constexpr int buf = keys_words > tab_words ? keys_words : tab_words;
    return buf + W * H + H + 32 + Lookback<R, T, K>::LDS_WORDS;
}

// XF: this launch may carry a key transform (PassParams::xin on a sort's first pass, ::xout on its last);
// plain uint32 sorts use the XF = false instantiations, which contain none of it.
template <int R, int T, int K, int CAP, int RANK, bool PAIRS, bool CHAINED, bool XF = false>
__global__ void __launch_bounds__(T, (min_waves_per_simd<T, K>())) rank_scatter_kernel(const PassParams p)
{
    if (CHAINED && p.plan_first && p.plan[0] == 2u) return;   // uniform (PassParams::plan_first)
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
    constexpr int LB = Lookback<R, T, K>::LB;          // status rows per thread per look-back step
    constexpr int LSLOTS = Lookback<R, T, K>::SLOTS;   // thread slots sharing the first step
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
    lds_u32* const s_misc = s_gdelta + H;                             // [1..17] wave totals, [24..29] claimed tile
    lds_u32* const s_look = s_misc + 32;                              // [LSLOTS-1][LB][H] first-step status rows

    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u;
    const uint32_t wave = tid >> 6;
    const uint32_t shift = p.shift_word ? *p.shift_word : p.shift;   // uniform (the hybrid form's passes: planned on the device)
    // Digit of a key.  The narrow-digit kernels double as the splitter partition (PassParams): there the
    // digit is the key's bucket.  Wider digits compile to the plain bit-field extract.
    auto digit_of = [&](uint32_t k) -> uint32_t {
        if constexpr (R <= 3) {
            if (p.num_splitters) {   // uniform across the grid
                uint32_t b = 0;
#pragma unroll
                for (int i = 0; i < (1 << R) - 1; i++) b += ((uint32_t)i < p.live_splitters && k >= p.splitters[i]) ? 1u : 0u;
                return b;
            }
        }
        return digit_at<R>(k, shift);
    };
    // A key on its way out: the last pass of a typed sort undoes the transform (XF launches only).
    const bool undo = XF && p.xout.on;
    auto leaving = [&](uint32_t k) -> uint32_t {
        if constexpr (XF) return undo ? from_sortable(k, p.xout) : k;
        return k;
    };

    // Heavy digit values of a wave row (see the rank phase): h1 = the value of lane 0 or of lane 32, whichever more lanes
    // share, if at least kHeavy lanes do (a quarter of the wave: below that the atomics cost less than the care);
    // h2 = the first value in the row that differs from it (kNoDigit if there is none).
    constexpr uint32_t kHeavy = 16;
    constexpr uint32_t kNoDigit = 0xFFFFFFFFu;
    auto pick_heavy = [&](uint32_t d, uint32_t d_last, uint32_t& h1, uint32_t& h2) -> bool {
        const uint32_t a = __builtin_amdgcn_readfirstlane(d), b = (uint32_t)__builtin_amdgcn_readlane((int)d, 32);
        const uint64_t ma = __ballot(d == a), mb = __ballot(d == b);
        const uint32_t na = popc64_add(ma, 0u), nb = popc64_add(mb, 0u);
        if ((na > nb ? na : nb) < kHeavy) return false;
        h1 = na >= nb ? a : b;
        const uint64_t rest = ~(na >= nb ? ma : mb);
        h2 = rest != 0ull ? (uint32_t)__builtin_amdgcn_readlane((int)d, (int)__builtin_ctzll(rest)) : kNoDigit;
        // a second value is worth its ballots only if it is frequent as well (eight lanes of the row); otherwise its
        // few holders take their atomics like everybody else and the careful loop counts ONE value
        if (h2 != kNoDigit && popc64_add(__ballot(d == h2), 0u) < 8u) h2 = kNoDigit;
        // Runs (sorted input, or the hybrid form's passes on position-correlated high bits: 4096 keys of one digit in a row, a
        // wave holds 2048): the first row knows only the run the wave STARTS in; half of the waves end in the next one, whose keys
        // would all take the same atomic, 64 lanes on one word, row after row (round 3: 10.7 us of rank phase per tile instead
        // of 3.2).  The wave's last key names that second value.
        if (h2 == kNoDigit) {
            const uint32_t z = (uint32_t)__builtin_amdgcn_readlane((int)d_last, 63);
            if (z != h1) h2 = z;
        }
        return true;
    };
    auto row_is_heavy = [&](uint32_t d) -> bool {
        const uint32_t a = __builtin_amdgcn_readfirstlane(d), b = (uint32_t)__builtin_amdgcn_readlane((int)d, 32);
        const uint32_t na = popc64_add(__ballot(d == a), 0u), nb = popc64_add(__ballot(d == b), 0u);
        return (na > nb ? na : nb) >= kHeavy;
    };

    // Housekeeping for the NEXT pass (it runs in the other status array): the grid's workgroups share the
    // rows out (one each when the grid is the row count).  Called after the look-back, where no load of the
    // wave is waited for any more: memory operations of a wave retire in issue order, so in front of the
    // ticket, the key loads or the status loads these stores' acknowledgements would be waited for as well.
    auto clear_next = [&]() {
        if (CHAINED && p.status_clear) {
            for (uint32_t row = blockIdx.x; row < p.num_tiles; row += gridDim.x)
                for (uint32_t i = tid; i < (uint32_t)H; i += (uint32_t)T) p.status_clear[(size_t)row * H + i] = 0;
        }
    };

    // The pass plan (lsd_kernels.hpp): requested here, first looked at by the thread that is about to take the ticket (a
    // skipped pass takes none) and by everybody behind the ticket's round trip -- a wait for these two words at the top of
    // the kernel would add their latency to every tile (+1 % per sort, measured).
    uint32_t plan_skip = 0, plan_swapped = 0;   // uniform
    if (CHAINED && p.plan) {
        plan_skip = p.plan[0];
        plan_swapped = p.plan[1];
    }

    // wave-private tables start at zero
#pragma unroll
    for (int j = 0; j < (H + kWave - 1) / kWave; j++) {
        const uint32_t d = j * kWave + lane;
        if (H >= kWave || d < H) {
            s_cnt[wave * H + d] = 0;
            if (RANK == kRankLdsOr) s_tab[wave * H + d] = 0;
        }
    }

    // "this tile stores nothing": set by a digit thread whose look-back gives up (bounded spin) or whose destinations
    // would leave the output (destination guard), both below; read by everybody behind the look-back's barrier
    if (tid == 0) s_misc[30] = 0;

    uint32_t tile;             // row of this tile in the status array
    uint32_t chain_pos = 0;    // position in its region's chain (chained form)
    uint32_t tile_base;        // index of the tile's first key
    uint32_t range_end;        // one past the last key this tile may touch
    uint32_t region = 0;
    uint32_t chain_row0 = 0;
    if (CHAINED) {
        // A tile comes from a ticket taken on arrival from its region's dispenser, and it only ever
        // waits on earlier tickets of the SAME dispenser -- workgroups that have already started.
        // So the look-back cannot deadlock whatever the dispatch order, placement or residency (the
        // MI355X guide: never assume any of them).  A workgroup serves the region of its own XCD
        // first (hardware XCC_ID) and moves on to the others when that one is used up: placement is
        // for speed only (neighbouring runs meet in one L2; each XCD walks its own short chain).
        if (wave == 0) {
            uint32_t xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            // home regions of this XCD first (spread over them by block index), then everyone else's
            constexpr uint32_t NREG = (uint32_t)regions_for_radix(R);
            constexpr uint32_t PER_XCD = NREG >= (uint32_t)kXcds ? NREG / (uint32_t)kXcds : 1u;
            const uint32_t home = NREG >= (uint32_t)kXcds ? (xcc & (uint32_t)(kXcds - 1)) * PER_XCD + (blockIdx.x / (uint32_t)kXcds) % PER_XCD : 0u;
            // The home region first, by lane 0 alone: ticket and extents in ONE round trip (the ticket is taken before the tile
            // count is known; an over-run ticket of an exhausted region is harmless).  Uniform keys never get past this.
            uint32_t x = home;
            uint32_t r_tiles = 0, r_start = 0, r_len = 0, row0 = 0, my_ticket = 0;
            if (lane == 0 && !plan_skip) {   // a skipped pass (plan) takes no ticket: "no tile" below sends every workgroup home
                r_tiles = p.regions[2 * kMaxRegions + x];
                r_start = p.regions[x];
                r_len = p.regions[kMaxRegions + x];
                row0 = p.regions[3 * kMaxRegions + x];
                my_ticket = atomicAdd(p.tickets + x, 1u);
            }
            // the extents are needed only by a winning ticket, and the optimiser would sink their loads behind the
            // comparison: a second dependent round trip (0.5 us) per tile
            asm volatile("" : : "v"(r_start), "v"(r_len), "v"(row0));
            uint32_t winner = __ballot(lane == 0 && my_ticket < r_tiles) ? 0u : 64u;   // uniform
            if (NREG > 1 && winner == 64u && !plan_skip) {
                // Home is used up.  Lane a looks at region (home + a) % NREG, all of them in one more round trip: how many
                // tiles, and how far its dispenser has got (a relaxed load: a lower bound, which is all the choice needs).
                // The workgroup then goes straight to the first region that still has tiles instead of finding out one atomic
                // round trip per exhausted region: with skewed digits (one region holding 90 % of a pass's keys) that walk
                // was 7.4 of a tile's 22 us at 4-bit digits, 16 regions (tools/phase_stats.py --radix 4 --heavy 90, round 3).
                const bool mine = lane != 0 && lane < NREG;
                uint32_t seen = 0;
                if (mine) {
                    x = (home + lane) % NREG;
                    r_tiles = p.regions[2 * kMaxRegions + x];
                    r_start = p.regions[x];
                    r_len = p.regions[kMaxRegions + x];
                    row0 = p.regions[3 * kMaxRegions + x];
                    seen = load_status(p.tickets + x);
                }
                uint64_t open = __ballot(mine && seen < r_tiles);   // regions that had tickets left when looked at
                while (winner == 64u && open != 0ull) {
                    const uint32_t l = (uint32_t)__builtin_ctzll(open);
                    uint32_t t = 0xFFFFFFFFu;
                    if (lane == l) t = atomicAdd(p.tickets + x, 1u);
                    if (__ballot(lane == l && t < r_tiles)) {
                        winner = l;
                        if (lane == l) my_ticket = t;
                    } else {
                        open &= ~(1ull << l);   // used up in the meantime
                    }
                }
            }
            if (lane == winner) {   // no lane when there is no tile left
                s_misc[24] = r_start;
                s_misc[25] = r_len;
                s_misc[26] = row0;
                s_misc[29] = my_ticket;
                s_misc[28] = x;
            }
            if (winner == 64u && lane == 0) s_misc[28] = 0xFFFFFFFFu;
        }
        __syncthreads();
        region = __builtin_amdgcn_readfirstlane(s_misc[28]);
        if (region == 0xFFFFFFFFu) {         // uniform: the grid is an upper bound on the tile count
            if (plan_skip != 2u) clear_next();   // 2: the other form of the sort runs (hybrid.hip); this pass owns nothing
            return;
        }
        chain_pos = __builtin_amdgcn_readfirstlane(s_misc[29]);
        const uint32_t r_start = __builtin_amdgcn_readfirstlane(s_misc[24]);
        const uint32_t r_len = __builtin_amdgcn_readfirstlane(s_misc[25]);
        chain_row0 = __builtin_amdgcn_readfirstlane(s_misc[26]);
        tile = chain_row0 + chain_pos;
        tile_base = r_start + chain_pos * (uint32_t)TILE;
        range_end = r_start + r_len;
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
    // The buffers of this pass: with a plan every pass is launched with the same pair, and the plan says whether the pass
    // runs at all (a digit that is the same for every key makes it the identity) and which way round.
    const uint32_t* in_keys = p.in;
    uint32_t* out_keys = p.out;
    const uint32_t* in_vals = p.vals_in;
    uint32_t* out_vals = p.vals_out;
    const bool swapped = CHAINED && p.plan && plan_swapped;
    if (swapped) {
        in_keys = p.out;
        out_keys = const_cast<uint32_t*>(p.in);
        in_vals = p.vals_out;
        out_vals = const_cast<uint32_t*>(p.vals_in);
    }
    // payload array e of this launch (e = 0: in_vals / out_vals above), the way round the plan says
    auto payload_in = [&](uint32_t e) -> const uint32_t* {
        if (e == 0) return in_vals;
        return swapped ? p.more_out[e - 1] : p.more_in[e - 1];
    };
    auto payload_out = [&](uint32_t e) -> uint32_t* {
        if (e == 0) return out_vals;
        return swapped ? const_cast<uint32_t*>(p.more_in[e - 1]) : p.more_out[e - 1];
    };
    const uint32_t num_payloads = PAIRS ? (p.num_payloads > 1u ? p.num_payloads : 1u) : 0u;   // uniform
#ifdef LSD_PHASE_STATS
    stat_row__ = tile;
    if (tid == 0 && p.stats) {
        p.stats[(size_t)stat_row__ * kStatsStride + 9] = stamp__;
        // counters are per pass: the rows are reused by every pass of a sort
        p.stats[(size_t)stat_row__ * kStatsStride + 7] = 0;
        p.stats[(size_t)stat_row__ * kStatsStride + 8] = 0;
        p.stats[(size_t)stat_row__ * kStatsStride + 10] = 0;
    }
    LSD_STAMP(0);   // ticket
#endif

    const uint32_t remaining = range_end - tile_base;
    const uint32_t valid = remaining < (uint32_t)TILE ? remaining : (uint32_t)TILE;
    const bool full = valid == (uint32_t)TILE;

    // ---- 1. load: wave-striped, so (register row, lane) order == key order ------------------------
    uint32_t key[K];
    uint32_t rank[K];
    bool ranked = false;   // wave-uniform: the full-tile fast path below has already ranked this wave's keys
    const uint32_t first = tile_base + wave * (uint32_t)(kWave * K) + lane;
    // one 64-bit base, constant offsets: an index sum per load would cost an address pair per load
    const uint32_t* const keys_in = in_keys + first;
    if (full) {
#pragma unroll
        for (int i = 0; i < K; i++) key[i] = keys_in[i * kWave];
#ifndef LSD_NO_RANK_HALVES   // (-DLSD_NO_RANK_HALVES builds the form that waits for every load first: +1.3 % kernel time)
        // Full tile, returning-LDS-add rank, no key transform: rank the first half of the rows as soon as THEIR loads are
        // in (the loads retire in issue order) and the second half behind a second wait, in the basic block of the loads --
        // behind the join with the tail-tile path the compiler can only wait for everything (vmcnt(0)).
        if constexpr (RANK == kRankLdsAdd && K >= 16 && !XF) {
#ifndef LSD_RANK_BATCHES
#define LSD_RANK_BATCHES 2
#endif
            constexpr int NB = LSD_RANK_BATCHES, PER = K / NB;
            static_assert(K % NB == 0, "batches divide the rows");
            asm volatile("s_waitcnt vmcnt(%0)" : : "n"(K - PER) : "memory");
            if (!row_is_heavy(digit_of(key[0]))) {   // heavy digits take the path below (phase 2)
#pragma unroll
                for (int b = 0; b < NB; b++) {
                    if (b > 0) {
                        // batch b's keys become usable only behind its own wait: without the pins their digit extraction is
                        // hoisted above the earlier batches' atomics, and the wait for them with it
                        if (b == 1) asm volatile("s_waitcnt vmcnt(%0)" : : "n"(K - 2 * PER < 0 ? 0 : K - 2 * PER) : "memory");
                        else if (b == 2) asm volatile("s_waitcnt vmcnt(%0)" : : "n"(K - 3 * PER < 0 ? 0 : K - 3 * PER) : "memory");
                        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
                        for (int i = b * PER; i < (b + 1) * PER; i++) asm volatile("" : "+v"(key[i]));
                    }
#pragma unroll
                    for (int i = b * PER; i < (b + 1) * PER; i++)
                        rank[i] = __hip_atomic_fetch_add((lds_u32*)&s_cnt[wave * H + digit_of(key[i])], 1u, __ATOMIC_RELAXED,
                                                         __HIP_MEMORY_SCOPE_WAVEFRONT);
                }
                ranked = true;
            }
        }
#endif
    } else {
#pragma unroll
        for (int i = 0; i < K; i++) {
            const uint32_t idx = first + i * kWave;
            key[i] = idx < range_end ? keys_in[i * kWave] : 0xFFFFFFFFu;
        }
    }

    if constexpr (XF) {
        if (p.xin.on) {   // first pass of a typed sort: keys become "sortable" uint32 (padding stays the maximum)
#pragma unroll
            for (int i = 0; i < K; i++)
                if (full || first + i * kWave < range_end) key[i] = to_sortable(key[i], p.xin);
        }
    }

    // ---- 2. intra-wave stable rank ----------------------------------------------------------
#ifdef LSD_PHASE_STATS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    LSD_STAMP(1);   // key load
#endif
    if (ranked) {
        // done above
    } else if (RANK == kRankLdsAdd) {
        // The returned old value is (same-digit keys in earlier rows) + (peers in lower lanes):
        // the K atomics are independent, so they issue back to back.
        // HEAVY digits need care: LDS atomics of one wave instruction that meet on ONE word are served a lane per clock
        // (tools/ceiling/lds_atomic.hip: 63 clocks when the sixteen lanes of every 16-lane group share a word, against
        // 7.9 for random words), so a digit value that a quarter, half or all of the keys carry (zeros, a default value,
        // constant or sorted input, dead digits) would slow the whole phase eightfold.  Only waves whose FIRST row shows
        // such a value pay for the careful form: up to two heavy values h1, h2 are taken from that row, their keys are
        // ranked from running counts kept in scalar registers (count so far + lower lanes with the same value: no LDS
        // operation at all), everybody else still takes an atomic, and the counts are written to the wave's table at the
        // end (no atomic ever touches those two words: a key either has the value or it has not).  Uniform random input
        // takes the straight path, where the K atomics issue back to back.
        uint32_t h1 = 0, h2 = 0;
        if (pick_heavy(digit_of(key[0]), digit_of(key[K - 1]), h1, h2)) {
            // first every key that holds neither value takes its returning add (nothing else writes rank[] in this loop, so
            // the adds issue one after the other and nothing waits for them) ...
#pragma unroll
            for (int i = 0; i < K; i++) {
                const uint32_t d = digit_of(key[i]);
                if (d != h1 && d != h2) {
                    rank[i] = __hip_atomic_fetch_add((lds_u32*)&s_cnt[wave * H + d], 1u, __ATOMIC_RELAXED,
                                                     __HIP_MEMORY_SCOPE_WAVEFRONT);
                }
            }
            // ... then the holders are ranked from the running counts: vector and scalar ALU only
            uint32_t c1 = 0, c2 = 0;
            // (ONE loop for one or two candidates: h2 == kNoDigit matches no key.  A second, single-candidate copy of this
            // unrolled loop saved the second ballot's six instructions per row on such waves, and cost every wave -- the
            // plain path included -- 18 registers and a scratch slot: measured round 3, 110 -> 128 VGPRs + 8 B of scratch.)
#pragma unroll
            for (int i = 0; i < K; i++) {
                const uint32_t d = digit_of(key[i]);
                const bool in1 = d == h1, in2 = d == h2;
                const uint64_t m1 = __ballot(in1), m2 = __ballot(in2);
                const uint32_t r1 = mbcnt_add(m1, c1), r2 = mbcnt_add(m2, c2);
                rank[i] = in1 ? r1 : (in2 ? r2 : rank[i]);
                c1 = popc64_add(m1, c1);
                c2 = popc64_add(m2, c2);
            }
            if (lane == 0) {
                s_cnt[wave * H + h1] = c1;
                if (h2 != kNoDigit) s_cnt[wave * H + h2] = c2;
            }
        } else {
#pragma unroll
            for (int i = 0; i < K; i++) {
                const uint32_t d = digit_of(key[i]);
                rank[i] = __hip_atomic_fetch_add((lds_u32*)&s_cnt[wave * H + d], 1u, __ATOMIC_RELAXED,
                                                 __HIP_MEMORY_SCOPE_WAVEFRONT);
            }
        }
    } else {
        const uint64_t lane_bit = 1ull << lane;
#pragma unroll
        for (int i = 0; i < K; i++) {
            const uint32_t d = digit_of(key[i]);
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
    // digit totals that other tiles may see: the tail's padding is not data.  The padding carries whatever digit 0xFFFFFFFF
    // has: the highest, H - 1, for a bit field -- but under the splitter partition its bucket is the number of LIVE splitters,
    // which may be lower (thresholds above every key are not compared); no real key has a higher digit either way.
    uint32_t pub_total = total;
    if (tid == digit_of(0xFFFFFFFFu)) pub_total -= (uint32_t)TILE - valid;

    const uint32_t parity = p.parity;
    const uint32_t c_stale = code_stale(parity);
    const uint32_t c_prefix = code_prefix(parity);
    // this thread's (digit, slot) column of its region's chain
    const uint32_t my_digit = tid % (uint32_t)H;
    const uint32_t my_slot = tid / (uint32_t)H;
    const uint32_t* const status_col = CHAINED ? p.status + (size_t)chain_row0 * H + my_digit : nullptr;
    uint32_t window[LB] = {};
    int32_t j = (int32_t)chain_pos - 1;   // nearest predecessor in the chain not yet consumed
    uint32_t region_base = 0;   // where this region's keys of digit `tid` start in the output
    auto first_step = [&]() {
        // fetched here rather than where it is used, after the walk: one dependent round trip less
        if (CHAINED && tid < (uint32_t)H) region_base = p.regions[kRegionHeaderWords + region * H + tid];
        if (CHAINED && my_slot < (uint32_t)LSLOTS) {
            // first look-back step, consumed after the LDS writes of round 0: slot s covers
            // predecessors j - s*LB - l, so LSLOTS*LB status rows cost one round trip
            const int32_t j0 = j - (int32_t)(my_slot * LB);
#pragma unroll
            for (int l = 0; l < LB; l++) window[l] = (j0 - l >= 0) ? load_status(status_col + (size_t)(j0 - l) * H) : c_stale;
        }
    };
#ifdef LSD_FAULT_INJECT
    // Diagnostic build only (make faultinject): the tile in status row mute_row - 1 never publishes, so its
    // successors exercise the bounded-spin expiry below (tests/test_fault_path.py).
    const bool muted = CHAINED && p.mute_row != 0 && tile + 1u == p.mute_row;
#else
    constexpr bool muted = false;
#endif
    if (CHAINED && tid < (uint32_t)H && !muted) {
        // publish as early as possible: successors can already add this tile's counts (behind the loads
        // above in issue order, so that their consumer need not outwait this store's acknowledgement)
        const uint32_t code = chain_pos == 0 ? c_prefix : code_aggregate(parity);
        store_status(p.status + (size_t)tile * H + tid, (pub_total << 2) | code);
    }

    uint32_t incl = wave_inclusive_scan(tid < (uint32_t)H ? total : 0u, lane);
    if (H > kWave) {
        if (lane == 63u) s_misc[1 + wave] = incl;
        lds_barrier();
        // all wave totals are read as one batch and selected afterwards: a read per `if` is a basic block of its own
        // with its own lgkmcnt(0), i.e. H / 64 dependent LDS round trips on the tile's critical path
        uint32_t part[(H / kWave) > 0 ? H / kWave : 1];
#pragma unroll
        for (int w = 0; w < H / kWave; w++) part[w] = s_misc[1 + w];
        uint32_t carry = 0;
#pragma unroll
        for (int w = 0; w < H / kWave; w++) carry += (uint32_t)w < wave ? part[w] : 0u;
        incl += carry;
    }
    const uint32_t local_off = incl - total;   // exclusive scan over digits
    if (tid < (uint32_t)H) {
#pragma unroll
        for (int w = 0; w < W; w++) s_cnt[w * H + tid] = local_off + wave_excl[w];
    }
    lds_barrier();
    LSD_STAMP(3);   // totals, publish, scan, bases

    // Payload loads.  With a single reorder round they are issued only after the keys have gone to
    // LDS (below), so a payload never shares the register file with a live key: 2 registers per
    // pair instead of 3.  With several rounds keys stay live across rounds, so the payloads are
    // fetched here and land while the keys are reordered.
    uint32_t val[PAIRS ? K : 1];
    auto load_vals = [&](uint32_t e = 0) {
        const uint32_t* const vals_in = payload_in(e) + first;
        if (full) {
#pragma unroll
            for (int i = 0; i < K; i++) val[i] = vals_in[i * kWave];
        } else {
#pragma unroll
            for (int i = 0; i < K; i++) {
                const uint32_t idx = first + i * kWave;
                val[i] = idx < range_end ? vals_in[i * kWave] : 0u;
            }
        }
    };
    if (PAIRS && ROUNDS > 1) load_vals();

    // ---- 5. reorder through LDS in ROUNDS rounds of CAP tile positions each ------------------
    // A key whose tile-sorted position is q belongs to round q / CAP, slot q % CAP.  Rounds are
    // by position, not by digit, so their size never depends on the key distribution; the tile
    // (and with it the length of every digit's run in global memory) can exceed the LDS buffer.
    // key/value tiles of up to 64 Ki positions keep them two to a register (PACKED): they have to
    // survive until the payloads have gone through LDS as well
    constexpr bool PACKED = PAIRS && ROUNDS == 1 && TILE <= 65536 && K % 2 == 0;
    uint32_t pos[PACKED ? K / 2 : K];
#pragma unroll
    for (int i = 0; i < K; i++) {
        // recompute the digit here: carrying K digits (or LDS addresses) over from the rank phase
        // would cost K registers
        uint32_t kk = key[i];
        asm volatile("" : "+v"(kk));
        const uint32_t d = digit_of(kk);
        const uint32_t q = s_cnt[wave * H + d] + rank[i];
        if (!PACKED) pos[i] = q;
        else if (i % 2 == 0) pos[i / 2] = q;
        else pos[i / 2] |= q << 16;
    }
    auto pos_at = [&](int i) -> uint32_t {
        if (!PACKED) return pos[i];
        // opaque to the optimiser: otherwise it keeps (and spills) the K LDS addresses of the key
        // round for the payload round instead of unpacking again
        uint32_t w = pos[i / 2];
        asm volatile("" : "+v"(w));
        return i % 2 == 0 ? (w & 0xFFFFu) : (w >> 16);
    };

    // Issued here, not right after publishing: what matters is how fresh the rows are when they are
    // consumed (a row read early shows counts where a prefix would be by now, and the walk goes on
    // past it); the LDS writes below are cover enough.  A/B on 2^28 keys: 2.29-2.35 -> 2.22-2.23 ms.
    first_step();
#pragma unroll
    for (int round = 0; round < ROUNDS; round++) {
        if (round > 0) lds_barrier();   // the previous round has been read back
#pragma unroll
        for (int i = 0; i < K; i++) {
            if (ROUNDS == 1) s_keys[pos_at(i)] = key[i];      // one round: every position is below CAP
            else if ((pos_at(i) / (uint32_t)CAP) == (uint32_t)round) s_keys[pos_at(i) % (uint32_t)CAP] = key[i];
        }
        if (PAIRS && ROUNDS == 1) load_vals();   // the key registers are free now
        // One workgroup per CU (the 32768-key tile), keys only: nothing else runs on the CU while this
        // tile waits for its predecessors, so fetch the tile-ordered keys back from LDS BEFORE the
        // look-back is consumed and leave only destination lookups and stores behind it (+0.7 %; with
        // two workgroups per CU the extra barrier costs more than it hides: -1.5 %).
        // Round 1 kept this on for the 32768-key tile (+0.7 % then).  Since the read-back issues its destination-base reads in
        // batches (round 2) it LOSES 0.9 % (kernel 1.9 %): the slots' LDS reads now overlap the stores anyway, and the 32
        // registers it holds across the look-back are better spent there.  -DLSD_PREREAD builds it.
#ifdef LSD_PREREAD
        constexpr bool PREREAD = !PAIRS && ROUNDS == 1 && TILE >= 32768;
#else
        constexpr bool PREREAD = false;
#endif
        uint32_t back[PREREAD ? SLOTS : 1];
        if (PREREAD) {
            lds_barrier();   // the whole tile is in LDS
#pragma unroll
            for (int s2 = 0; s2 < SLOTS; s2++) back[s2] = s_keys[s2 * T + tid];
        }

        if (round == 0) {
            LSD_STAMP(4);   // first round's LDS writes
            // ---- 4. tile base per digit ("global offsets", .cu:885-894) ----------------------------
            if (CHAINED && LSLOTS > 1) {
                // the helper slots hand their share of the first step to the digit's owner
                if (my_slot >= 1 && my_slot < (uint32_t)LSLOTS) {
#pragma unroll
                    for (int l = 0; l < LB; l++) s_look[((my_slot - 1) * LB + l) * H + my_digit] = window[l];
                }
                lds_barrier();
            }
            if (CHAINED) {
                // region_base and the first look-back step were loaded before the LDS writes: take them HERE, by every
                // thread, ahead of the branch.  A wave's memory operations retire in issue order, so wherever the
                // compiler places its wait for them it waits for everything older too: left to itself it waited at
                // the join BEHIND the digit threads' block (the registers are reused there), and so sat out the
                // acknowledgement of the prefix store inside it before the wave could start on its key stores.
                asm volatile("" : "+v"(region_base));
#pragma unroll
                for (int l = 0; l < LB; l++) asm volatile("" : "+v"(window[l]));
            }
            if (tid < (uint32_t)H) {
                uint32_t gbase;
                uint64_t run_end;   // one past the last destination of this tile's keys of digit `tid`
                if (CHAINED) {
                    uint32_t excl = 0;
                    if (chain_pos > 0) {
                        // first step: own window, then the helper slots' rows, in chain order
                        int consumed = 0;
                        bool found = false;
                        if (tid == 0) LSD_SET(14, __builtin_amdgcn_s_memrealtime());
#pragma unroll
                        for (int m = 0; m < LSLOTS * LB; m++) {
                            const uint32_t word = m < LB ? window[m] : s_look[(m - LB) * H + tid];
                            const uint32_t code = word & 3u;
                            if (!found && consumed == m && code != c_stale) {
                                excl += word >> 2;
                                consumed = m + 1;
                                found = (code == c_prefix) || (j - m == 0);
                            }
                        }
                        j -= consumed;
                        if (tid == 0) {
                            LSD_COUNT(10, consumed);
                            LSD_SET(15, __builtin_amdgcn_s_memrealtime());   // first (prefetched) step consumed
                        }
                        uint32_t spins = 0;
                        bool gave_up = false;
                        while (!found) {
                            // further steps: LB rows at a time by the owner alone
#pragma unroll
                            for (int l = 0; l < LB; l++)
                                window[l] = (j - l >= 0) ? load_status(status_col + (size_t)(j - l) * H) : c_stale;
                            consumed = 0;
#pragma unroll
                            for (int l = 0; l < LB; l++) {
                                const uint32_t code = window[l] & 3u;
                                if (!found && consumed == l && code != c_stale) {
                                    excl += window[l] >> 2;
                                    consumed = l + 1;
                                    found = (code == c_prefix) || (j - l == 0);
                                }
                            }
                            if (tid == 0) LSD_COUNT(7, 1);
                            if (consumed == 0) {
                                // Bounded wait.  On expiry: raise the fault word and give the tile up -- no prefix is
                                // published from the partial sum and nothing of this tile is stored (below).  Every
                                // waiter also looks at the fault word now and then, so once one tile has given up the
                                // tiles behind it drain at once instead of each sitting out its own limit.
                                if (++spins > p.spin_limit || ((spins & 255u) == 0u && load_status(p.fault) != 0u)) {
                                    atomicOr(p.fault, 1u);
                                    gave_up = true;
                                    break;
                                }
                                __builtin_amdgcn_s_sleep(1);
                                if (tid == 0) LSD_COUNT(8, 1);
                            }
                            j -= consumed;
                            if (tid == 0) LSD_COUNT(10, consumed);
                        }
                        if (tid == 0) {
                            LSD_SET(12, __builtin_amdgcn_s_memrealtime());
                            LSD_SET(13, j + 1);
                        }
                        if (gave_up) s_misc[30] = 1u;
                        else if (!muted) store_status(p.status + (size_t)tile * H + tid, ((excl + pub_total) << 2) | c_prefix);
                        if (tid == 0) LSD_SET(11, __builtin_amdgcn_s_memrealtime());
                    }
                    gbase = region_base + excl;
                    run_end = (uint64_t)region_base + excl + pub_total;
                } else {
                    gbase = p.global_off[(size_t)tile * H + tid];
                    run_end = (uint64_t)gbase + pub_total;
                }
                // Destination guard.  Every store below goes to out[gbase + i], i < this tile's count of the digit, so
                // run_end <= n keeps all of them inside the output whatever the tables hold.  The tables come from
                // counts taken in another kernel (stage 1 / the caller's global_off): counts that do not describe the
                // keys -- a miscounting histogram variant (the dist8.log fault), a caller's wrong
                // table -- must end in LSDSORT_ERR_DEVICE_FAULT, never in a store outside the buffer.  The tile stores
                // nothing; its prefix is already published, so nobody behind it waits.
                if (run_end > (uint64_t)p.n) {
                    if (p.fault) atomicOr(p.fault, 2u);
                    s_misc[30] = 1u;
                }
                s_gdelta[tid] = gbase - local_off;
            }
        }
        lds_barrier();
        if (round == 0) LSD_STAMP(5);   // look-back (wave 0's digits) + barrier
        if (round == 0 && s_misc[30] != 0u) {   // the look-back gave up or the destination guard fired (uniform): the sort has
            clear_next();                       // failed (fault word set); store nothing from a base that is not known
            return;
        }

        // linear read-back: consecutive threads hold consecutive tile positions, so each digit's
        // keys leave as one contiguous run
        // key/value: remember each slot's digit (one byte, four to a register) so that the payload's
        // destination can be rebuilt without keeping SLOTS addresses alive
        uint32_t dbytes[PAIRS ? (SLOTS + 3) / 4 : 1];
        if (PAIRS) {
            // The payload loads have had the whole look-back to arrive: take them NOW, before the key stores
            // are issued.  A wave's memory operations retire in issue order, so a wait for the payloads placed
            // behind those stores (where the payloads are used) would also wait for every store's acknowledgement.
#pragma unroll
            for (int i = 0; i < K; i++) asm volatile("" : "+v"(val[i]));
            // `full` (every tile but a region's last) is tested ONCE, outside the slot loops: a test per slot puts every slot
            // into a basic block of its own, and the s_gdelta read of each is then waited for (lgkmcnt(0)) before the next
            // is issued -- 32 dependent LDS round trips per thread instead of one batch.
            auto key_slots = [&](auto all_valid) {
#pragma unroll
                for (int s2 = 0; s2 < SLOTS; s2++) {
                    const uint32_t slot = s2 * T + tid;
                    const uint32_t q = round * CAP + slot;
                    const uint32_t k = s_keys[slot];
                    const uint32_t d = digit_of(k);
                    if ((s2 & 3) == 0) dbytes[s2 / 4] = d;
                    else dbytes[s2 / 4] |= d << (8 * (s2 & 3));
                    if (decltype(all_valid)::value || q < valid) out_keys[s_gdelta[d] + q] = leaving(k);
                    if ((s2 & 7) == 7) __builtin_amdgcn_sched_barrier(0);   // keep at most eight slots in flight
                }
            };
            if (full) key_slots(std::true_type{});
            else key_slots(std::false_type{});
        } else if (PREREAD) {
            auto key_slots = [&](auto all_valid) {
#pragma unroll
                for (int s2 = 0; s2 < SLOTS; s2++) {
                    const uint32_t q = s2 * T + tid;
                    const uint32_t k = back[PREREAD ? s2 : 0];
                    const uint32_t d = digit_of(k);
                    if (decltype(all_valid)::value || q < valid) out_keys[s_gdelta[d] + q] = leaving(k);
                }
            };
            if (full) key_slots(std::true_type{});
            else key_slots(std::false_type{});
        } else {
            // keys only: sixteen slots at a time, which bounds the registers of the read-back
            // slots per batch.  One 16-wave workgroup per CU (1024 threads): 2 / 4 / 8 / 16 / 32 slots measure 2.006 / 2.011-2.016 /
            // 2.032 / 2.035 / 2.037 ms per 2^28-key sort and 1 slot 2.082 (small batches let a batch's LDS reads overlap the
            // stores of the one before; a single slot is two dependent LDS round trips with every wave of the CU in the same
            // phase).  With two 8-wave workgroups per CU (512 threads) the OTHER workgroup is what overlaps, and one slot at
            // a time is fastest (table-driven pass, 512 x 32 tile: 1 / 2 / 4 / 16 slots 0.430 / 0.437 / 0.456 / 0.467 ms).
#ifdef LSD_READBACK_STEP
            constexpr int STEP_WANTED = LSD_READBACK_STEP;
#else
            constexpr int STEP_WANTED = T >= 1024 ? 4 : 1;
#endif
            constexpr int STEP = SLOTS < STEP_WANTED ? SLOTS : STEP_WANTED;
            auto key_slots = [&](auto all_valid) {
#pragma unroll 1
                for (int s0 = 0; s0 < SLOTS; s0 += STEP) {
#pragma unroll
                    for (int u = 0; u < STEP; u++) {
                        const uint32_t slot = (s0 + u) * T + tid;
                        const uint32_t q = round * CAP + slot;
                        const uint32_t k = s_keys[slot];
                        const uint32_t d = digit_of(k);
                        if (decltype(all_valid)::value || q < valid) out_keys[s_gdelta[d] + q] = leaving(k);
                    }
                }
            };
            if (full) key_slots(std::true_type{});
            else key_slots(std::false_type{});
        }

        // ---- 6. payloads follow their keys through the same slots -----------------------------
        if (PAIRS) {
            // one payload array after the other (records carry up to three: wide.hip); with a single round the next array's
            // loads are issued as soon as this one's registers have gone to LDS, so they land while it is stored
            for (uint32_t e = 0; e < num_payloads; e++) {
                lds_barrier();   // every key (or payload e - 1) of this round has been read back
#pragma unroll
                for (int i = 0; i < K; i++) {
                    if (ROUNDS == 1) s_keys[pos_at(i)] = val[i];
                    else if ((pos_at(i) / (uint32_t)CAP) == (uint32_t)round) s_keys[pos_at(i) % (uint32_t)CAP] = val[i];
                }
                if (ROUNDS == 1 && e + 1 < num_payloads) load_vals(e + 1);
                lds_barrier();
                uint32_t* const vals_to = payload_out(e);
                auto val_slots = [&](auto all_valid) {
#pragma unroll
                    for (int s2 = 0; s2 < SLOTS; s2++) {
                        const uint32_t slot = s2 * T + tid;
                        const uint32_t q = round * CAP + slot;
                        const uint32_t d = (dbytes[s2 / 4] >> (8 * (s2 & 3))) & 0xFFu;
                        if (decltype(all_valid)::value || q < valid) vals_to[s_gdelta[d] + q] = s_keys[slot];
                        if ((s2 & 7) == 7) __builtin_amdgcn_sched_barrier(0);
                    }
                };
                if (full) val_slots(std::true_type{});
                else val_slots(std::false_type{});
            }
        }
    }
    // Housekeeping for the next pass goes LAST: nothing is waited for behind it.  (Placed in front of the read-back, as
    // in round 1, the compiler's vmcnt(0) at the join behind its store loop made the four waves that hold the digit threads
    // sit out these stores' acknowledgements before their first key store.)
    clear_next();
#ifdef LSD_PHASE_STATS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    LSD_STAMP(6);   // read-back + stores drained
#endif
}

// Launch one instantiation.  LDS above 64 KiB needs the attribute raised once per function.
template <int R, int T, int K, int CAP, int RANK, bool PAIRS, bool CHAINED, bool XF = false>
hipError_t launch_rank_scatter_inst(const PassParams& p, hipStream_t stream)
{
    constexpr size_t lds_bytes = (size_t)rank_scatter_lds_words<R, T, K, CAP, RANK>() * sizeof(uint32_t);
    auto kernel = rank_scatter_kernel<R, T, K, CAP, RANK, PAIRS, CHAINED, XF>;
    if (lds_bytes > 64 * 1024) {
        static std::atomic<uint64_t> told{0};
        const hipError_t attr = allow_dynamic_lds(reinterpret_cast<const void*>(kernel), lds_bytes, told);
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
    const uint32_t grid = p.num_tiles;
    if (t_launch_start && t_launch_stop) {
        // timed sorts: the events take the kernel's own begin and end (no marker packets between the
        // passes, whose queue bubbles would be counted as kernel time)
        hipExtLaunchKernelGGL(kernel, dim3(grid), dim3(T), (uint32_t)lds_bytes, stream, t_launch_start, t_launch_stop, 0u, p);
    } else {
        hipLaunchKernelGGL(kernel, dim3(grid), dim3(T), lds_bytes, stream, p);
    }
    return hipGetLastError();
}

template <int R, int T, int K, int CAP, int RANK>
hipError_t launch_rank_scatter_rank(bool chained, const PassParams& p, hipStream_t stream)
{
    const bool pairs = p.vals_in != nullptr;
    if (p.xin.on || p.xout.on) {
        // typed sorts (int32 / float32 / descending): chained form, 4- and 8-bit digits
        if constexpr (R >= 4) {
            if (chained)
                return pairs ? launch_rank_scatter_inst<R, T, K, CAP, RANK, true, true, true>(p, stream)
                             : launch_rank_scatter_inst<R, T, K, CAP, RANK, false, true, true>(p, stream);
        }
        return hipErrorInvalidValue;
    }
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
