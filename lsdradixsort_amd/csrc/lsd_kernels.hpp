// lsd_kernels.hpp -- host-visible launch interface of the gfx950 kernels.
//
// The C-ABI layer (lsdsort_api.hip) sequences these; each launcher picks the template
// instantiation for (radix_bits, tile shape) and returns the hipError_t of the launch.
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <stddef.h>
#include <stdint.h>

namespace lsd {

// Threads per workgroup and keys per thread of the rank-and-scatter / tile-histogram
// kernels; tile = threads * keys_per_thread keys (the reference's `block`, .cu:919).
struct TileShape {
    int threads;
    int keys_per_thread;
    int tile() const { return threads * keys_per_thread; }
};

// Compiled tile shapes for a radix; index 0 is the default.  Returns the count.
int tile_shapes(int radix_bits, const TileShape** out);
// Does shape `id` reorder its whole tile in ONE LDS round (CAP == tile)?  Further payload arrays need that.
bool single_round_shape(int radix_bits, int id);

// Regions.  A pass's input is split into contiguous regions whose digit histograms are known
// before the pass starts, so each region carries its own chained scan over its own tiles and no
// tile ever needs anything from another region.  Each XCD has its own share of the regions:
// workgroups serve those first (hardware XCC_ID), which keeps neighbouring runs in one L2; and the
// more chains there are, the fewer tiles are in flight on each, which is what the look-back's
// walk is proportional to (measured: 64 tiles in flight per chain 7-8 rows and 4.4 us, 32 tiles 3.7
// rows and 2.2 us).
//   pass 0     : region x = positions [x*R0, (x+1)*R0), R0 a multiple of the tile size
//   pass p >= 1: region x = keys whose digit p-1 has top log2(regions) bits x -- contiguous in
//                the array because pass p-1 just sorted on that digit; its histogram of digit p
//                is a joint count of two key fields, so it is permutation-invariant and comes
//                out of the same single upfront read as the plain digit histograms.
// 8-bit digits: 8 regions (top three bits of the previous digit, one per XCD).  16 regions (64 KiB of
// LDS counters per workgroup in the upfront histogram, which is as fast as with 8: it is bound by the
// number of LDS atomics, not by the table) sort uniform keys 1 % faster -- whole-sort A/B at 2^28 keys:
// 8 regions 2.181 ms, 16 regions 2.160 ms, 32 regions 2.205 ms -- but lose 10 % on sorted, reversed and
// few-valued inputs (104 -> 93 Gkeys/s sorted, 95 -> 84 with 16 values per digit), so 8 stays
// (-DLSD_R8_REGION_BITS=4 builds the other).  4-bit digits: 16 regions (the whole previous digit).
// Narrower digits and the multi-GPU partition: one region, i.e. one chain.
constexpr int kMaxRegions = 32;
constexpr int kXcds = 8;
#ifndef LSD_R8_REGION_BITS
#define LSD_R8_REGION_BITS 3
#endif
inline constexpr int region_bits_for_radix(int radix_bits) { return radix_bits == 8 ? LSD_R8_REGION_BITS : (radix_bits == 4 ? 4 : 0); }
inline constexpr int regions_for_radix(int radix_bits) { return 1 << region_bits_for_radix(radix_bits); }
// Per-pass region table, uint32 words: start[32] | len[32] | tiles[32] | tile_off[32] | base[regions][2^R]
constexpr int kRegionHeaderWords = 4 * kMaxRegions;
inline constexpr size_t region_table_words(int radix_bits) { return kRegionHeaderWords + (size_t)kMaxRegions * ((size_t)1 << radix_bits); }

// Key order other than ascending uint32: a bijection to "sortable" uint32 is applied
// where the keys enter (upfront histogram, first pass's load) and undone where they leave (last pass's
// store); every pass in between sees plain uint32.
//   to sortable : t = k ^ (sext(k & a) | b) ^ c          from sortable : k = u ^ (sext(~u & a) | b),  u = t ^ c
//   uint32: a = b = 0     int32: a = 0, b = 0x80000000     float32 (IEEE bit order): a = b = 0x80000000
//   descending: c = 0xFFFFFFFF (ascending on the complement; stable, ties keep their input order)
struct KeyTransform {
    uint32_t on;   // 0: identity (a, b, c ignored)
    uint32_t a, b, c;
};
__host__ __device__ inline uint32_t to_sortable(uint32_t k, const KeyTransform& x)
{
    return k ^ ((uint32_t)((int32_t)(k & x.a) >> 31) | x.b) ^ x.c;
}
__host__ __device__ inline uint32_t from_sortable(uint32_t t, const KeyTransform& x)
{
    const uint32_t u = t ^ x.c;
    return u ^ ((uint32_t)((int32_t)(~u & x.a) >> 31) | x.b);
}

// Kernels that ask for more than 64 KiB of dynamic LDS say so once PER DEVICE (a process may drive several: lsdsort_u32_ex with
// num_gpus > 1 runs one host thread per device): `done` is the call site's own bit set of devices already told.
inline hipError_t allow_dynamic_lds(const void* kernel, size_t bytes, std::atomic<uint64_t>& done)
{
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const uint64_t bit = 1ull << (dev & 63);
    if (done.load(std::memory_order_acquire) & bit) return hipSuccess;
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess) done.fetch_or(bit, std::memory_order_release);
    return e;
}

// Set (per host thread) around a rank-and-scatter launch by lsdsort_u32_device_timed: events that
// receive the kernel's own begin and end timestamps (hipExtLaunchKernelGGL).
inline thread_local hipEvent_t t_launch_start = nullptr;
inline thread_local hipEvent_t t_launch_stop = nullptr;

// Bounded spin of the chained scan's look-back: ~2^22 empty polls with a sleep in each is seconds of wall
// time, far beyond any legitimate wait.  On expiry the tile raises the workspace's fault word and gives
// up -- it publishes no prefix and stores nothing -- and every other waiter, seeing the fault word, does
// the same, so the grid always drains (lsdsort_check_device reports LSDSORT_ERR_DEVICE_FAULT).
constexpr uint32_t kSpinLimit = 1u << 22;

// Everything one rank-and-scatter launch needs.
struct PassParams {
    const uint32_t* in;
    uint32_t* out;
    const uint32_t* vals_in;   // null: keys only
    uint32_t* vals_out;
    // Further payload arrays (records: a 64-bit payload's second word, the other word of a 64-bit key -- wide.hip): the key/value
    // kernel sends each through the same LDS slots and destinations as the first, one after the other; num_payloads = 1 + the
    // number of these that are set (0 or 1 = the ordinary keys / pairs launch).  Single-round tile shapes only.
    const uint32_t* more_in[2];
    uint32_t* more_out[2];
    uint32_t num_payloads;
    // Chained form only, may be null.  plan[0] != 0: this pass's digit is the same for every key -- the pass is the
    // identity and is skipped (its workgroups only clear the next pass's status rows).  plan[1] != 0: an odd number of
    // passes before this one really ran, so the keys are in `out` and go to `in` (the roles swap).  Written by stage 2
    // (scan_regions_kernel) from the digit counts; with it every pass is launched with the SAME in / out.
    const uint32_t* plan;
    // plan[0] is looked at FIRST, and a pass it marks 2 ("the other form of the sort runs") returns before anything else: set for
    // the four-pass form's launches of a sort that tried the hybrid form, which otherwise cost 16 us each to say no (8200
    // workgroups that set up their LDS, take the barrier and leave).  The wait costs a pass that does run about 1 %.
    uint32_t plan_first;
    uint32_t n;
    uint32_t shift;            // bit_group * radix_bits
    const uint32_t* shift_word;  // not null: the shift is read from here instead (the hybrid form's passes: planned on the device)
    uint32_t num_tiles;        // grid size: chained = upper bound on the regions' tile counts
    // chained (onesweep) form
    const uint32_t* regions;     // this pass's region table (device, written by the scan kernel)
    uint32_t* status;            // [num_tiles][2^R] tile-status words (lsd_device.hpp)
    uint32_t* status_clear;      // the next pass's status array: workgroup b zeroes row b; may be null
    uint32_t* tickets;           // [kMaxRegions] arrival ticket dispensers for this pass (zeroed)
    uint32_t parity;             // pass parity for the status codes
    // staged form
    const uint32_t* global_off;  // [num_tiles][2^R] digit-major exclusive scan, block-major
    uint32_t* fault;             // workspace fault word
    uint32_t spin_limit;         // empty look-back polls a tile sits out before it gives up (lsd_device.hpp kSpinLimit)
    uint32_t mute_row;           // diagnostic builds only (LSD_FAULT_INJECT): status row + 1 that never publishes; 0 = none
    uint32_t xcd_chunk;          // staged form: consecutive tiles kept on one XCD (0 = no affinity)
    unsigned long long* stats;   // diagnostic builds only (LSD_PHASE_STATS); null otherwise
    // Splitter partition (narrow-digit kernels only, multi-GPU step 1 for skewed keys): when
    // num_splitters != 0 it is 2^R - 1 and the "digit" of a key is its bucket, the number of
    // splitters <= key (ascending splitters); shift is ignored.  Only the first live_splitters are
    // compared: the rest stand for thresholds above every key (a splitter (0xFFFFFFFF, rank q) seen from
    // a rank below q, sharded.hip), which no 32-bit value can express.
    uint32_t num_splitters;
    uint32_t live_splitters;
    uint32_t splitters[7];
    KeyTransform xin;    // first pass: applied to every key as it is loaded
    KeyTransform xout;   // last pass: undone on every key as it is stored
};

// How stage 3 ranks a key among the same-digit keys of its wave (rank_scatter.hpp).
enum RankMethod : int {
    kRankBallot = 0,   // R wave-wide ballots                      (any hardware)
    kRankLdsOr = 1,    // peer mask through a wave-private LDS OR   (any hardware)
    kRankLdsAdd = 2,   // one returning LDS add per key; needs lane-ordered LDS atomics (probed)
};

// Stage 3.  chained=true: tile bases by decoupled look-back; false: read from global_off.
// rank_method: kRankLdsAdd, or any other value for the mask form suited to the digit width.
hipError_t launch_rank_scatter(int radix_bits, const TileShape& shape, int rank_method, bool chained,
                               const PassParams& p, hipStream_t stream);

// Device probe of the property kRankLdsAdd relies on: returns hipSuccess and sets *ok.
hipError_t probe_lds_add_lane_order(bool* ok, hipStream_t stream);

// Stage 1, onesweep: all `groups` digit histograms (digit g at bit shift0 + g*radix_bits) in
// one read; hist[g][d] must be zero on entry.
// counts[b] += keys whose bucket (number of the first `live` ascending splitters <= key) is b; 2^bits buckets, bits <= 3
hipError_t launch_bucket_histogram(int bits, const uint32_t* splitters_host, int live, const uint32_t* keys, uint32_t n,
                                   uint32_t* hist, hipStream_t stream);
hipError_t launch_digit_histograms(int radix_bits, int groups, uint32_t shift0, const uint32_t* keys,
                                   uint32_t n, uint32_t* hist, hipStream_t stream);

// Stage 2, onesweep: base[g][d] = exclusive scan over d of hist[g][d], for every group.
hipError_t launch_scan_digit_counts(int radix_bits, int groups, const uint32_t* hist, uint32_t* base,
                                    hipStream_t stream);

// Stage 1, onesweep with regions (radix 4 and 8): joint[p][(digit_p << B) | region_p(key)], B =
// region_bits_for_radix, for every pass in one read; `region0_keys` is R0 (pass-0 regions are by
// position).  joint must be zero on entry.
// `keys` may be the slice [first_key, first_key + n) of the array (first_key a multiple of 4096): counts accumulate.
hipError_t launch_joint_histograms(int radix_bits, const uint32_t* keys, uint32_t n, uint32_t region0_keys,
                                   uint32_t* joint, hipStream_t stream, const KeyTransform& xform = KeyTransform{},
                                   uint32_t first_key = 0, const uint32_t* skip = nullptr);   // *skip != 0: the launch does nothing

// Stage 2, onesweep: region tables of every pass from the joint counts (`regions` = 16 or 32) or
// from plain digit histograms (`regions` = 1; passes may then be 1 for the multi-GPU partition).
// counts: [passes][2^R][regions]; tables: [passes][region_table_words(R)].
// `fault` (may be null): the workspace fault word, raised (bit 2) if a pass's counts do not sum to n.
// `hybrid_ok` (may be null): *hybrid_ok != 0 = the hybrid form runs instead: the plan then says "skip, and leave the status rows
// alone" (2) for every pass and nothing else is written.
// `skip_dead_passes` = false: the plan never marks a pass as the identity (typed sorts: their first and last pass carry the key
// transform), it only says which way round each pass runs.
hipError_t launch_scan_regions(int radix_bits, int passes, int regions, const uint32_t* counts, uint32_t n,
                               uint32_t tile_keys, uint32_t region0_keys, uint32_t* tables, hipStream_t stream,
                               uint32_t* plan = nullptr, uint32_t* fault = nullptr, const uint32_t* hybrid_ok = nullptr,
                               bool skip_dead_passes = true);
// The pass plan stage 2 writes when asked to (PassParams::plan): 2 words per pass, then plan[2 * passes] != 0 if the sorted
// keys ended up in the second buffer.  launch_finish_plan copies them (and the payloads) back in that case.
constexpr int kPlanWords = 2 * 16 + 1;   // up to 16 passes (2-bit digits)
hipError_t launch_finish_plan(const uint32_t* plan_final, uint32_t* keys, const uint32_t* alt_keys, uint32_t* vals,
                              const uint32_t* alt_vals, uint32_t n, hipStream_t stream);

// Stage 1, staged: hist[t][d] per tile (BuildHistogramsKernel, .cu:660-702).
hipError_t launch_tile_histograms(int radix_bits, const TileShape& shape, const uint32_t* keys,
                                  uint32_t n, uint32_t shift, uint32_t* hist, hipStream_t stream);

// Stage 2, staged: local[t][d] (per-tile exclusive scan) and global[t][d] (digit-major
// exclusive scan) from hist[t][d]; either output may be null.  `scratch` holds
// tile_offsets_scratch_words(tiles, radix_bits) words.
size_t tile_offsets_scratch_words(size_t tiles, int radix_bits);
hipError_t launch_tile_offsets(int radix_bits, const uint32_t* hist, uint32_t* local, uint32_t* global,
                               uint32_t tiles, uint32_t* scratch, hipStream_t stream);

// ---- the hybrid form's local stage (local_sort.hip) -------------------------------------------------------------------
// Buckets of at most kLocalSortCap keys -- bucket b = keys [bases[b], bases[b + 1]) of `keys` -- are sorted in place, one workgroup
// each, by up to three digit passes (shift, width <= 9 bits; width 0 = no pass) run from LDS to LDS.  A bucket above the
// capacity raises fault bit 3 and is left alone.  *skip != 0 (may be null): the launch does nothing.
constexpr int kLocalSortCap = 16384;
constexpr int kLocalSortCapSmall = 10240;   // the three-workgroups-per-CU variant (small_variant), keys only
constexpr int kLocalSortCapSmallPairs = 8192;   // ... with payloads (a third register per pair)
constexpr int kLocalSortCapTiny = 5120;         // small_variant 2: four workgroups per CU (keys and pairs)
struct LocalSortParams {
    uint32_t* keys;
    uint32_t* vals;              // null: keys only.  Else a payload word per key, permuted with it (stable)
    uint32_t* more[2];           // further payload arrays (num_payloads 2, 3), at most two digit passes then
    uint32_t num_payloads;       // 0 = as `vals` says (none or one)
    const uint32_t* bases;
    uint32_t num_buckets;
    uint32_t shift[4], width[4];   // the fourth: launch_small_sort only
    const uint32_t* low_bits_word; // not null: the stage sorts bits [0, *low_bits_word) in one or two passes (9 bits, then the
                                   // rest) and shift / width are ignored (the hybrid form: planned on the device)
    const uint32_t* skip;
    uint32_t* fault;
    uint32_t small_variant;   // 1: buckets of up to kLocalSortCapSmall keys, three workgroups per CU; 2: up to kLocalSortCapTiny, four
    uint32_t larger_elsewhere;   // 1: a bucket above this launch's capacity is another launch's (no fault)
    KeyTransform xout;           // typed sorts: the stage's store turns the sortable keys back (from_sortable)
    const uint32_t* list;        // null: bucket = workgroup index.  Else the buckets to sort, *list_count of them, walked by a
    const uint32_t* list_count;  // grid of 512 workgroups
};
hipError_t launch_local_sort(const LocalSortParams& p, hipStream_t stream);
// A whole sort of up to kLocalSortCap items in ONE launch: one workgroup loads keys (and vals, may be null), sorts all 32 bits
// in four 8-bit digit passes inside its LDS and stores them back.  clear0 / clear1 (may be null): words the kernel zeroes first
// (the workspace's fault word and form word: no memset launch precedes it).
hipError_t launch_small_sort(uint32_t* keys, uint32_t* vals, uint32_t n, uint32_t* clear0, uint32_t* clear1, hipStream_t stream);

// ---- the hybrid form's upfront read and planner (hybrid.hip) ------------------------------------------------------------
// A bucket = the keys that agree on their top `bucket_bits` bits: 14 while what uniform keys put into one of 2^14 buckets (mean
// + 6 sigma) fits the three-per-CU variant of the local stage (10240 keys, 8192 pairs), 15 while nearly all of 2^15 buckets do (mean + 1.5 sigma),
// 16 above (to 9.6e8 items).  Larger buckets are cheaper per key (a bucket's scans, barriers and its
// unoverlapped first load and last store are per bucket: 2^27 keys in buckets of 4096 cost the local stage 0.35-0.39 ms, in
// buckets of 8192 0.29).  Every table is sized for 2^16.
constexpr int kHybridBuckets = 1 << 16;
constexpr size_t kHybridMaxMeanBucket = 14648;   // above, the largest bucket of even uniform keys nears the 16384-key capacity
constexpr int hybrid_bucket_bits(size_t n, bool pairs)
{
    const size_t mean = n >> 14, cap = pairs ? 8192 : 10240;
    size_t root = 1;                               // ceil(sqrt(mean)), no <cmath> in device headers
    while (root * root < mean) root++;
    if (mean + 6 * root <= cap) return 14;
    // 2^15 buckets while nearly all of them still fit that variant (mean + 1.5 sigma: a few per cent go on the planner's list);
    // beyond, 2^16 buckets of half the size beat the list's one- or two-per-CU variant (400e6 keys 152 -> 160 Gkeys/s, 2^28 pairs
    // 81 -> 93 Gpairs/s; at 330e6 keys, a mean of 10071, the list of the few larger ones is still ahead: 159-162 against 153; at
    // 360e6, a mean of 10986, no longer: 146)
    const size_t mean15 = n >> 15;
    size_t root15 = 1;
    while (root15 * root15 < mean15) root15++;
    return 2 * mean15 + 3 * root15 <= 2 * cap ? 15 : 16;
}
// plan words (uint32, in the workspace's control block): written by the planner, read by every kernel of either form
constexpr int kHybridWordOk = 0;          // 1: the hybrid form runs (the ordinary form's kernels return at once)
constexpr int kHybridWordSkipLocal = 1;   // 1: the local stage returns at once (both launches)
constexpr int kHybridWordPlan = 2;        // PassParams::plan of the g-th global pass: words [2 + 2 g, 4 + 2 g), g < 4
constexpr int kHybridWordLargest = 10;    // the largest bucket (diagnostics)
constexpr int kHybridWordLargeCount = 11; // buckets above the small variant's capacity: the entries of the planner's list
constexpr int kHybridWordHopeless = 12;   // 1: a sample of the keys already shows a bucket far above the capacity: the upfront read is skipped
constexpr int kHybridWordViolated = 13;   // 1: the sampled key prefix does not hold (some key differs in its top bits): ordinary form
constexpr int kHybridWordDiffer = 14;     // OR of (key ^ first key) over the 65536 sampled keys: its leading zeros are the prefix
constexpr int kHybridWordPrefix = 15;     // t = the prefix the form was planned for (0 .. kHybridMaxPrefix), written by the planner
constexpr int kHybridWordShift = 16;      // [4]: PassParams::shift of the g-th global pass = 16 - t + g x radix_bits
constexpr int kHybridWordLowBits = 20;    // the local stage's bits: 32 - t - bucket_bits
constexpr int kHybridWords = 21;
// A prefix of eight bits or more is a constant top byte: the ordinary form then skips a pass and moves as few bytes as the hybrid
// form would, so the form is only planned for 0 .. 7 shared bits.
constexpr uint32_t kHybridMaxPrefix = 7;
__host__ __device__ inline uint32_t hybrid_prefix_of(uint32_t differ) { return differ ? (uint32_t)__builtin_clz(differ) : 32u; }
// The global passes of the hybrid form cover bits 16-31: two at 8-bit digits, four at 4-bit digits.
inline constexpr int hybrid_global_passes(int radix_bits) { return 16 / radix_bits; }
// Count words of the form (zeroed with the workspace): the passes' [pass][digit][region] fields (8-bit: A from the upfront read,
// B from the planner; 4-bit: all four from the planner), at 4-bit digits the upfront read's joint field [position region][bits
// 16-23] the planner derives A and B from, then the buckets.
inline constexpr size_t hybrid_field_words(int radix_bits) { return radix_bits == 8 ? 2 * 2048 : 4 * 256; }
inline constexpr size_t hybrid_joint_words(int radix_bits) { return radix_bits == 8 ? 0 : 16 * 256; }
inline constexpr size_t hybrid_count_words(int radix_bits) { return hybrid_field_words(radix_bits) + hybrid_joint_words(radix_bits) + kHybridBuckets; }
// The key prefix: the top bits every key shares (keys below 2^31, a shard of an array partitioned by its top bits, ...).  Buckets
// are the bucket_bits bits BELOW it and the global passes' digits start at bit 16 - prefix; with the prefix inside them the buckets
// would be 2^(bucket_bits - prefix) non-empty ones, each 2^prefix times too large.  The sample finds it (words[kHybridWordDiffer]),
// the upfront read checks it against every key (words[kHybridWordViolated]), the planner writes what follows from it.
// A look at 65536 keys taken at a regular stride: words[Differ] |= key ^ first key, and words[Hopeless] = 1 if some bucket holds
// 0.8 % or more of a workgroup's 1024 samples (its share is 0.003 %): such keys cannot take the hybrid form, and the 0.2-0.3 ms of
// its upfront read are saved (zeros, a default value, small ranges, few-valued keys).
hipError_t launch_hybrid_sample(const uint32_t* keys, uint32_t n, int bucket_bits, uint32_t* words, hipStream_t stream);
// 8-bit digits: field[(digit of bits 16-23) * 8 + position region]; 4-bit digits: field[position region * 256 + bits 16-23] (the
// joint field); and bucket[key >> (32 - bucket_bits)] += counts (all zero on entry) -- bits counted below the prefix.
// Nothing if words[Hopeless] or the sampled prefix is eight bits or more.  xf: the keys are counted as to_sortable(key, xf)
// (typed sorts: what the first global pass will store).
hipError_t launch_hybrid_histograms(int radix_bits, const uint32_t* keys, uint32_t n, uint32_t region0_keys, uint32_t* field, uint32_t* bucket,
                                    int bucket_bits, uint32_t* words, hipStream_t stream, const KeyTransform& xf = KeyTransform{});
// verdict, bucket bases (2^bucket_bits + 1 words), plan words and the count fields the upfront read has not written: 8-bit digits
// fields_out = the second pass's field B [256][8] (joint unused); 4-bit digits fields_out = all four passes' [4][16][16], from
// joint and the buckets
// ... and large_list[0 .. words[kHybridWordLargeCount]): the buckets of more than small_cap keys (up to 2^bucket_bits words)
hipError_t launch_hybrid_plan(int radix_bits, const uint32_t* bucket, uint32_t n, int bucket_bits, uint32_t* bases, uint32_t* fields_out,
                              const uint32_t* joint, uint32_t* words, uint32_t* large_list, uint32_t small_cap, hipStream_t stream);

// counts64[b] = hist32[b], b < bins (multi-GPU bucket sizes as uint64).
hipError_t launch_widen_counts(const uint32_t* hist32, uint64_t* counts64, int bins, hipStream_t stream);

// *sticky |= *fault (stream-ordered): callers that run several sorts in ONE workspace keep each sort's fault word this way --
// the next sort's opening memset clears the workspace's own word (wide.hip, sharded.hip).
hipError_t launch_keep_fault(uint32_t* sticky, const uint32_t* fault, hipStream_t stream);

// *out = value (stream-ordered).
hipError_t launch_store_u64(uint64_t* out, uint64_t value, hipStream_t stream);

// out[0] = m = min(samples, n); out[1 + i] = keys[i * n / m], i < m (the splitter choice's sample of a shard).
hipError_t launch_sample_keys(const uint32_t* keys, uint32_t n, uint32_t samples, uint32_t* out, hipStream_t stream);

}  // namespace lsd
