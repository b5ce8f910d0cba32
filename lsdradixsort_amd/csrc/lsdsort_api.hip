// lsdsort_api.hip -- the C-ABI (include/lsdsort.h) and the host-side pass sequencing.
//
// Host counterpart of GPULSDRadixSort (.cu:839-910) and of the alloc/copy/sort/copy body of
// TestGPULSDRadixSort (.cu:966-1005).  Differences by design: one stream and explicit stream
// order (the reference leans on legacy default-stream implicit sync between its default
// stream and two private streams, .cu:841-842); nothing allocated inside the device entry;
// every HIP error is returned, never fatal; kernel launches are checked (the reference never
// calls cudaGetLastError).
#define LSDSORT_BUILD 1
#include "../../include/lsdsort.h"

#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>

#include "lsd_kernels.hpp"

namespace {

using lsd::PassParams;
using lsd::TileShape;

thread_local hipError_t g_last_hip = hipSuccess;

#define LSD_HIP(expr)                         \
    do {                                      \
        hipError_t e__ = (expr);              \
        if (e__ != hipSuccess) {              \
            g_last_hip = e__;                 \
            (void)hipGetLastError();          \
            return LSDSORT_ERR_HIP;           \
        }                                     \
    } while (0)

constexpr size_t kAlign = 256;
constexpr size_t kControlBytes = 512;            // u32[128]: [0] fault word, [16 .. 16 + kPlanWords) the pass plan, then the hybrid form's words
constexpr size_t kPlanOffsetWords = 16;
constexpr size_t kHybridOffsetWords = kPlanOffsetWords + lsd::kPlanWords + 1;   // the hybrid form's plan words (hybrid.hip)
static_assert(kHybridOffsetWords + lsd::kHybridWords <= kControlBytes / sizeof(uint32_t), "the plans live in the control block");
// what the bucket rule gives BASELINE's sizes: 2^28 keys 2^15 buckets of 8192, 2^27 keys 2^14 of 8192, 2^27 pairs 2^15 of 4096
static_assert(lsd::hybrid_bucket_bits((size_t)1 << 28, false) == 15 && lsd::hybrid_bucket_bits((size_t)1 << 27, false) == 14 &&
              lsd::hybrid_bucket_bits((size_t)1 << 27, true) == 15 && lsd::hybrid_bucket_bits((size_t)1 << 26, true) == 14, "bucket rule");
std::atomic<int> g_hybrid{[] {                                          // lsdsort_set_hybrid; LSDSORT_HYBRID=0 starts it off
    const char* e = getenv("LSDSORT_HYBRID");
    return (e && e[0] == '0') ? 0 : 1;
}()};
// The hybrid form is tried for 8-bit-digit sorts whose AVERAGE bucket (top 15 bits from 2^27 keys, top 14 below) leaves the local
// stage room: 4096 .. 14648 keys per bucket, 2^14 .. 2^16 buckets (below, tens of thousands of workgroups of almost nothing cost more than the two passes
// they replace; above, the largest bucket of even uniform keys nears the 16384-key capacity).  Whether it RUNS is decided on the
// device from the exact bucket counts.
// where the form starts to pay (2^14 buckets of 1024 .. 2400 keys there; measured with tools/size_perf.py): keys at 8-bit digits
// from 3.8e7 (2^25 keys: 101 Gkeys/s in four passes against 95; 4e7: 101 against 108), pairs from 2.2e7 (2.4e7: 61 against 66
// Gpairs/s; 2^24: 59 against 53), 4-bit digits from 2^24 (48 against 50 Gkeys/s; 2.4e7: 55 against 64)
inline size_t hybrid_min_items(int radix_bits, bool pairs)
{
    if (radix_bits == 4) return (size_t)1 << 24;
    return pairs ? (size_t)22 * 1000 * 1000 : (size_t)38 * 1000 * 1000;
}
constexpr size_t kHybridMaxKeys = (size_t)960 * 1000 * 1000;
// The capacity of the local stage's launch over all buckets: the smallest variant that holds what uniform keys put into a bucket
// (mean + 6 sigma); larger buckets go on the planner's list for the 16384-key variant.
int hybrid_small_cap(size_t n, bool pairs)
{
    static const bool tiny = [] { const char* e = getenv("LSDSORT_LOCAL_TINY"); return !(e && e[0] == '0'); }();   // experiment knob
    const double mean = (double)(n >> lsd::hybrid_bucket_bits(n, pairs));
    if (tiny && mean + 6.0 * std::sqrt(mean) <= (double)lsd::kLocalSortCapTiny) return lsd::kLocalSortCapTiny;
    const int small = pairs ? lsd::kLocalSortCapSmallPairs : lsd::kLocalSortCapSmall;
    // where most buckets of uniform keys are above the three-per-CU variant (the top of a bucket count's range: 6.7e8 .. 9.6e8
    // keys) the launch over all buckets is the 16384-key variant itself and the planner's list stays empty
    if (mean + 1.5 * std::sqrt(mean) > (double)small) return lsd::kLocalSortCap;
    return small;
}
std::atomic<int> g_small_sort{[] {                                      // lsdsort_set_small_sort; LSDSORT_SMALL_SORT=0 starts it off
    const char* e = getenv("LSDSORT_SMALL_SORT");
    return (e && e[0] == '0') ? 0 : 1;
}()};
std::atomic<int> g_skip_dead_passes{[] {                                // lsdsort_set_pass_skipping; LSDSORT_PASS_SKIPPING=0 starts it off
    const char* e = getenv("LSDSORT_PASS_SKIPPING");
    return (e && e[0] == '0') ? 0 : 1;
}()};
constexpr uint32_t kMaxXcdChunk = 64;

// Pass-0 regions are by position: R0 keys each, a multiple of the tile (every tile is a multiple of 4096 keys, the
// histogram kernel's chunk at 1024 threads; launch_joint_histograms checks), eight of them covering n.
uint32_t region0_keys(size_t n, size_t tile, int regions)
{
    const size_t per = (n + regions - 1) / regions;
    const size_t tiles = (per + tile - 1) / tile;
    return (uint32_t)((tiles ? tiles : 1) * tile);
}

size_t align_up(size_t x) { return (x + kAlign - 1) / kAlign * kAlign; }

bool valid_radix(int r) { return r == 1 || r == 2 || r == 4 || r == 8; }

std::atomic<int> g_shape_override[9];   // per radix_bits: 0 = default, k + 1 = compiled shape k forced

// Tile shape of a sort: the one lsdsort_set_tile_config pinned, else the compiled default for the
// job.  The chained form picks by size (tools/size_sweep.py, shape ids of aux_kernels.hip):
//   n >= 2^23 : the one-workgroup-per-CU 1024x32 tile (32768 keys: half the status rows per key, 32
//               instead of 64 tiles in flight per chain; fastest on uniform keys and the most even
//               across key distributions);
//   n >= 2^21 : 16384 keys;   n >= 2^19 : 8192 keys;   below : 4096 keys -- small sorts are launch-bound
//               (seven launches, ~40 us) and need enough tiles to occupy 256 CUs at all.
// Everything else -- the staged form and the stage-level entries, whose tables callers index by
// lsdsort_tile_keys() -- uses shape 0.
struct ShapeClass {
    size_t below;   // applies to n < below
    int shape8, shape4, shape_narrow;
};
constexpr ShapeClass kShapeClasses[] = {
    {(size_t)1 << 19, 5, 1, 0},
    {(size_t)1 << 21, 3, 1, 0},
    {(size_t)1 << 23, 0, 0, 1},
    {~(size_t)0, 4, 4, 2},
};
constexpr int kNumShapeClasses = (int)(sizeof(kShapeClasses) / sizeof(kShapeClasses[0]));

int class_shape(const ShapeClass& c, int radix_bits)
{
    return radix_bits == 8 ? c.shape8 : (radix_bits == 4 ? c.shape4 : c.shape_narrow);
}

const TileShape* current_shape(int radix_bits, bool pairs = false, size_t n = 0, int algorithm = LSDSORT_ALGO_STAGED)
{
    (void)pairs;
    const TileShape* shapes = nullptr;
    const int count = lsd::tile_shapes(radix_bits, &shapes);
    if (count == 0) return nullptr;
    int id = g_shape_override[radix_bits].load(std::memory_order_relaxed) - 1;
    if (id < 0) {
        id = 0;
        if (algorithm == LSDSORT_ALGO_ONESWEEP && n > 0) {
            for (int c = 0; c < kNumShapeClasses; c++)
                if (n < kShapeClasses[c].below) {
                    id = class_shape(kShapeClasses[c], radix_bits);
                    break;
                }
        }
    }
    if (id >= count) id = 0;
    return &shapes[id];
}

// Device workspace carve-up.  Everything before `zero_bytes` is cleared at the start of a sort.
struct Layout {
    size_t control = 0;      // fault word
    size_t tickets = 0;      // onesweep: [P][8] arrival ticket dispensers
    size_t counts = 0;       // onesweep: [P][H][regions] joint / digit counts
    size_t status = 0;       // onesweep: [rows][H] tile-status words, even passes
    size_t zero_bytes = 0;
    size_t status_odd = 0;   // onesweep: the same for odd passes (cleared by the even pass before it, and vice versa)
    size_t tables = 0;       // onesweep: [P] region tables
    size_t tile_hist = 0;    // staged: [tiles][H] counts, then local offsets in place
    size_t tile_global = 0;  // staged: [tiles][H] global offsets
    size_t scratch = 0;      // staged: strip sums
    size_t alt_keys = 0;
    size_t alt_vals = 0;
    size_t alt_more[2] = {0, 0};   // further payload arrays (records: lsdsort_multi_u32_device)
    size_t hyb_counts = 0;         // hybrid form: the global passes' count fields | (4-bit digits: the joint field) | buckets [32768], zeroed
    size_t hyb_bases = 0;          // [32769] bucket bases
    size_t total = 0;
    uint32_t tiles = 0;      // ceil(n / tile)
    uint32_t rows = 0;       // onesweep: status rows = grid size = tiles + one ragged tile per region
    uint32_t region0 = 0;    // onesweep: keys per pass-0 region
    int regions = 1;
};

// payloads: number of 32-bit payload arrays that travel with the keys (0 = keys only, 1 = pairs, up to 3)
Layout make_layout(size_t n, int radix_bits, int payloads, int algorithm, const TileShape& shape)
{
    const bool pairs = payloads > 0;
    Layout L;
    const size_t bins = (size_t)1 << radix_bits;
    const size_t tile = (size_t)shape.tile();
    const size_t passes = 32 / radix_bits;
    L.tiles = (uint32_t)((n + tile - 1) / tile);
    size_t off = 0;
    L.control = off;
    off += kControlBytes;
    if (algorithm == LSDSORT_ALGO_ONESWEEP) {
        L.regions = lsd::regions_for_radix(radix_bits);
        L.rows = L.tiles + (uint32_t)L.regions;          // one ragged last tile per region at most
        L.region0 = region0_keys(n, tile, L.regions);
        // the hybrid form's global passes (two at 8-bit digits, four at 4-bit) have ticket, count and table slots of their own
        const size_t hyb = (radix_bits == 8 || radix_bits == 4) ? (size_t)lsd::hybrid_global_passes(radix_bits) : 0;
        L.tickets = off;
        off = align_up(off + (passes + hyb) * lsd::kMaxRegions * sizeof(uint32_t));
        L.counts = off;
        off = align_up(off + passes * bins * (size_t)L.regions * sizeof(uint32_t));
        if (hyb) {
            L.hyb_counts = off;
            off = align_up(off + lsd::hybrid_count_words(radix_bits) * sizeof(uint32_t));
        }
        L.status = off;
        off = align_up(off + (size_t)L.rows * bins * sizeof(uint32_t));
        L.zero_bytes = off;
        L.status_odd = off;
        off = align_up(off + (size_t)L.rows * bins * sizeof(uint32_t));
        L.tables = off;
        off = align_up(off + (passes + hyb) * lsd::region_table_words(radix_bits) * sizeof(uint32_t));
        if (hyb) {
            L.hyb_bases = off;
            off = align_up(off + (size_t)(2 * lsd::kHybridBuckets + 1) * sizeof(uint32_t));   // bases, then the list of large buckets
        }
    } else {
        L.zero_bytes = off;
        L.tile_hist = off;
        off = align_up(off + (size_t)L.tiles * bins * sizeof(uint32_t));
        L.tile_global = off;
        off = align_up(off + (size_t)L.tiles * bins * sizeof(uint32_t));
        L.scratch = off;
        off = align_up(off + lsd::tile_offsets_scratch_words(L.tiles, radix_bits) * sizeof(uint32_t));
    }
    L.alt_keys = off;
    off = align_up(off + n * sizeof(uint32_t));
    if (pairs) {
        L.alt_vals = off;
        off = align_up(off + n * sizeof(uint32_t));
    }
    for (int e = 0; e + 1 < payloads && e < 2; e++) {
        L.alt_more[e] = off;
        off = align_up(off + n * sizeof(uint32_t));
    }
    L.total = off;
    return L;
}

// Per-device facts learned once: is it gfx950, and does it serve colliding LDS atomics of one
// wave instruction in lane order (needed by kRankLdsAdd)?  The probe allocates and synchronises,
// so it runs once per device, on first use or from lsdsort_prepare_device(), never per sort.
struct DeviceState {
    std::atomic<int> state{0};   // 0 unknown, 1 usable, -1 not gfx950
    bool lds_add_in_lane_order = false;
};
DeviceState g_device[64];
std::mutex g_device_mutex;

int check_device_ready(int* device_out = nullptr)
{
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
        (void)hipGetLastError();
        return LSDSORT_ERR_NO_DEVICE;
    }
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) {
        (void)hipGetLastError();
        return LSDSORT_ERR_NO_DEVICE;
    }
    if (device_out) *device_out = dev;
    DeviceState& st = g_device[dev];
    int state = st.state.load(std::memory_order_acquire);
    if (state == 0) {
        std::lock_guard<std::mutex> lock(g_device_mutex);
        state = st.state.load(std::memory_order_acquire);
        if (state == 0) {
            hipDeviceProp_t prop;
            if (hipGetDeviceProperties(&prop, dev) != hipSuccess) {
                (void)hipGetLastError();
                return LSDSORT_ERR_NO_DEVICE;
            }
            if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {   // code objects are gfx950 only
                st.state.store(-1, std::memory_order_release);
                return LSDSORT_ERR_NO_DEVICE;
            }
            bool ok = false;
            LSD_HIP(lsd::probe_lds_add_lane_order(&ok, nullptr));
            st.lds_add_in_lane_order = ok;
            st.state.store(1, std::memory_order_release);
            state = 1;
        }
    }
    return state == 1 ? LSDSORT_OK : LSDSORT_ERR_NO_DEVICE;
}

std::atomic<uint32_t> g_xcd_chunk{16};   // consecutive tiles kept on one XCD
std::atomic<unsigned long long*> g_stats{nullptr};   // diagnostic builds only
std::atomic<uint32_t> g_spin_limit{lsd::kSpinLimit};   // empty look-back polls before a tile gives up
std::atomic<uint32_t> g_mute_row{0};                   // diagnostic builds only (LSD_FAULT_INJECT)
[[maybe_unused]] std::atomic<int> g_mute_sorts{-1};                     // diagnostic builds only: sorts the muted row still applies to (-1: all)
std::atomic<int> g_rank_setting{-1};   // -1 auto, 0 mask forms only, 2 returning LDS add wherever probed ok

#ifdef LSD_FAULT_INJECT
// Diagnostic build only: counts[from] -= delta, counts[to] += delta behind stage 1 (word indices into the sort's
// [pass][digit][region] count table), i.e. tables that do not describe the keys -- what a miscounting stage-1 variant
// produces.  tests/test_fault_path.py drives the destination guard with it, once.
struct CorruptCounts {
    uint32_t from = 0, to = 0, delta = 0, keep_sum = 1;   // keep_sum bit 1: the words are the HYBRID form's count fields
};
std::mutex g_corrupt_mutex;
CorruptCounts g_corrupt;
__global__ void corrupt_counts_kernel(uint32_t* counts, CorruptCounts c)
{
    if (c.keep_sum & 1u) counts[c.from] -= c.delta;
    counts[c.to] += c.delta;
}
#endif

// The rank form a sort on `dev` with this radix will use.
int resolve_rank_method(int dev, int radix_bits)
{
    const int setting = g_rank_setting.load(std::memory_order_relaxed);
    const bool ok = g_device[dev].lds_add_in_lane_order;
    if (setting == 0 || !ok) return radix_bits > 4 ? lsd::kRankLdsOr : lsd::kRankBallot;
    if (setting == 2) return lsd::kRankLdsAdd;
    // auto: one LDS op per key beats the ballots (or an OR round trip) from 4-bit digits up -- measured
    // 0.55 -> 0.47 ms per pass at 4 bits, where the four ballots cost ~28 vector instructions a key;
    // with 2..8 counters per wave nearly every lane collides and the ballots are level or ahead.
    return radix_bits >= 4 ? lsd::kRankLdsAdd : lsd::kRankBallot;
}

struct StageEvents {
    hipEvent_t ev[3 * LSDSORT_MAX_PASSES + 4];
    hipEvent_t kernel_ev[2 * LSDSORT_MAX_PASSES];   // begin/end of each pass's rank-and-scatter kernel
    int kernel_count = 0;
    int count = 0;
    bool enabled = false;
    hipStream_t stream = nullptr;
    int arm_kernel_events()   // the next rank-and-scatter launch of this thread reports into a fresh pair
    {
        if (!enabled || kernel_count + 2 > 2 * LSDSORT_MAX_PASSES) return LSDSORT_OK;
        LSD_HIP(hipEventCreate(&kernel_ev[kernel_count]));
        kernel_count++;
        LSD_HIP(hipEventCreate(&kernel_ev[kernel_count]));
        kernel_count++;
        lsd::t_launch_start = kernel_ev[kernel_count - 2];
        lsd::t_launch_stop = kernel_ev[kernel_count - 1];
        return LSDSORT_OK;
    }
    static void disarm_kernel_events() { lsd::t_launch_start = lsd::t_launch_stop = nullptr; }
    int mark()
    {
        if (!enabled) return LSDSORT_OK;
        LSD_HIP(hipEventCreate(&ev[count]));
        LSD_HIP(hipEventRecord(ev[count], stream));
        count++;
        return LSDSORT_OK;
    }
    void destroy()
    {
        disarm_kernel_events();
        for (int i = 0; i < count; i++) (void)hipEventDestroy(ev[i]);
        for (int i = 0; i < kernel_count; i++) (void)hipEventDestroy(kernel_ev[i]);
        count = kernel_count = 0;
    }
};

#define LSD_TRY(expr)                  \
    do {                               \
        int s__ = (expr);              \
        if (s__ != LSDSORT_OK) return s__; \
    } while (0)

// Host-pointer entry (lsdsort_u32 and friends): the input arrives over PCIe in chunks on a copy stream and each
// chunk's share of the upfront histogram runs behind it on the sort's stream, so stage 1 is hidden under the
// transfer (the passes need every key and cannot start earlier; the copy back needs the last pass).
struct HostFeed {
    const uint32_t* host_keys = nullptr;
    const uint32_t* host_vals = nullptr;
    hipStream_t copy = nullptr;
    hipEvent_t* events = nullptr;   // one per chunk, plus one for the payloads
};
#ifndef LSD_FEED_CHUNK_LOG2
#define LSD_FEED_CHUNK_LOG2 24
#endif
const size_t kFeedChunkKeys = [] {                                      // 64 MiB per chunk
    const char* e = getenv("LSDSORT_FEED_CHUNK_LOG2");                  // experiment knob (tools/host_entry_perf.py)
    const int v = e ? atoi(e) : LSD_FEED_CHUNK_LOG2;
    return (size_t)1 << (v >= 22 && v <= 30 ? v : LSD_FEED_CHUNK_LOG2);
}();
constexpr int kMaxFeedEvents = (int)(((size_t)LSDSORT_MAX_KEYS >> 22) + 3);

// The pass loop.  Marks (when timing), onesweep: 0 start | 1 after clear | 2 after stage 1 |
// 3 after stage 2 | 4 after the last pass.  Staged: 0 start | 1 after clear | 2, 3 (empty) | then per
// pass three marks: after its histogram, after its offset scan, after its scatter.
int run_sort(uint32_t* d_keys, uint32_t* d_vals, void* d_ws, size_t ws_bytes, size_t n, int radix_bits,
             int algorithm, hipStream_t stream, StageEvents* ev, lsdsort_timing* timing,
             const lsd::KeyTransform& xf = lsd::KeyTransform{}, const HostFeed* feed = nullptr,
             uint32_t* const* d_more = nullptr, int more = 0,   // further payload arrays (0..2), chained form only
             int prefix = 0)   // lsdsort_u32_device_prefixed's argument: checked, not needed (the device finds the key prefix itself)
{
    if (prefix < 0 || prefix > 8) return LSDSORT_ERR_INVALID_ARG;
    if (more < 0 || more > 2 || (more > 0 && (!d_vals || !d_more || algorithm != LSDSORT_ALGO_ONESWEEP || feed))) return LSDSORT_ERR_INVALID_ARG;
    for (int e = 0; e < more; e++)
        if (n > 0 && !d_more[e]) return LSDSORT_ERR_INVALID_ARG;
    if (feed && algorithm != LSDSORT_ALGO_ONESWEEP) return LSDSORT_ERR_INVALID_ARG;
    if (xf.on && (algorithm != LSDSORT_ALGO_ONESWEEP || radix_bits < 4)) return LSDSORT_ERR_UNSUPPORTED;
    if (!valid_radix(radix_bits)) return LSDSORT_ERR_INVALID_ARG;
    if (algorithm != LSDSORT_ALGO_ONESWEEP && algorithm != LSDSORT_ALGO_STAGED) return LSDSORT_ERR_INVALID_ARG;
    if (n > LSDSORT_MAX_KEYS) return LSDSORT_ERR_TOO_LARGE;
    if (n == 0) return LSDSORT_OK;
    if (!d_keys) return LSDSORT_ERR_INVALID_ARG;
    int dev = 0;
    LSD_TRY(check_device_ready(&dev));
    const int rank_method = resolve_rank_method(dev, radix_bits);
    const bool pairs = d_vals != nullptr;
    const TileShape* shape = current_shape(radix_bits, pairs, n, algorithm);
    if (!shape) return LSDSORT_ERR_INVALID_ARG;
    const Layout L = make_layout(n, radix_bits, pairs ? 1 + more : 0, algorithm, *shape);
    if (!d_ws || (reinterpret_cast<uintptr_t>(d_ws) & (kAlign - 1)) || ws_bytes < L.total) return LSDSORT_ERR_WORKSPACE;

    char* ws = static_cast<char*>(d_ws);
    uint32_t* control = reinterpret_cast<uint32_t*>(ws + L.control);
    uint32_t* alt_keys = reinterpret_cast<uint32_t*>(ws + L.alt_keys);
    uint32_t* alt_vals = pairs ? reinterpret_cast<uint32_t*>(ws + L.alt_vals) : nullptr;
    const int passes = 32 / radix_bits;
    uint32_t mute_row = g_mute_row.load(std::memory_order_relaxed);
#ifdef LSD_FAULT_INJECT
    {   // "the next k sorts only" (lsdsort_debug_fault_inject_sorts): e.g. the first of the two sorts inside lsdsort_u64_device
        int left = g_mute_sorts.load(std::memory_order_relaxed);
        if (left == 0) mute_row = 0;
        else if (left > 0) g_mute_sorts.store(left - 1, std::memory_order_relaxed);
    }
#endif
    if (timing) {
        timing->passes = passes;
        timing->tile_keys = shape->tile();
        timing->tiles = (int)L.tiles;
    }

    // Up to 16384 items: one workgroup sorts them in its LDS, one launch (local_sort.hip) -- at this size the chained form's eight
    // launches are nothing but their own latencies.  Plain uint32 keys and pairs with the returning-add rank form; not when timed.
    if (g_small_sort.load(std::memory_order_relaxed) && n <= (size_t)lsd::kLocalSortCap && algorithm == LSDSORT_ALGO_ONESWEEP && !ev && !timing && !xf.on && more == 0 && !feed &&
        rank_method == lsd::kRankLdsAdd && mute_row == 0) {
        LSD_HIP(lsd::launch_small_sort(d_keys, d_vals, (uint32_t)n, control, control + kHybridOffsetWords + lsd::kHybridWordOk, stream));
        return LSDSORT_OK;
    }

    if (ev) LSD_TRY(ev->mark());
    LSD_HIP(hipMemsetAsync(ws, 0, L.zero_bytes, stream));
    if (ev) LSD_TRY(ev->mark());

    uint32_t* tables = nullptr;
    const size_t table_words = lsd::region_table_words(radix_bits);
    // Passes whose digit is the same for every key are the identity: stage 2 sees that in the counts and writes a plan the
    // pass kernels follow (lsd_kernels.hpp, PassParams::plan) -- small key ranges, dead digits and constant input then cost
    // the passes that move something, plus one copy if their number is odd.  Not for typed sorts (their first and last
    // pass carry the key transform) nor where the plan would not fit the control block.
    // A typed sort gets a plan only where the hybrid form is tried (its kernels are told by the plan which form runs), and that plan
    // never skips a pass.
    const bool try_hybrid = algorithm == LSDSORT_ALGO_ONESWEEP && (radix_bits == 8 || radix_bits == 4) && !feed &&
                            rank_method == lsd::kRankLdsAdd && n >= hybrid_min_items(radix_bits, pairs) &&
                            n <= kHybridMaxKeys &&
                            (n >> lsd::hybrid_bucket_bits(n, pairs)) <= lsd::kHybridMaxMeanBucket && shape->tile() == 32768 &&
                            g_hybrid.load(std::memory_order_relaxed);
    uint32_t* plan = nullptr;
    if (algorithm == LSDSORT_ALGO_ONESWEEP && (!xf.on || try_hybrid) && 2 * passes + 1 <= lsd::kPlanWords &&
        g_skip_dead_passes.load(std::memory_order_relaxed))
        plan = control + kPlanOffsetWords;
    // The hybrid form (hybrid.hip): two global passes on the high bytes, then every top-15-bit bucket finished in LDS -- 28 B/key
    // instead of 36.  Tried here for keys-only sorts of the sizes it pays for; the device decides from the exact bucket counts, and
    // every kernel of the form that does NOT run returns at once (plan words in the control block).
    uint32_t* hyb = nullptr;
    if (plan && try_hybrid) hyb = control + kHybridOffsetWords;
    if (timing) timing->hybrid = hyb ? -1 : 0;   // -1: tried; lsdsort_u32_device_timed reads the device's verdict back
    if (algorithm == LSDSORT_ALGO_ONESWEEP) {
        uint32_t* counts = reinterpret_cast<uint32_t*>(ws + L.counts);
        tables = reinterpret_cast<uint32_t*>(ws + L.tables);
        if (hyb) {
            // 8-bit digits: field A (upfront read) | field B (planner).  4-bit: fields A-D (planner) | the joint field (upfront read)
            uint32_t* fields = reinterpret_cast<uint32_t*>(ws + L.hyb_counts);
            uint32_t* joint = fields + lsd::hybrid_field_words(radix_bits);
            uint32_t* bucket = joint + lsd::hybrid_joint_words(radix_bits);
            const int bb = lsd::hybrid_bucket_bits(n, pairs);
            LSD_HIP(lsd::launch_hybrid_sample(d_keys, (uint32_t)n, bb, hyb, stream));
            LSD_HIP(lsd::launch_hybrid_histograms(radix_bits, d_keys, (uint32_t)n, L.region0, radix_bits == 8 ? fields : joint, bucket, bb,
                                                  hyb, stream, xf));
            uint32_t* bases = reinterpret_cast<uint32_t*>(ws + L.hyb_bases);
            LSD_HIP(lsd::launch_hybrid_plan(radix_bits, bucket, (uint32_t)n, bb, bases, radix_bits == 8 ? fields + 2048 : fields, joint, hyb,
                                            bases + lsd::kHybridBuckets + 1,
                                            (uint32_t)hybrid_small_cap(n, pairs), stream));
#ifdef LSD_FAULT_INJECT
            {   // diagnostic build only: the hybrid form's own fields falsified behind the planner (its verdict stands)
                CorruptCounts c;
                {
                    std::lock_guard<std::mutex> lock(g_corrupt_mutex);
                    c = g_corrupt;
                }
                if (c.delta && (c.keep_sum & 2u)) {
                    hipLaunchKernelGGL(corrupt_counts_kernel, dim3(1), dim3(1), 0, stream, fields, c);
                    LSD_HIP(hipGetLastError());
                }
            }
#endif
            // the global passes' region tables: the first one's regions are by position (like any first pass), the others' by the
            // top bits of the digit before -- exactly what stage 2 builds for consecutive passes
            // (no fault word: where the sample or the planner has said no these counts are partial or absent, and nobody uses the tables)
            LSD_HIP(lsd::launch_scan_regions(radix_bits, lsd::hybrid_global_passes(radix_bits), L.regions, fields, (uint32_t)n,
                                             (uint32_t)shape->tile(), L.region0, tables + (size_t)passes * table_words, stream, nullptr, nullptr));
        }
        // stage 1 over [first, first + len): the whole array at once, or chunk by chunk behind the host's copies
        auto histogram = [&](size_t first, size_t len) -> int {
            if (L.regions > 1)
                LSD_HIP(lsd::launch_joint_histograms(radix_bits, d_keys + first, (uint32_t)len, L.region0, counts, stream, xf, (uint32_t)first,
                                                     hyb ? hyb + lsd::kHybridWordOk : nullptr));
            else
                LSD_HIP(lsd::launch_digit_histograms(radix_bits, passes, 0, d_keys + first, (uint32_t)len, counts, stream));
            return LSDSORT_OK;
        };
        if (!feed) {
            LSD_TRY(histogram(0, n));
        } else {
            int e = 0;
            for (size_t first = 0; first < n; first += kFeedChunkKeys, e++) {
                const size_t len = n - first < kFeedChunkKeys ? n - first : kFeedChunkKeys;
                LSD_HIP(hipMemcpyAsync(d_keys + first, feed->host_keys + first, len * sizeof(uint32_t), hipMemcpyHostToDevice, feed->copy));   // .cu:1001
                LSD_HIP(hipEventRecord(feed->events[e], feed->copy));
                LSD_HIP(hipStreamWaitEvent(stream, feed->events[e], 0));
                LSD_TRY(histogram(first, len));
            }
            if (pairs) {
                LSD_HIP(hipMemcpyAsync(d_vals, feed->host_vals, n * sizeof(uint32_t), hipMemcpyHostToDevice, feed->copy));
                LSD_HIP(hipEventRecord(feed->events[e], feed->copy));
                LSD_HIP(hipStreamWaitEvent(stream, feed->events[e], 0));
            }
        }
        if (ev) LSD_TRY(ev->mark());
#ifdef LSD_FAULT_INJECT
        {   // diagnostic build only: move counts between two bins so that the tables no longer describe the keys
            CorruptCounts c;
            {
                std::lock_guard<std::mutex> lock(g_corrupt_mutex);
                c = g_corrupt;
            }
            if (c.delta && !(c.keep_sum & 2u)) {
                hipLaunchKernelGGL(corrupt_counts_kernel, dim3(1), dim3(1), 0, stream, counts, c);
                LSD_HIP(hipGetLastError());
            }
        }
#endif
        LSD_HIP(lsd::launch_scan_regions(radix_bits, passes, L.regions, counts, (uint32_t)n, (uint32_t)shape->tile(),
                                         L.region0, tables, stream, plan, control, hyb ? hyb + lsd::kHybridWordOk : nullptr, !xf.on));
        if (ev) LSD_TRY(ev->mark());
        if (hyb) {
            // bits 16-31, lowest digit first: caller's buffer -> alternate and back (an even number of passes); then the buckets in place
            const int hyb_passes = lsd::hybrid_global_passes(radix_bits);
            for (int g = 0; g < hyb_passes; g++) {
                PassParams p{};
                p.in = d_keys;
                p.out = alt_keys;
                p.vals_in = d_vals;
                p.vals_out = alt_vals;
                p.num_payloads = pairs ? (uint32_t)(1 + more) : 0u;
                for (int e = 0; e < more; e++) {   // the plan word says which way round (as for the first payload array)
                    p.more_in[e] = d_more[e];
                    p.more_out[e] = reinterpret_cast<uint32_t*>(ws + L.alt_more[e]);
                }
                p.n = (uint32_t)n;
                p.shift = (uint32_t)(16 + radix_bits * g);
                p.shift_word = hyb + lsd::kHybridWordShift + g;   // ... less the key prefix the device found: bits [16 - t, 32 - t)
                p.num_tiles = L.rows;
                p.fault = control;
                p.spin_limit = g_spin_limit.load(std::memory_order_relaxed);
                p.regions = tables + (size_t)(passes + g) * table_words;
                p.status = reinterpret_cast<uint32_t*>(ws + ((g & 1) ? L.status_odd : L.status));
                p.status_clear = g + 1 < hyb_passes ? reinterpret_cast<uint32_t*>(ws + ((g & 1) ? L.status : L.status_odd)) : nullptr;
                p.tickets = reinterpret_cast<uint32_t*>(ws + L.tickets) + (size_t)(passes + g) * lsd::kMaxRegions;
                p.plan = hyb + lsd::kHybridWordPlan + 2 * g;
                if (xf.on && g == 0) p.xin = xf;   // typed sorts: sortable keys from the first store on; the local stage's store turns them back
                p.stats = g_stats.load(std::memory_order_relaxed);
                if (ev) LSD_TRY(ev->arm_kernel_events());
                const hipError_t launched = lsd::launch_rank_scatter(radix_bits, *shape, rank_method, true, p, stream);
                StageEvents::disarm_kernel_events();
                LSD_HIP(launched);
            }
            lsd::LocalSortParams lp{};
            lp.keys = d_keys;
            lp.vals = d_vals;
            lp.num_payloads = pairs ? (uint32_t)(1 + more) : 0u;
            for (int e = 0; e < more; e++) lp.more[e] = d_more[e];
            lp.bases = reinterpret_cast<const uint32_t*>(ws + L.hyb_bases);
            const int bb = lsd::hybrid_bucket_bits(n, pairs);
            lp.num_buckets = 1u << bb;
            lp.low_bits_word = hyb + lsd::kHybridWordLowBits;   // the bits below a bucket's own: 32 - bb less the key prefix the device found
            lp.skip = hyb + lsd::kHybridWordSkipLocal;
            lp.xout = xf;
            lp.fault = control;
            if (ev) LSD_TRY(ev->mark());
            // buckets of up to 10240 keys (all of them on uniform keys of most sizes): three workgroups per CU; where uniform keys
            // stay under 5120 a bucket, four
            const int all_cap = hybrid_small_cap(n, pairs);
            lp.small_variant = all_cap == lsd::kLocalSortCapTiny ? 2u : all_cap == lsd::kLocalSortCap ? 0u : 1u;
            lp.larger_elsewhere = 1;
            LSD_HIP(lsd::launch_local_sort(lp, stream));
            lp.small_variant = 0;     // the planner's list of larger ones (up to 16384 keys): two per CU, a grid of 512 walks the list
            lp.larger_elsewhere = 0;
            lp.list = lp.bases + lsd::kHybridBuckets + 1;
            lp.list_count = hyb + lsd::kHybridWordLargeCount;
            LSD_HIP(lsd::launch_local_sort(lp, stream));
            if (ev) LSD_TRY(ev->mark());
        }
    } else if (ev) {
        LSD_TRY(ev->mark());
        LSD_TRY(ev->mark());
    }

    uint32_t* src = d_keys;
    uint32_t* dst = alt_keys;
    uint32_t* vsrc = d_vals;
    uint32_t* vdst = alt_vals;
    for (int pass = 0; pass < passes; pass++) {
        PassParams p{};
        p.in = src;
        p.out = dst;
        p.vals_in = vsrc;
        p.vals_out = vdst;
        p.n = (uint32_t)n;
        p.shift = (uint32_t)(pass * radix_bits);
        p.num_tiles = L.tiles;
        p.fault = control;
        p.spin_limit = g_spin_limit.load(std::memory_order_relaxed);
        p.mute_row = mute_row;
        p.xcd_chunk = g_xcd_chunk.load(std::memory_order_relaxed);
        p.stats = g_stats.load(std::memory_order_relaxed);
        if (algorithm == LSDSORT_ALGO_ONESWEEP) {
            p.num_tiles = L.rows;
            p.regions = tables + (size_t)pass * table_words;
            // The status rows a pass uses depend on its regions, so a row may sit out a pass and the
            // parity-coded reuse of lsd_device.hpp (every word rewritten every pass) does not apply.
            // Two status arrays instead: a pass works in one and its workgroups clear the other, one
            // row each (the grid has exactly `rows` workgroups), for the pass after it -- 1 KiB of plain
            // stores per 64 KiB tile and no launch between the passes.  The sort's opening memset
            // covers the even array.
            p.status = reinterpret_cast<uint32_t*>(ws + ((pass & 1) ? L.status_odd : L.status));
            p.status_clear = pass + 1 < passes ? reinterpret_cast<uint32_t*>(ws + ((pass & 1) ? L.status : L.status_odd)) : nullptr;
            p.tickets = reinterpret_cast<uint32_t*>(ws + L.tickets) + (size_t)pass * lsd::kMaxRegions;
            p.parity = 0;
            // further payload arrays ping-pong like the first: even passes read the caller's, odd passes the alternates
            p.num_payloads = pairs ? (uint32_t)(1 + more) : 0u;
            for (int e = 0; e < more; e++) {
                uint32_t* mine = d_more[e];
                uint32_t* alt = reinterpret_cast<uint32_t*>(ws + L.alt_more[e]);
                p.more_in[e] = (pass & 1) && !plan ? alt : mine;
                p.more_out[e] = (pass & 1) && !plan ? mine : alt;
            }
            if (plan) {   // the same pair for every pass: the plan says which way round (and whether at all)
                p.in = d_keys;
                p.out = alt_keys;
                p.vals_in = d_vals;
                p.vals_out = alt_vals;
                p.plan = plan + 2 * pass;
                p.plan_first = hyb ? 1u : 0u;   // a sort that tried the hybrid form: these passes leave at once if it runs
            }
            if (xf.on && pass == 0) p.xin = xf;
            if (xf.on && pass + 1 == passes) p.xout = xf;
            if (ev) LSD_TRY(ev->arm_kernel_events());
            const hipError_t launched = lsd::launch_rank_scatter(radix_bits, *shape, rank_method, true, p, stream);
            StageEvents::disarm_kernel_events();
            LSD_HIP(launched);
        } else {
            uint32_t* tile_hist = reinterpret_cast<uint32_t*>(ws + L.tile_hist);
            uint32_t* tile_global = reinterpret_cast<uint32_t*>(ws + L.tile_global);
            uint32_t* scratch = reinterpret_cast<uint32_t*>(ws + L.scratch);
            LSD_HIP(lsd::launch_tile_histograms(radix_bits, *shape, src, (uint32_t)n, p.shift, tile_hist, stream));
            if (ev) LSD_TRY(ev->mark());
            // local offsets are recomputed inside the rank-and-scatter kernel (it needs them for
            // ranking anyway); only the global table is materialised here.
            LSD_HIP(lsd::launch_tile_offsets(radix_bits, tile_hist, nullptr, tile_global, L.tiles, scratch, stream));
            if (ev) LSD_TRY(ev->mark());
            p.global_off = tile_global;
            LSD_HIP(lsd::launch_rank_scatter(radix_bits, *shape, rank_method, false, p, stream));
        }
        // Chained form: the passes are launched back to back and only the last one is followed by a
        // mark -- an event record between two dependent kernels costs ~20 us of queue bubble, which is
        // not the kernel's time (rocprofv3's kernel trace would disagree by 5 %).
        if (ev && (algorithm != LSDSORT_ALGO_ONESWEEP || pass + 1 == passes)) LSD_TRY(ev->mark());
        uint32_t* t = src; src = dst; dst = t;
        t = vsrc; vsrc = vdst; vdst = t;
    }
    // 32 / radix_bits is even for every accepted radix: the result is back in d_keys/d_vals,
    // as the reference relies on (.cu:905, .cu:1005) -- unless the plan skipped an odd number of passes
    if (plan) {
        LSD_HIP(lsd::launch_finish_plan(plan + 2 * passes, d_keys, alt_keys, d_vals, alt_vals, (uint32_t)n, stream));
        for (int e = 0; e < more; e++)   // the same copy back for each further payload array
            LSD_HIP(lsd::launch_finish_plan(plan + 2 * passes, d_more[e], reinterpret_cast<uint32_t*>(ws + L.alt_more[e]), nullptr, nullptr,
                                            (uint32_t)n, stream));
    }
    return LSDSORT_OK;
}

int read_fault(void* d_ws, hipStream_t stream)
{
    uint32_t fault = 0;
    LSD_HIP(hipMemcpyAsync(&fault, d_ws, sizeof(fault), hipMemcpyDeviceToHost, stream));
    LSD_HIP(hipStreamSynchronize(stream));
    return fault ? LSDSORT_ERR_DEVICE_FAULT : LSDSORT_OK;
}

// Device buffers, streams and events of the host-pointer entries, kept per device between calls (the reference
// allocates and frees around every sort, .cu:967-975 / .cu:1020-1028; a hipMalloc + hipFree pair of this size costs
// milliseconds).  Calls on one device take turns (the mutex); lsdsort_release_host_cache() gives the memory back.
struct HostCache {
    std::mutex mutex;
    uint32_t* d_keys = nullptr;
    uint32_t* d_vals = nullptr;
    void* d_ws = nullptr;
    size_t keys_bytes = 0, vals_bytes = 0, ws_bytes = 0;
    hipStream_t copy = nullptr, compute = nullptr;
    hipEvent_t events[kMaxFeedEvents] = {};
    bool ready = false;
};
HostCache g_host_cache[64];

int grow(void** ptr, size_t* have, size_t need)
{
    if (*have >= need) return LSDSORT_OK;
    if (*ptr) {
        (void)hipFree(*ptr);
        *ptr = nullptr;
        *have = 0;
    }
    LSD_HIP(hipMalloc(ptr, need));
    *have = need;
    return LSDSORT_OK;
}

int sort_host(uint32_t* keys, uint32_t* vals, size_t n, int radix_bits)
{
    if (!valid_radix(radix_bits)) return LSDSORT_ERR_INVALID_ARG;
    if (n > LSDSORT_MAX_KEYS) return LSDSORT_ERR_TOO_LARGE;
    if (n == 0) return LSDSORT_OK;
    if (!keys) return LSDSORT_ERR_INVALID_ARG;
    int dev = 0;
    LSD_TRY(check_device_ready(&dev));
    const bool pairs = vals != nullptr;
    const size_t ws_bytes = lsdsort_workspace_bytes_ex(n, radix_bits, pairs ? 1 : 0, LSDSORT_ALGO_ONESWEEP);
    const size_t bytes = n * sizeof(uint32_t);
    HostCache& C = g_host_cache[dev];
    std::lock_guard<std::mutex> lock(C.mutex);
    if (!C.ready) {
        LSD_HIP(hipStreamCreateWithFlags(&C.copy, hipStreamNonBlocking));
        LSD_HIP(hipStreamCreateWithFlags(&C.compute, hipStreamNonBlocking));
        for (int i = 0; i < kMaxFeedEvents; i++) LSD_HIP(hipEventCreateWithFlags(&C.events[i], hipEventDisableTiming));
        C.ready = true;
    }
    LSD_TRY(grow(reinterpret_cast<void**>(&C.d_keys), &C.keys_bytes, bytes));
    if (pairs) LSD_TRY(grow(reinterpret_cast<void**>(&C.d_vals), &C.vals_bytes, bytes));
    LSD_TRY(grow(&C.d_ws, &C.ws_bytes, ws_bytes));
    HostFeed feed;
    feed.host_keys = keys;
    feed.host_vals = vals;
    feed.copy = C.copy;
    feed.events = C.events;
    // H2D in chunks with stage 1 behind each (.cu:1001), the passes (.cu:1003), the copy back (.cu:1005)
    const int status = run_sort(C.d_keys, pairs ? C.d_vals : nullptr, C.d_ws, C.ws_bytes, n, radix_bits, LSDSORT_ALGO_ONESWEEP, C.compute,
                                nullptr, nullptr, lsd::KeyTransform{}, &feed);
    if (status != LSDSORT_OK) {
        (void)hipStreamSynchronize(C.copy);       // nothing of the caller's array may still be in flight when we return
        (void)hipStreamSynchronize(C.compute);
        return status;
    }
    LSD_TRY(read_fault(C.d_ws, C.compute));       // synchronises the sort
    LSD_HIP(hipMemcpy(keys, C.d_keys, bytes, hipMemcpyDeviceToHost));
    if (pairs) LSD_HIP(hipMemcpy(vals, C.d_vals, bytes, hipMemcpyDeviceToHost));
    return LSDSORT_OK;
}

struct MsbLayout {
    size_t control = 0, tickets = 256, counts = 512, status = 768, zero_bytes = 0, table = 0, total = 0;
    uint32_t rows = 0;
};

MsbLayout make_msb_layout(size_t n, int msb_bits)
{
    MsbLayout L;
    const int r = msb_bits ? msb_bits : 1;
    // sized for the smallest tile a partition of up to n keys may use (most rows): the default shape of the stage entries
    // and every size class's
    size_t tile = (size_t)current_shape(r)->tile();
    {
        const TileShape* shapes = nullptr;
        const int count = lsd::tile_shapes(r, &shapes);
        for (int c = 0; c < kNumShapeClasses; c++) {
            const int id = class_shape(kShapeClasses[c], r);
            if (id < count && (size_t)shapes[id].tile() < tile) tile = (size_t)shapes[id].tile();
        }
    }
    L.rows = (uint32_t)((n + tile - 1) / tile) + 1u;
    L.zero_bytes = align_up(L.status + (size_t)L.rows * ((size_t)1 << r) * sizeof(uint32_t));
    L.table = L.zero_bytes;
    L.total = align_up(L.table + lsd::region_table_words(r) * sizeof(uint32_t));
    return L;
}

}  // namespace

// internals other translation units of the library use (sharded.hip)
namespace lsd {
int sort_host_multi(uint32_t* keys, size_t n, int radix_bits, int num_gpus, bool loopback);   // sharded.hip
void set_last_hip_error(hipError_t e) { g_last_hip = e; }
}  // namespace lsd

// =============================================================================== C-ABI
extern "C" {

const char* lsdsort_strerror(int status)
{
    switch (status) {
        case LSDSORT_OK: return "ok";
        case LSDSORT_ERR_INVALID_ARG: return "invalid argument";
        case LSDSORT_ERR_NO_DEVICE: return "no usable gfx950 HIP device (there is no CPU path behind this library)";
        case LSDSORT_ERR_HIP: return "HIP runtime error (see lsdsort_last_hip_error_string)";
        case LSDSORT_ERR_WORKSPACE: return "workspace null, not 256-byte aligned, or too small";
        case LSDSORT_ERR_TOO_LARGE: return "n exceeds LSDSORT_MAX_KEYS";
        case LSDSORT_ERR_UNSUPPORTED: return "unsupported request (typed keys need radix 4 or 8; multi-GPU needs librccl)";
        case LSDSORT_ERR_DEVICE_FAULT: return "a kernel gave up a bounded wait or refused destinations outside the output (inconsistent counts); output undefined";
        case LSDSORT_ERR_COMM: return "a collective failed or a peer left it (see lsdsort_last_comm_error)";
        case LSDSORT_ERR_CAPACITY: return "a rank would receive more keys than its out_capacity (collective: every rank gets this, nothing was exchanged)";
        default: return "unknown lsdsort status";
    }
}

int lsdsort_last_hip_error(void) { return (int)g_last_hip; }
const char* lsdsort_last_hip_error_string(void) { return hipGetErrorString(g_last_hip); }

const char* lsdsort_version(void)
{
    static char text[96];
    static bool init = false;
    if (!init) {
        int rt = 0;
        if (hipRuntimeGetVersion(&rt) != hipSuccess) { (void)hipGetLastError(); rt = 0; }
        std::snprintf(text, sizeof(text), "lsdsort 0.1.0 gfx950 hip %d", rt);
        init = true;
    }
    return text;
}

int lsdsort_device_count(void)
{
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    int usable = 0;
    for (int d = 0; d < count; d++) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, d) == hipSuccess && std::strncmp(prop.gcnArchName, "gfx950", 6) == 0) usable++;
    }
    (void)hipGetLastError();
    return usable;
}

int lsdsort_set_tile_config(int radix_bits, int config_id)
{
    if (radix_bits < 1 || radix_bits > 8) return LSDSORT_ERR_INVALID_ARG;
    const TileShape* shapes = nullptr;
    const int count = lsd::tile_shapes(radix_bits, &shapes);
    if (count == 0) return LSDSORT_ERR_INVALID_ARG;
    if (config_id >= count) return LSDSORT_ERR_INVALID_ARG;
    g_shape_override[radix_bits].store(config_id < 0 ? 0 : config_id + 1, std::memory_order_relaxed);
    return LSDSORT_OK;
}

int lsdsort_set_xcd_chunk(int chunk)
{
    if (chunk < 0 || chunk > (int)kMaxXcdChunk) return LSDSORT_ERR_INVALID_ARG;
    g_xcd_chunk.store((uint32_t)chunk, std::memory_order_relaxed);
    return LSDSORT_OK;
}

int lsdsort_workspace_form(const void* d_workspace, void* hip_stream, int* hybrid)
{
    if (!d_workspace || !hybrid) return LSDSORT_ERR_INVALID_ARG;
    // the opening memset of every sort clears the word; only the hybrid planner sets it
    uint32_t ok = 0;
    LSD_HIP(hipMemcpyAsync(&ok, static_cast<const uint32_t*>(d_workspace) + kHybridOffsetWords + lsd::kHybridWordOk, sizeof(ok),
                           hipMemcpyDeviceToHost, static_cast<hipStream_t>(hip_stream)));
    LSD_HIP(hipStreamSynchronize(static_cast<hipStream_t>(hip_stream)));
    *hybrid = ok ? 1 : 0;
    return LSDSORT_OK;
}

int lsdsort_set_small_sort(int on)
{
    g_small_sort.store(on ? 1 : 0, std::memory_order_relaxed);
    return LSDSORT_OK;
}

int lsdsort_set_hybrid(int on)
{
    g_hybrid.store(on ? 1 : 0, std::memory_order_relaxed);
    return LSDSORT_OK;
}

int lsdsort_set_pass_skipping(int on)
{
    g_skip_dead_passes.store(on ? 1 : 0, std::memory_order_relaxed);
    return LSDSORT_OK;
}

#ifdef LSD_FAULT_INJECT
// Diagnostic build only (make faultinject -> liblsdsort_faultinject.so, never the product): a small spin limit and a
// status row that never publishes, so that a test can drive the bounded-wait expiry path (tests/test_fault_path.py).
LSDSORT_API int lsdsort_debug_fault_inject(unsigned spin_limit, unsigned mute_row_plus_1)
{
    g_spin_limit.store(spin_limit ? spin_limit : lsd::kSpinLimit, std::memory_order_relaxed);
    g_mute_row.store(mute_row_plus_1, std::memory_order_relaxed);
    return LSDSORT_OK;
}
#endif

#ifdef LSD_FAULT_INJECT
LSDSORT_API int lsdsort_debug_fault_inject_sorts(int sorts)   // the muted row applies to the next `sorts` sorts only (-1: to all)
{
    g_mute_sorts.store(sorts, std::memory_order_relaxed);
    return LSDSORT_OK;
}

LSDSORT_API int lsdsort_debug_corrupt_counts(unsigned from_word, unsigned to_word, unsigned delta, unsigned keep_sum)
{
    CorruptCounts c;
    c.from = from_word;
    c.to = to_word;
    c.delta = delta;
    c.keep_sum = keep_sum;
    std::lock_guard<std::mutex> lock(g_corrupt_mutex);
    g_corrupt = c;
    return LSDSORT_OK;
}
#endif

#ifdef LSD_PHASE_STATS
LSDSORT_API int lsdsort_debug_set_stats(unsigned long long* d_stats)
{
    g_stats.store(d_stats, std::memory_order_relaxed);
    return LSDSORT_OK;
}
#endif

int lsdsort_release_host_cache(void)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) {
        (void)hipGetLastError();
        return LSDSORT_ERR_NO_DEVICE;
    }
    HostCache& C = g_host_cache[dev];
    std::lock_guard<std::mutex> lock(C.mutex);
    if (C.d_keys) (void)hipFree(C.d_keys);
    if (C.d_vals) (void)hipFree(C.d_vals);
    if (C.d_ws) (void)hipFree(C.d_ws);
    C.d_keys = C.d_vals = nullptr;
    C.d_ws = nullptr;
    C.keys_bytes = C.vals_bytes = C.ws_bytes = 0;
    (void)hipGetLastError();
    return LSDSORT_OK;
}

int lsdsort_prepare_device(void)
{
    return check_device_ready();
}

int lsdsort_set_rank_method(int method)
{
    if (method != -1 && method != 0 && method != 2) return LSDSORT_ERR_INVALID_ARG;
    g_rank_setting.store(method, std::memory_order_relaxed);
    return LSDSORT_OK;
}

int lsdsort_rank_method(int radix_bits)
{
    if (radix_bits < 1 || radix_bits > 8) return LSDSORT_ERR_INVALID_ARG;
    int dev = 0;
    LSD_TRY(check_device_ready(&dev));
    return resolve_rank_method(dev, radix_bits);
}

size_t lsdsort_tile_keys(int radix_bits)
{
    const TileShape* shape = (radix_bits >= 1 && radix_bits <= 8) ? current_shape(radix_bits) : nullptr;
    return shape ? (size_t)shape->tile() : 0;
}

size_t lsdsort_workspace_bytes_ex(size_t n, int radix_bits, int pairs, int algorithm)
{
    if (!valid_radix(radix_bits) || n > LSDSORT_MAX_KEYS) return 0;
    if (algorithm != LSDSORT_ALGO_ONESWEEP && algorithm != LSDSORT_ALGO_STAGED) return 0;
    // Sized for every shape a sort of up to n keys may pick -- its own size class, and each smaller
    // class at that class's largest n (smaller tiles mean more status rows per key) -- so that a
    // workspace made for n serves every smaller sort as well; each term, hence the figure, is
    // monotonic in n.
    if (pairs < 0 || pairs > 3) return 0;
    size_t need = make_layout(n, radix_bits, pairs, algorithm, *current_shape(radix_bits, pairs != 0, n, algorithm)).total;
    if (algorithm == LSDSORT_ALGO_ONESWEEP) {
        for (int c = 0; c < kNumShapeClasses && kShapeClasses[c].below <= n; c++) {
            const size_t m = kShapeClasses[c].below - 1;
            const size_t t = make_layout(m, radix_bits, pairs, algorithm, *current_shape(radix_bits, pairs != 0, m, algorithm)).total;
            if (t > need) need = t;
        }
    }
    const size_t stage = make_layout(n, radix_bits, pairs, algorithm, *current_shape(radix_bits, pairs != 0, 0, algorithm)).total;
    return need > stage ? need : stage;
}

size_t lsdsort_workspace_bytes(size_t n, int radix_bits, int pairs)
{
    return lsdsort_workspace_bytes_ex(n, radix_bits, pairs, LSDSORT_ALGO_ONESWEEP);
}

int lsdsort_u32_device_ex(uint32_t* d_keys, uint32_t* d_vals, void* d_workspace, size_t workspace_bytes, size_t n,
                          int radix_bits, int algorithm, void* hip_stream)
{
    return run_sort(d_keys, d_vals, d_workspace, workspace_bytes, n, radix_bits, algorithm,
                    static_cast<hipStream_t>(hip_stream), nullptr, nullptr);
}

int lsdsort_u32_device(uint32_t* d_keys, void* d_workspace, size_t workspace_bytes, size_t n, int radix_bits,
                       void* hip_stream)
{
    return lsdsort_u32_device_ex(d_keys, nullptr, d_workspace, workspace_bytes, n, radix_bits, LSDSORT_ALGO_ONESWEEP,
                                 hip_stream);
}

int lsdsort_u32_device_prefixed(uint32_t* d_keys, void* d_workspace, size_t workspace_bytes, size_t n, int radix_bits,
                                int common_prefix_bits, void* hip_stream)
{
    return run_sort(d_keys, nullptr, d_workspace, workspace_bytes, n, radix_bits, LSDSORT_ALGO_ONESWEEP, static_cast<hipStream_t>(hip_stream),
                    nullptr, nullptr, lsd::KeyTransform{}, nullptr, nullptr, 0, common_prefix_bits);
}

int lsdsort_pairs_u32_device(uint32_t* d_keys, uint32_t* d_vals, void* d_workspace, size_t workspace_bytes, size_t n,
                             int radix_bits, void* hip_stream)
{
    if (n > 0 && !d_vals) return LSDSORT_ERR_INVALID_ARG;
    return lsdsort_u32_device_ex(d_keys, d_vals, d_workspace, workspace_bytes, n, radix_bits, LSDSORT_ALGO_ONESWEEP,
                                 hip_stream);
}

int lsdsort_multi_u32_device(uint32_t* d_keys, uint32_t* const* d_vals, int num_vals, void* d_workspace, size_t workspace_bytes,
                             size_t n, int radix_bits, void* hip_stream)
{
    if (num_vals < 1 || num_vals > 3 || !d_vals) return LSDSORT_ERR_INVALID_ARG;
    if (n > 0 && !d_vals[0]) return LSDSORT_ERR_INVALID_ARG;
    // the key/value kernel sends further payload arrays through its single-round shapes only (every default shape is one)
    const TileShape* shape = current_shape(radix_bits, true, n, LSDSORT_ALGO_ONESWEEP);
    if (num_vals > 1 && shape && valid_radix(radix_bits)) {
        const TileShape* shapes = nullptr;
        (void)lsd::tile_shapes(radix_bits, &shapes);
        if (!lsd::single_round_shape(radix_bits, (int)(shape - shapes))) return LSDSORT_ERR_UNSUPPORTED;
    }
    return run_sort(d_keys, d_vals[0], d_workspace, workspace_bytes, n, radix_bits, LSDSORT_ALGO_ONESWEEP,
                    static_cast<hipStream_t>(hip_stream), nullptr, nullptr, lsd::KeyTransform{}, nullptr, d_vals + 1, num_vals - 1);
}

int lsdsort_keys_device(void* d_keys, uint32_t* d_vals, void* d_workspace, size_t workspace_bytes, size_t n,
                        int radix_bits, int key_type, int descending, void* hip_stream)
{
    lsd::KeyTransform xf{};
    switch (key_type) {
        case LSDSORT_KEY_U32: break;
        case LSDSORT_KEY_I32: xf.b = 0x80000000u; break;
        case LSDSORT_KEY_F32: xf.a = 0x80000000u; xf.b = 0x80000000u; break;
        default: return LSDSORT_ERR_INVALID_ARG;
    }
    if (descending) xf.c = 0xFFFFFFFFu;
    xf.on = (xf.a | xf.b | xf.c) != 0u;
    return run_sort(static_cast<uint32_t*>(d_keys), d_vals, d_workspace, workspace_bytes, n, radix_bits, LSDSORT_ALGO_ONESWEEP,
                    static_cast<hipStream_t>(hip_stream), nullptr, nullptr, xf);
}

int lsdsort_check_device(void* d_workspace, void* hip_stream)
{
    if (!d_workspace) return LSDSORT_ERR_WORKSPACE;
    int dev = 0;
    LSD_TRY(check_device_ready(&dev));
    const int status = read_fault(d_workspace, static_cast<hipStream_t>(hip_stream));
    // The default rank form rests on a probed hardware property (lane-ordered returning LDS adds, rank_scatter.hpp); the probe
    // runs once per device.  Diagnostic builds, and any build with LSDSORT_REPROBE=1 in the environment, run it AGAIN whenever a
    // caller checks a sort: a device that stops honouring the property is reported here (and the library falls back to the mask
    // forms from then on) instead of silently mis-ordering equal digits.
#ifdef LSD_FAULT_INJECT
    constexpr bool always = true;
#else
    constexpr bool always = false;
#endif
    static const bool by_env = [] {
        const char* e = getenv("LSDSORT_REPROBE");
        return e && e[0] == '1';
    }();
    if (status == LSDSORT_OK && (always || by_env) && g_device[dev].lds_add_in_lane_order) {
        bool ok = false;
        LSD_HIP(lsd::probe_lds_add_lane_order(&ok, static_cast<hipStream_t>(hip_stream)));
        if (!ok) {
            g_device[dev].lds_add_in_lane_order = false;
            return LSDSORT_ERR_DEVICE_FAULT;
        }
    }
    return status;
}

int lsdsort_u32_device_timed(uint32_t* d_keys, uint32_t* d_vals, void* d_workspace, size_t workspace_bytes, size_t n,
                             int radix_bits, int algorithm, void* hip_stream, lsdsort_timing* out)
{
    if (!out) return LSDSORT_ERR_INVALID_ARG;
    std::memset(out, 0, sizeof(*out));
    StageEvents ev;
    ev.enabled = true;
    ev.stream = static_cast<hipStream_t>(hip_stream);
    int status = run_sort(d_keys, d_vals, d_workspace, workspace_bytes, n, radix_bits, algorithm, ev.stream, &ev, out);
    if (status == LSDSORT_OK && ev.count >= 4) {
        auto finish = [&]() -> int {
            LSD_HIP(hipEventSynchronize(ev.ev[ev.count - 1]));
            LSD_HIP(hipEventElapsedTime(&out->total_ms, ev.ev[0], ev.ev[ev.count - 1]));
            LSD_HIP(hipEventElapsedTime(&out->clear_ms, ev.ev[0], ev.ev[1]));
            LSD_HIP(hipEventElapsedTime(&out->histogram_ms, ev.ev[1], ev.ev[2]));
            LSD_HIP(hipEventElapsedTime(&out->scan_ms, ev.ev[2], ev.ev[3]));
            if (algorithm == LSDSORT_ALGO_ONESWEEP) {
                // marks: 3 = before the first pass, the last = after the last; per pass: the kernel's own events.  Where the hybrid
                // form was tried its global passes come first (event pairs 0, 1 and at 4-bit digits 2, 3) and marks 4, 5 bracket the local stage.
                int first_pair = 0;
                if (out->hybrid == -1) {
                    uint32_t ok = 0;
                    LSD_HIP(hipMemcpy(&ok, static_cast<const uint32_t*>(d_workspace) + kHybridOffsetWords + lsd::kHybridWordOk, sizeof(ok),
                                      hipMemcpyDeviceToHost));
                    out->hybrid = ok ? 1 : 0;
                    first_pair = ok ? 0 : lsd::hybrid_global_passes(radix_bits);
                    if (ok) {
                        out->passes = lsd::hybrid_global_passes(radix_bits);
                        if (ev.count >= 6) LSD_HIP(hipEventElapsedTime(&out->local_ms, ev.ev[4], ev.ev[5]));
                    }
                }
                for (int p = 0; p < out->passes && p < LSDSORT_MAX_PASSES && 2 * (p + first_pair) + 1 < ev.kernel_count; p++)
                    LSD_HIP(hipEventElapsedTime(&out->scatter_ms[p], ev.kernel_ev[2 * (p + first_pair)], ev.kernel_ev[2 * (p + first_pair) + 1]));
            } else {
                out->histogram_ms = 0.f;
                out->scan_ms = 0.f;
                for (int p = 0; 3 * p + 6 < ev.count && p < LSDSORT_MAX_PASSES; p++) {
                    float h = 0.f, s = 0.f;
                    LSD_HIP(hipEventElapsedTime(&h, ev.ev[3 + 3 * p], ev.ev[4 + 3 * p]));
                    LSD_HIP(hipEventElapsedTime(&s, ev.ev[4 + 3 * p], ev.ev[5 + 3 * p]));
                    LSD_HIP(hipEventElapsedTime(&out->scatter_ms[p], ev.ev[5 + 3 * p], ev.ev[6 + 3 * p]));
                    out->histogram_ms += h;
                    out->scan_ms += s;
                }
            }
            return LSDSORT_OK;
        };
        status = finish();
    }
    ev.destroy();
    return status;
}

int lsdsort_u32_ex(uint32_t* keys, size_t n, int radix_bits, int num_gpus)
{
    if (num_gpus == 0) return LSDSORT_ERR_NO_DEVICE;     // the CPU path is the oracle, not the product
    if (num_gpus < 0) return LSDSORT_ERR_INVALID_ARG;
    if (num_gpus > 1) return lsd::sort_host_multi(keys, n, radix_bits, num_gpus, false);   // sharded.hip: one thread per device, RCCL
    return sort_host(keys, nullptr, n, radix_bits);
}

int lsdsort_u32_loopback(uint32_t* keys, size_t n, int radix_bits, int virtual_gpus)
{
    if (virtual_gpus == 1) return sort_host(keys, nullptr, n, radix_bits);
    return lsd::sort_host_multi(keys, n, radix_bits, virtual_gpus, true);
}

int lsdsort_u32(uint32_t* keys, size_t n) { return lsdsort_u32_ex(keys, n, 8, 1); }

int lsdsort_pairs_u32(uint32_t* keys, uint32_t* vals, size_t n)
{
    if (n > 0 && !vals) return LSDSORT_ERR_INVALID_ARG;
    return sort_host(keys, vals, n, 8);
}

// ---- stage entries ---------------------------------------------------------------------------
int lsdsort_tile_histograms_u32_device(const uint32_t* d_keys, size_t n, int radix_bits, int bit_group,
                                       uint32_t* d_hist, void* hip_stream)
{
    if (!valid_radix(radix_bits) || bit_group < 0 || (bit_group + 1) * radix_bits > 32) return LSDSORT_ERR_INVALID_ARG;
    if (n > LSDSORT_MAX_KEYS) return LSDSORT_ERR_TOO_LARGE;
    if (n == 0) return LSDSORT_OK;
    if (!d_keys || !d_hist) return LSDSORT_ERR_INVALID_ARG;
    LSD_TRY(check_device_ready());
    LSD_HIP(lsd::launch_tile_histograms(radix_bits, *current_shape(radix_bits), d_keys, (uint32_t)n,
                                        (uint32_t)(bit_group * radix_bits), d_hist, static_cast<hipStream_t>(hip_stream)));
    return LSDSORT_OK;
}

size_t lsdsort_tile_offsets_scratch_bytes(size_t tiles, int radix_bits)
{
    if (!valid_radix(radix_bits)) return 0;
    return align_up(lsd::tile_offsets_scratch_words(tiles, radix_bits) * sizeof(uint32_t));
}

int lsdsort_tile_offsets_u32_device(const uint32_t* d_hist, uint32_t* d_local, uint32_t* d_global, size_t tiles,
                                    int radix_bits, void* d_scratch, void* hip_stream)
{
    if (!valid_radix(radix_bits)) return LSDSORT_ERR_INVALID_ARG;
    if (tiles == 0) return LSDSORT_OK;
    if (tiles > 0xffffffffu || !d_hist || (d_global && !d_scratch)) return LSDSORT_ERR_INVALID_ARG;
    LSD_TRY(check_device_ready());
    LSD_HIP(lsd::launch_tile_offsets(radix_bits, d_hist, d_local, d_global, (uint32_t)tiles,
                                     static_cast<uint32_t*>(d_scratch), static_cast<hipStream_t>(hip_stream)));
    return LSDSORT_OK;
}

int lsdsort_rank_scatter_u32_device(const uint32_t* d_in, uint32_t* d_out, const uint32_t* d_vals_in,
                                    uint32_t* d_vals_out, const uint32_t* d_global, size_t n, int radix_bits,
                                    int bit_group, void* hip_stream)
{
    if (!valid_radix(radix_bits) || bit_group < 0 || (bit_group + 1) * radix_bits > 32) return LSDSORT_ERR_INVALID_ARG;
    if (n > LSDSORT_MAX_KEYS) return LSDSORT_ERR_TOO_LARGE;
    if (n == 0) return LSDSORT_OK;
    if (!d_in || !d_out || !d_global || ((d_vals_in == nullptr) != (d_vals_out == nullptr))) return LSDSORT_ERR_INVALID_ARG;
    int dev = 0;
    LSD_TRY(check_device_ready(&dev));
    const TileShape* shape = current_shape(radix_bits);
    PassParams p{};
    p.in = d_in;
    p.out = d_out;
    p.vals_in = d_vals_in;
    p.vals_out = d_vals_out;
    p.n = (uint32_t)n;
    p.shift = (uint32_t)(bit_group * radix_bits);
    p.num_tiles = (uint32_t)((n + shape->tile() - 1) / shape->tile());
    p.global_off = d_global;
    LSD_HIP(lsd::launch_rank_scatter(radix_bits, *shape, resolve_rank_method(dev, radix_bits), false, p,
                                     static_cast<hipStream_t>(hip_stream)));
    return LSDSORT_OK;
}

int lsdsort_local_sort_u32_device(uint32_t* d_keys, uint32_t* d_vals, const uint32_t* d_bases, size_t num_buckets, int low_bits,
                                  void* hip_stream)
{
    if (low_bits < 1 || low_bits > 27 || num_buckets > 0xffffffffu) return LSDSORT_ERR_INVALID_ARG;
    if (num_buckets == 0) return LSDSORT_OK;
    if (!d_keys || !d_bases) return LSDSORT_ERR_INVALID_ARG;
    int dev = 0;
    LSD_TRY(check_device_ready(&dev));
    if (!g_device[dev].lds_add_in_lane_order) return LSDSORT_ERR_UNSUPPORTED;   // the local stage ranks by returning LDS adds only
    lsd::LocalSortParams p{};
    p.keys = d_keys;
    p.vals = d_vals;
    p.bases = d_bases;
    p.num_buckets = (uint32_t)num_buckets;
    const int passes = (low_bits + 8) / 9;                     // digits of at most nine bits, as even as they come
    int shift = 0;
    for (int i = 0; i < passes; i++) {
        const int width = (low_bits - shift + (passes - i) - 1) / (passes - i);
        p.shift[i] = (uint32_t)shift;
        p.width[i] = (uint32_t)width;
        shift += width;
    }
    LSD_HIP(lsd::launch_local_sort(p, static_cast<hipStream_t>(hip_stream)));
    return LSDSORT_OK;
}

int lsdsort_digit_histograms_u32_device(const uint32_t* d_keys, size_t n, int radix_bits, uint32_t* d_hist,
                                        void* hip_stream)
{
    if (!valid_radix(radix_bits)) return LSDSORT_ERR_INVALID_ARG;
    if (n > LSDSORT_MAX_KEYS) return LSDSORT_ERR_TOO_LARGE;
    if (!d_hist || (n > 0 && !d_keys)) return LSDSORT_ERR_INVALID_ARG;
    LSD_TRY(check_device_ready());
    hipStream_t stream = static_cast<hipStream_t>(hip_stream);
    const int groups = 32 / radix_bits;
    LSD_HIP(hipMemsetAsync(d_hist, 0, (size_t)groups * ((size_t)1 << radix_bits) * sizeof(uint32_t), stream));
    if (n == 0) return LSDSORT_OK;
    LSD_HIP(lsd::launch_digit_histograms(radix_bits, groups, 0, d_keys, (uint32_t)n, d_hist, stream));
    return LSDSORT_OK;
}

// ---- multi-GPU building block ----------------------------------------------------------------
size_t lsdsort_msb_partition_workspace_bytes(size_t n, int msb_bits)
{
    if (msb_bits < 0 || msb_bits > 4 || n > LSDSORT_MAX_KEYS) return 0;
    return make_msb_layout(n, msb_bits).total;
}

// Stable partition into 2^msb_bits buckets: by the top msb_bits key bits (splitters == nullptr) or by
// 2^msb_bits - 1 ascending splitters (bucket = number of splitters <= key), of which the first `live` are
// given (live < 0: all of them); the others lie above every key.
static int partition_impl(const uint32_t* d_in, uint32_t* d_out, size_t n, int msb_bits, const uint32_t* splitters,
                          uint64_t* d_counts, void* d_workspace, size_t workspace_bytes, void* hip_stream,
                          hipEvent_t counts_ready = nullptr, int live = -1)
{
    // by bit field: up to sixteen buckets (the 4-bit kernels); by value: eight (the splitters travel in the launch)
    if (msb_bits < 0 || msb_bits > (splitters ? 3 : 4) || !d_counts) return LSDSORT_ERR_INVALID_ARG;
    if (live < 0) live = (1 << msb_bits) - 1;
    if (live > (1 << msb_bits) - 1) return LSDSORT_ERR_INVALID_ARG;
    if (splitters)
        for (int i = 1; i < live; i++)
            if (splitters[i] < splitters[i - 1]) return LSDSORT_ERR_INVALID_ARG;
    if (n > LSDSORT_MAX_KEYS) return LSDSORT_ERR_TOO_LARGE;
    if (n > 0 && (!d_in || !d_out || d_in == d_out)) return LSDSORT_ERR_INVALID_ARG;
    int dev = 0;
    LSD_TRY(check_device_ready(&dev));
    hipStream_t stream = static_cast<hipStream_t>(hip_stream);
    const MsbLayout L = make_msb_layout(n, msb_bits);
    if (!d_workspace || (reinterpret_cast<uintptr_t>(d_workspace) & (kAlign - 1)) || workspace_bytes < L.total)
        return LSDSORT_ERR_WORKSPACE;
    char* ws = static_cast<char*>(d_workspace);
    uint32_t* control = reinterpret_cast<uint32_t*>(ws + L.control);
    uint32_t* hist = reinterpret_cast<uint32_t*>(ws + L.counts);
    uint32_t* table = reinterpret_cast<uint32_t*>(ws + L.table);
    const int bins = 1 << msb_bits;
    LSD_HIP(hipMemsetAsync(ws, 0, L.zero_bytes, stream));
    if (msb_bits == 0) {
        // one bucket: the shard itself
        LSD_HIP(lsd::launch_store_u64(d_counts, (uint64_t)n, stream));
        if (counts_ready) LSD_HIP(hipEventRecord(counts_ready, stream));
        if (n) LSD_HIP(hipMemcpyAsync(d_out, d_in, n * sizeof(uint32_t), hipMemcpyDeviceToDevice, stream));
        return LSDSORT_OK;
    }
    if (n) {
        const uint32_t shift = (uint32_t)(32 - msb_bits);
        const TileShape* shape = current_shape(msb_bits, false, n, LSDSORT_ALGO_ONESWEEP);
        if (splitters)
            LSD_HIP(lsd::launch_bucket_histogram(msb_bits, splitters, live, d_in, (uint32_t)n, hist, stream));
        else
            LSD_HIP(lsd::launch_digit_histograms(msb_bits, 1, shift, d_in, (uint32_t)n, hist, stream));
        LSD_HIP(lsd::launch_scan_regions(msb_bits, 1, 1, hist, (uint32_t)n, (uint32_t)shape->tile(), 0, table, stream, nullptr, control));
        // the bucket sizes are final here: hand them out before the partition pass, so that a caller's count
        // exchange (multi-GPU step, sharded.hip) runs beside it
        LSD_HIP(lsd::launch_widen_counts(hist, d_counts, bins, stream));
        if (counts_ready) LSD_HIP(hipEventRecord(counts_ready, stream));
        PassParams p{};
        p.in = d_in;
        p.out = d_out;
        p.n = (uint32_t)n;
        p.shift = shift;
        p.num_tiles = (uint32_t)((n + (size_t)shape->tile() - 1) / (size_t)shape->tile()) + 1u;   // <= L.rows
        p.regions = table;
        p.status = reinterpret_cast<uint32_t*>(ws + L.status);
        p.tickets = reinterpret_cast<uint32_t*>(ws + L.tickets);
        p.parity = 0;
        p.fault = control;
        p.spin_limit = g_spin_limit.load(std::memory_order_relaxed);
        if (splitters) {
            p.num_splitters = (1u << msb_bits) - 1u;
            p.live_splitters = (uint32_t)live;
            for (uint32_t i = 0; i < p.live_splitters; i++) p.splitters[i] = splitters[i];
        }
        LSD_HIP(lsd::launch_rank_scatter(msb_bits, *shape, resolve_rank_method(dev, msb_bits), true, p, stream));
    } else {
        LSD_HIP(lsd::launch_widen_counts(hist, d_counts, bins, stream));   // all zero
        if (counts_ready) LSD_HIP(hipEventRecord(counts_ready, stream));
    }
    return LSDSORT_OK;
}

}  // extern "C"
namespace lsd {
// the MSB partition with its bucket counts published (and `counts_ready` recorded) BEFORE the partition pass
int partition_with_event(const uint32_t* d_in, uint32_t* d_out, size_t n, int msb_bits, uint64_t* d_counts,
                         void* d_workspace, size_t workspace_bytes, hipStream_t stream, hipEvent_t counts_ready)
{
    return partition_impl(d_in, d_out, n, msb_bits, nullptr, d_counts, d_workspace, workspace_bytes, stream, counts_ready);
}
// the same by thresholds: bucket of a key = number of thresholds <= key; ascending values in [0, 2^32], 2^32 = above every key
int threshold_partition_with_event(const uint32_t* d_in, uint32_t* d_out, size_t n, int log2_buckets, const uint64_t* thresholds,
                                   uint64_t* d_counts, void* d_workspace, size_t workspace_bytes, hipStream_t stream,
                                   hipEvent_t counts_ready)
{
    if (log2_buckets < 0 || log2_buckets > 3) return LSDSORT_ERR_INVALID_ARG;
    if (log2_buckets == 0)
        return partition_impl(d_in, d_out, n, 0, nullptr, d_counts, d_workspace, workspace_bytes, stream, counts_ready);
    if (!thresholds) return LSDSORT_ERR_INVALID_ARG;
    uint32_t values[7] = {};
    int live = 0;
    const int count = (1 << log2_buckets) - 1;
    for (int i = 0; i < count; i++) {
        if (thresholds[i] > (1ull << 32) || (i > 0 && thresholds[i] < thresholds[i - 1])) return LSDSORT_ERR_INVALID_ARG;
        if (thresholds[i] < (1ull << 32)) values[live++] = (uint32_t)thresholds[i];
    }
    return partition_impl(d_in, d_out, n, log2_buckets, values, d_counts, d_workspace, workspace_bytes, stream, counts_ready, live);
}
}  // namespace lsd
extern "C" {

int lsdsort_msb_partition_u32_device(const uint32_t* d_in, uint32_t* d_out, size_t n, int msb_bits,
                                     uint64_t* d_counts, void* d_workspace, size_t workspace_bytes, void* hip_stream)
{
    return partition_impl(d_in, d_out, n, msb_bits, nullptr, d_counts, d_workspace, workspace_bytes, hip_stream);
}

int lsdsort_splitter_partition_u32_device(const uint32_t* d_in, uint32_t* d_out, size_t n, int log2_buckets,
                                          const uint32_t* splitters, uint64_t* d_counts, void* d_workspace,
                                          size_t workspace_bytes, void* hip_stream)
{
    if (log2_buckets > 0 && !splitters) return LSDSORT_ERR_INVALID_ARG;
    return partition_impl(d_in, d_out, n, log2_buckets, log2_buckets > 0 ? splitters : nullptr, d_counts, d_workspace,
                          workspace_bytes, hip_stream);
}

int lsdsort_threshold_partition_u32_device(const uint32_t* d_in, uint32_t* d_out, size_t n, int log2_buckets,
                                           const uint64_t* thresholds, uint64_t* d_counts, void* d_workspace,
                                           size_t workspace_bytes, void* hip_stream)
{
    return lsd::threshold_partition_with_event(d_in, d_out, n, log2_buckets, thresholds, d_counts, d_workspace, workspace_bytes,
                                               static_cast<hipStream_t>(hip_stream), nullptr);
}

}  // extern "C"
