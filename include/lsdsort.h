/*
 * lsdsort.h -- C-ABI of the MI355X-native LSD radix sort (liblsdsort.so).
 *
 * Plain pointers and sizes only; no C++ or torch types.  Every entry returns 0
 * (LSDSORT_OK) or a negative lsdsort_status; nothing here aborts the process (the reference
 * crashes on any error: CUDA_CALL, LSDRadixSort/CudaUtils.h:7-8; MYASSERT, Utils.h:6-15).
 *
 * What each entry replaces in the reference (paths relative to /root/reference/, ".cu" =
 * LSDRadixSort/LSDRadixSort.cu) is cited at its declaration.  The reference has no
 * sort(uint32_t*, size_t) symbol; lsdsort_u32 is defined as "what
 * TestGPULSDRadixSort does between .cu:1001 and .cu:1005, minus RNG and checking".
 *
 * There is NO CPU fallback behind this ABI: without a usable gfx950 device every compute
 * entry returns LSDSORT_ERR_NO_DEVICE.  The CPU legs of the reference (std::sort .cu:97,
 * CPU LSD .cu:25-69) live in oracle/ as test infrastructure only.
 */
#ifndef LSDSORT_H
#define LSDSORT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(LSDSORT_BUILD)
#define LSDSORT_API __attribute__((visibility("default")))
#else
#define LSDSORT_API
#endif

typedef enum lsdsort_status {
    LSDSORT_OK = 0,
    LSDSORT_ERR_INVALID_ARG = -1,  /* null pointer with n > 0, bad radix_bits, bad enum          */
    LSDSORT_ERR_NO_DEVICE = -2,    /* no HIP device / not gfx950; also num_gpus == 0 (no CPU path) */
    LSDSORT_ERR_HIP = -3,          /* a HIP runtime call failed; see lsdsort_last_hip_error()    */
    LSDSORT_ERR_WORKSPACE = -4,    /* workspace null, misaligned or smaller than required        */
    LSDSORT_ERR_TOO_LARGE = -5,    /* n above LSDSORT_MAX_KEYS (a local argument check, never collective) */
    LSDSORT_ERR_UNSUPPORTED = -6,  /* valid request that cannot be served here (e.g. no librccl) */
    LSDSORT_ERR_DEVICE_FAULT = -7, /* a kernel gave up a bounded wait, or refused destinations outside  */
                                   /* the output because the counts do not describe the keys (see below) */
    LSDSORT_ERR_COMM = -8,         /* a collective failed or a peer left it; lsdsort_last_comm_error() */
    LSDSORT_ERR_CAPACITY = -9      /* sharded step: some rank's out_capacity is too small; EVERY rank  */
                                   /* returns this before anything is exchanged (retry with *n_out)    */
} lsdsort_status;

/* Largest n any entry accepts: the chained tile prefix keeps 30 value bits per word.  The
 * reference stops at int count < 2^31 (.cu:839) and in practice at 2^30 (.cu:1033-1042). */
#define LSDSORT_MAX_KEYS ((size_t)0x3fffffffu)

/* Which pass structure runs on the device. */
typedef enum lsdsort_algorithm {
    /* default: one read of all keys for every digit histogram, one scan of the digit counts,
     * then per pass ONE rank-and-scatter kernel whose per-tile bases come from a chained
     * (decoupled look-back) scan over tiles.  4*(2P+1) bytes per key. */
    LSDSORT_ALGO_ONESWEEP = 0,
    /* the reference's own stage structure (GPULSDRadixSort, .cu:839-910): per pass
     * BuildHistogram -> local/global offsets -> rank-and-scatter as separate kernels with
     * the [tile][digit] tables in memory.  12 bytes per key per pass.  Same results. */
    LSDSORT_ALGO_STAGED = 1
} lsdsort_algorithm;

/* ---- host-pointer entries ------------------------------------------------------------- */

/* sort(uint32_t* keys, size_t n) of BASELINE.json's north_star: host pointer, in place,
 * ascending.  Replaces the alloc / H2D / GPULSDRadixSort / D2H sequence of
 * TestGPULSDRadixSort, .cu:966-1005.  Blocking.  radix 8, current HIP device.
 * The input crosses PCIe in 64 MiB chunks with the upfront histogram of each chunk running behind it; the
 * device buffers and workspace are kept per device between calls (the reference allocates and frees around
 * every sort) -- lsdsort_release_host_cache() frees them; calls on one device take turns. */
LSDSORT_API int lsdsort_u32(uint32_t* keys, size_t n);
LSDSORT_API int lsdsort_release_host_cache(void);

/* Same with the radix width (1, 2, 4 or 8 -- the reference's sweep `rs`, .cu:1055-1062) and
 * a GPU count: num_gpus in {1, 2, 4, 8}.  0 (the "CPU path") returns
 * LSDSORT_ERR_NO_DEVICE.  More than one GPU: this one process drives devices 0 .. num_gpus-1 (one host
 * thread per device, one RCCL communicator made with ncclCommInitAll and kept for later calls): the array
 * is cut into num_gpus shards, each device runs lsdsort_sharded_u32_device (below) on its shard, and the
 * slices come back in rank order.  LSDSORT_ERR_NO_DEVICE if fewer gfx950 devices are visible,
 * LSDSORT_ERR_UNSUPPORTED if librccl cannot be loaded.  Every device first sets itself up (buffers for its share plus a
 * quarter, upload); the ranks then AGREE on a status before the first collective, so a device that fails there takes the call
 * down with an error instead of leaving its peers in an all-gather; skewed keys repeat the step once with exact sizes.
 * Calls with num_gpus > 1 take turns (one set of communicators per device count).  NOT YET RUN ON MORE THAN ONE DEVICE:
 * the same code is exercised with virtual ranks on one GPU (lsdsort_u32_loopback).  The one-process-per-GPU form (bench.py, a
 * torch.distributed or MPI launcher) uses the lsdsort_comm_* entries directly. */
LSDSORT_API int lsdsort_u32_ex(uint32_t* keys, size_t n, int radix_bits, int num_gpus);
/* The num_gpus > 1 code of lsdsort_u32_ex with `virtual_gpus` (1, 2, 4, 8) VIRTUAL ranks on the current device (loopback
 * transport, lsdsort_comm_create_loopback below): threads, set-up agreement, capacity retry, the step and the copy back run
 * as they would on a multi-GPU node.  A rehearsal entry for one-GPU machines; the result is the sorted array all the same. */
LSDSORT_API int lsdsort_u32_loopback(uint32_t* keys, size_t n, int radix_bits, int virtual_gpus);

/* Key/value form, stable by key (BASELINE.json configs[4]); no reference counterpart. */
LSDSORT_API int lsdsort_pairs_u32(uint32_t* keys, uint32_t* vals, size_t n);

/* ---- device-resident entries (the timed path) ---------------------------------------- */
/* Threads: every call keeps its state in the workspace it is given, so calls with different workspaces may run
 * concurrently from different host threads and on different streams; two calls sharing a workspace must be
 * ordered by the caller (same stream, or an event).  The tuning setters (lsdsort_set_*) are process-wide. */

/* Bytes of device workspace the device entries need for (n, radix_bits, pairs, algorithm); `pairs` = the number of 32-bit
 * payload arrays that travel with the keys: 0 keys only, 1 key/value pairs, 2 or 3 for lsdsort_multi_u32_device.
 * Replaces the reference's d_b + d_h + d_block_sums sizing, .cu:919-930 and
 * GetGPUPrefixSumBlockSumsCount .cu:265-276.  Returns 0 for invalid arguments.  The figure is
 * monotonic in n and covers every tile shape the library may pick for up to n keys: a workspace
 * made for n serves every smaller sort with the same (radix_bits, pairs, algorithm). */
LSDSORT_API size_t lsdsort_workspace_bytes(size_t n, int radix_bits, int pairs);
LSDSORT_API size_t lsdsort_workspace_bytes_ex(size_t n, int radix_bits, int pairs, int algorithm);

/* Replaces GPULSDRadixSort(a, b, h, block_sums, d, grid, block, ...), .cu:839-910: device
 * pointers, result in d_keys (the reference's `a`; the pass count is even), stream-ordered
 * on hip_stream (a hipStream_t; NULL = the null stream), does not synchronise.  The
 * workspace (256-byte aligned) holds the ping-pong buffer (`b`) and every table (`h`,
 * `block_sums`); nothing is allocated or freed inside, so the call can be graph-captured. */
LSDSORT_API int lsdsort_u32_device(uint32_t* d_keys, void* d_workspace, size_t workspace_bytes,
                                   size_t n, int radix_bits, void* hip_stream);
LSDSORT_API int lsdsort_pairs_u32_device(uint32_t* d_keys, uint32_t* d_vals, void* d_workspace,
                                         size_t workspace_bytes, size_t n, int radix_bits,
                                         void* hip_stream);
/* Keys with up to THREE 32-bit payload arrays (d_vals[0 .. num_vals-1], num_vals 1..3), each permuted exactly like the
 * keys, stable: the building block of the record sorts below (a 64-bit payload = two arrays; the other word of a 64-bit
 * key = one more).  The key/value kernel sends the arrays through the same LDS slots one after the other: 8 + 8 num_vals
 * bytes per key per pass.  Workspace: lsdsort_workspace_bytes(n, radix_bits, num_vals).  No reference counterpart (.cu:62). */
LSDSORT_API int lsdsort_multi_u32_device(uint32_t* d_keys, uint32_t* const* d_vals, int num_vals, void* d_workspace,
                                         size_t workspace_bytes, size_t n, int radix_bits, void* hip_stream);
/* Same with the pass structure chosen explicitly; d_vals may be NULL (keys only). */
LSDSORT_API int lsdsort_u32_device_ex(uint32_t* d_keys, uint32_t* d_vals, void* d_workspace,
                                      size_t workspace_bytes, size_t n, int radix_bits,
                                      int algorithm, void* hip_stream);

/* Keys of another 32-bit type or order (no reference counterpart: it sorts ascending uint32 only,
 * .cu:62).  The keys are mapped to order-preserving uint32 where they are first read
 * (upfront histogram, first pass) and mapped back where the last pass stores them: same passes, same
 * traffic as lsdsort_u32_device.  float32 sorts in IEEE total order (-NaN < -inf < ... < -0 < +0 < ... <
 * +inf < +NaN).  Descending = ascending on the complemented key; with payloads it is stable (equal keys
 * keep their input order).  d_vals may be NULL.  Chained algorithm, radix_bits 4 or 8. */
typedef enum lsdsort_key_type { LSDSORT_KEY_U32 = 0, LSDSORT_KEY_I32 = 1, LSDSORT_KEY_F32 = 2 } lsdsort_key_type;
LSDSORT_API int lsdsort_keys_device(void* d_keys, uint32_t* d_vals, void* d_workspace, size_t workspace_bytes,
                                    size_t n, int radix_bits, int key_type, int descending, void* hip_stream);

/* 64-bit keys and 64-bit payloads (no reference counterpart, .cu:62), built on the 32-bit
 * pass kernels: an LSD sort on a 64-bit key is an LSD sort on its low word followed by a stable one on its
 * high word (lsdradixsort_amd/csrc/wide.hip).  Device pointers, in place, stream-ordered, nothing allocated.
 *   lsdsort_u64_device     : uint64 keys only.  Split into words, two key/value sorts (each word once the key,
 *                            once the payload), merge: 8 passes at radix 8, 16 B/key/pass.
 *   lsdsort_records_device : keys of key_bits (32 | 64) with payloads of val_bits (32 | 64; 32/32 is
 *                            lsdsort_pairs_u32_device), stable by key: every word that is not the key word being sorted on
 *                            rides through the passes as a payload array of its own (lsdsort_multi_u32_device); no gather.
 * lsdsort_wide_workspace_bytes(n, radix_bits, key_bits, val_bits) sizes the workspace (val_bits 0 = keys only);
 * lsdsort_wide_check_device reads the fault word of the sorts inside it (like lsdsort_check_device). */
LSDSORT_API size_t lsdsort_wide_workspace_bytes(size_t n, int radix_bits, int key_bits, int val_bits);
LSDSORT_API int lsdsort_u64_device(uint64_t* d_keys, void* d_workspace, size_t workspace_bytes, size_t n,
                                   int radix_bits, void* hip_stream);
LSDSORT_API int lsdsort_records_device(void* d_keys, void* d_vals, int key_bits, int val_bits, void* d_workspace,
                                       size_t workspace_bytes, size_t n, int radix_bits, void* hip_stream);
LSDSORT_API int lsdsort_wide_check_device(void* d_workspace, size_t n, int radix_bits, int key_bits, int val_bits,
                                          void* hip_stream);

/* After the stream has drained: LSDSORT_OK, or LSDSORT_ERR_DEVICE_FAULT if a kernel of the
 * last sort on this workspace gave up a bounded spin or refused destinations outside the output
 * (never expected; the output is then undefined).  Synchronises hip_stream.  With LSDSORT_REPROBE=1
 * in the environment (always in the diagnostic build) the device probe behind the default rank form
 * (lsdsort_set_rank_method) is run again here; a failure is reported as LSDSORT_ERR_DEVICE_FAULT and
 * the library uses the mask forms from then on. */
LSDSORT_API int lsdsort_check_device(void* d_workspace, void* hip_stream);

/* Per-kernel device times of one sort, by hipEvent on hip_stream (the reference times only
 * the whole sort, .cu:1002-1004).  Blocking.  Stages beyond `passes` are zero. */
#define LSDSORT_MAX_PASSES 32
typedef struct lsdsort_timing {
    float total_ms;       /* first kernel start to last kernel end                         */
    float clear_ms;       /* workspace control words + tile-status memset                   */
    float histogram_ms;   /* stage 1 (onesweep: all digits in one read; staged: summed)     */
    float scan_ms;        /* stage 2 (digit-count scan / tile offset tables)                */
    float scatter_ms[LSDSORT_MAX_PASSES]; /* stage 3, one entry per pass (chained form: the  */
                                          /* kernel's own begin/end, hipExtLaunchKernel)    */
    int passes;           /* global passes that ran (hybrid form: 2)                          */
    int tile_keys;        /* keys per rank-and-scatter tile                                 */
    int tiles;
    int hybrid;           /* 1: the hybrid form ran (two global passes + the local stage)    */
    float local_ms;       /* hybrid form: the local stage (every bucket finished in LDS)     */
} lsdsort_timing;
LSDSORT_API int lsdsort_u32_device_timed(uint32_t* d_keys, uint32_t* d_vals, void* d_workspace,
                                         size_t workspace_bytes, size_t n, int radix_bits,
                                         int algorithm, void* hip_stream, lsdsort_timing* out);

/* ---- stage entries (device pointers, stream-ordered) ---------------------------------- */
/* The three stages of a pass as separate calls, for stage-level parity and the histogram /
 * scan micro-benchmarks (counterparts of TestBuildHistogram .cu:704 and TestGPUPrefixSum
 * .cu:304).  Tables are block-major [tiles][2^radix_bits] exactly like the reference's `h`. */

/* Keys per tile of the stage entries for this radix (the reference's `block`).  Whole sorts may
 * use a larger tile for large n (lsdsort_timing.tile_keys reports it); the stage entries and the
 * staged algorithm always use this one. */
LSDSORT_API size_t lsdsort_tile_keys(int radix_bits);

/* Replaces BuildHistogramsKernel, .cu:660-702 (launch .cu:850): d_hist[t][d] = number of
 * keys of tile t whose digit `bit_group` equals d. */
LSDSORT_API int lsdsort_tile_histograms_u32_device(const uint32_t* d_keys, size_t n, int radix_bits,
                                                   int bit_group, uint32_t* d_hist, void* hip_stream);

/* Replaces .cu:862-895 (D2D copy, BlockPrefixSumKernel as local scan, two transposes,
 * GPUPrefixSum + AddBlockSumsKernel): from the counts in d_hist writes d_local (per-tile
 * exclusive scan) and d_global (digit-major exclusive scan, stored block-major).
 * d_scratch holds lsdsort_tile_offsets_scratch_bytes(tiles, radix_bits) bytes. */
LSDSORT_API size_t lsdsort_tile_offsets_scratch_bytes(size_t tiles, int radix_bits);
LSDSORT_API int lsdsort_tile_offsets_u32_device(const uint32_t* d_hist, uint32_t* d_local,
                                                uint32_t* d_global, size_t tiles, int radix_bits,
                                                void* d_scratch, void* hip_stream);

/* Replaces LSDRadixSortKernel, .cu:795-837 (launch .cu:902): stable rank inside each tile,
 * dst = rank - local[d] + global[d] (.cu:833), scatter.  d_vals_in/out may be NULL. */
LSDSORT_API int lsdsort_rank_scatter_u32_device(const uint32_t* d_in, uint32_t* d_out,
                                                const uint32_t* d_vals_in, uint32_t* d_vals_out,
                                                const uint32_t* d_global, size_t n, int radix_bits,
                                                int bit_group, void* hip_stream);

/* The hybrid form's local stage on its own (lsdradixsort_amd/csrc/local_sort.hip): bucket b = d_keys[d_bases[b] .. d_bases[b + 1])
 * (num_buckets + 1 ascending device words) is sorted IN PLACE by its keys' low `low_bits` bits (1..27; stable LSD passes of at
 * most nine bits each, run from LDS to LDS by one workgroup per bucket); d_vals (may be NULL) holds a payload word per key that
 * is permuted with it.  Buckets of more than 16384 keys are left as they are (inside a whole sort the planner has ruled them
 * out).  Needs the returning-LDS-add rank form (LSDSORT_ERR_UNSUPPORTED where the device probe failed). */
LSDSORT_API int lsdsort_local_sort_u32_device(uint32_t* d_keys, uint32_t* d_vals, const uint32_t* d_bases, size_t num_buckets,
                                              int low_bits, void* hip_stream);

/* One read of all keys -> all 32/radix_bits digit histograms, d_hist[g][d] (uint32). */
LSDSORT_API int lsdsort_digit_histograms_u32_device(const uint32_t* d_keys, size_t n, int radix_bits,
                                                    uint32_t* d_hist, void* hip_stream);

/* ---- multi-GPU building block (one process per GPU) ----------------------------------- */
/* Stable partition of this rank's shard by the top msb_bits bits (0..4): d_out holds bucket
 * 0, bucket 1, ... contiguously, d_counts[b] (uint64, 2^msb_bits entries) their sizes.  The
 * caller exchanges buckets with its RCCL communicator (all-to-all over xGMI) and then runs
 * lsdsort_u32_device on what it received.  New work; the reference is single-GPU. */
LSDSORT_API size_t lsdsort_msb_partition_workspace_bytes(size_t n, int msb_bits);
LSDSORT_API int lsdsort_msb_partition_u32_device(const uint32_t* d_in, uint32_t* d_out, size_t n,
                                                 int msb_bits, uint64_t* d_counts, void* d_workspace,
                                                 size_t workspace_bytes, void* hip_stream);
/* The same partition by value instead of by bit field, for keys that fixed MSB buckets would not
 * balance: `splitters` is a HOST array of 2^log2_buckets - 1 ascending values (copied into the launch),
 * bucket(key) = number of splitters <= key, so bucket b holds splitters[b-1] <= key < splitters[b].
 * Stable; same workspace size as the MSB form (lsdsort_msb_partition_workspace_bytes(n, log2_buckets)).
 * lsdradixsort_amd/dist.py picks the splitters from a gathered sample. */
LSDSORT_API int lsdsort_splitter_partition_u32_device(const uint32_t* d_in, uint32_t* d_out, size_t n,
                                                      int log2_buckets, const uint32_t* splitters,
                                                      uint64_t* d_counts, void* d_workspace,
                                                      size_t workspace_bytes, void* hip_stream);
/* The same with 64-bit thresholds (HOST array, 2^log2_buckets - 1 ascending values in [0, 2^32]):
 * bucket(key) = number of thresholds <= key, and 2^32 stands for "above every key", which a 32-bit splitter
 * cannot say.  The form the sharded step's splitter rule uses (lsdsort_sharded_thresholds). */
LSDSORT_API int lsdsort_threshold_partition_u32_device(const uint32_t* d_in, uint32_t* d_out, size_t n,
                                                       int log2_buckets, const uint64_t* thresholds,
                                                       uint64_t* d_counts, void* d_workspace,
                                                       size_t workspace_bytes, void* hip_stream);

/* ---- multi-GPU sort over RCCL / xGMI (BASELINE.json configs[3]) ------------------------ */
/* New work: the reference is single-GPU.  Rank b of `world` (1, 2, 4 or 8) ends up
 * owning every key whose top log2(world) bits equal b; the sorted array is the concatenation of the ranks'
 * outputs in rank order.  One step per call, on the caller's stream:
 *   1. histogram of the local keys' top bits                                    (one read)
 *   2. stable partition of the shard by those bits  ||  on a side stream: ncclAllGather of the bucket
 *      counts and capacities, count matrix to pinned host memory               (the only host wait)
 *   3. ONE grouped ncclSend/ncclRecv exchange, every peer at once (all xGMI links busy; no ring),
 *      the own bucket by a device copy
 *   4. lsdsort_u32_device on what arrived.
 * librccl is loaded on first use (dlopen); the rest of the library does not depend on it. */
typedef struct lsdsort_comm lsdsort_comm;
#define LSDSORT_COMM_ID_BYTES 128
/* One-process-per-GPU bootstrap: rank 0 makes an id (ncclGetUniqueId), the launcher's own channel
 * (torch.distributed, MPI, a file) carries its 128 bytes to every rank, every rank calls create with the HIP
 * device it will sort on current.  Collective over the `world` ranks. */
LSDSORT_API int lsdsort_comm_unique_id(void* id_out);
LSDSORT_API int lsdsort_comm_create(const void* id, int world, int rank, lsdsort_comm** out);
/* LOOPBACK: `world` (1, 2, 4 or 8) VIRTUAL ranks in this process on the CURRENT device -- out[0 .. world-1] receive their
 * communicators.  Same step, same code; the all-gathers and the grouped exchange are device copies ordered by events
 * instead of RCCL calls.  For machines with one GPU (the step's offsets, ordering, capacities and error paths can then be
 * run with world > 1: tests/test_sharded_loopback.py) and for rehearsing a launcher.  Each virtual rank must be driven by
 * its OWN host thread (the step is collective and waits for the others) and should be given its own stream.  A rank that
 * fails inside a step marks the world aborted: its peers return LSDSORT_ERR_COMM instead of waiting, and the communicators
 * are then only good for lsdsort_comm_destroy. */
LSDSORT_API int lsdsort_comm_create_loopback(int world, lsdsort_comm** out);
LSDSORT_API int lsdsort_comm_destroy(lsdsort_comm* comm);
/* Sub-buckets (1, 2 or 4; world x sub_buckets <= 16, and <= 8 under the splitter rule): the step cuts every rank's key range
 * into that many consecutive sub-ranges, exchanges them one grouped exchange after the other and sorts sub-bucket j on an
 * internal stream while sub-bucket j + 1 is still on the links -- the exchange hides under the local sort instead of in front
 * of it.  Collective setting: every rank of the world must use the same value.  Default 1. */
LSDSORT_API int lsdsort_comm_set_sub_buckets(lsdsort_comm* comm, int sub_buckets);
LSDSORT_API int lsdsort_comm_world(const lsdsort_comm* comm);
LSDSORT_API int lsdsort_comm_rank(const lsdsort_comm* comm);
/* Device workspace for a rank that contributes up to n_local_max keys and may receive up to out_capacity. */
LSDSORT_API size_t lsdsort_sharded_workspace_bytes(size_t n_local_max, size_t out_capacity, int world, int radix_bits);
/* The step.  d_keys_in (n_local keys, left untouched) and d_out (room for out_capacity keys) are device
 * pointers on the communicator's device; on return *n_out keys of d_out are this rank's slice, *global_offset
 * is the index of its first key in the global order, counts_matrix (may be NULL; world*world entries,
 * [src][dst]) says who sent what.  Collective; blocks the host only for the count matrix (the exchange and
 * the local sort stay queued on hip_stream).  If ANY rank would receive more than its out_capacity every rank
 * returns LSDSORT_ERR_CAPACITY before the exchange (capacities travel with the counts; *n_out then holds what this rank
 * would have received), so nobody hangs and the caller can repeat the step with exact sizes.  Every OTHER error is this
 * rank's alone (its peers may be inside a collective): the loopback transport then releases them with LSDSORT_ERR_COMM; over
 * RCCL the launcher has to tear the job down, as with any failed rank of an RCCL job. */
LSDSORT_API int lsdsort_sharded_u32_device(lsdsort_comm* comm, const uint32_t* d_keys_in, size_t n_local,
                                           uint32_t* d_out, size_t out_capacity, size_t* n_out,
                                           uint64_t* global_offset, uint64_t* counts_matrix,
                                           void* d_workspace, size_t workspace_bytes, int radix_bits,
                                           void* hip_stream);
/* The same step with the rule that decides which rank owns a key:
 *   LSDSORT_PARTITION_MSB        the top log2(world) key bits, as above: no extra work, balanced for uniform keys;
 *   LSDSORT_PARTITION_SPLITTERS  sampled splitters, for keys fixed MSB buckets would not balance.  Step 0 is added:
 *      every rank samples LSDSORT_SPLITTER_SAMPLES of its keys at a regular stride, one ncclAllGather hands every
 *      sample to every rank (a second, earlier host wait), and the sorted sample of (key, source rank) pairs is cut into
 *      `world` equal parts (lsdsort_sharded_thresholds).  A splitter is such a PAIR: keys equal to a splitter's value go
 *      below it from ranks before the splitter's rank and above it from the others, so a run of one value longer than a
 *      bucket is still cut (between source ranks) and the exchange stays stable.  Rank b then owns the pairs between
 *      splitters b-1 and b; the concatenation of the ranks' outputs is the sorted array as before (an MSB-style
 *      "which rank owns key k" question has no single answer for a value that sits on a splitter). */
#define LSDSORT_PARTITION_MSB 0
#define LSDSORT_PARTITION_SPLITTERS 1
#define LSDSORT_SPLITTER_SAMPLES 512
LSDSORT_API int lsdsort_sharded_u32_device_ex(lsdsort_comm* comm, const uint32_t* d_keys_in, size_t n_local,
                                              uint32_t* d_out, size_t out_capacity, size_t* n_out,
                                              uint64_t* global_offset, uint64_t* counts_matrix,
                                              void* d_workspace, size_t workspace_bytes, int radix_bits,
                                              int partition, void* hip_stream);
/* Host-side arithmetic of step 0, exported so that it can be checked without a GPU.  `gathered` is
 * [world][1 + samples_per_rank] uint32: per source rank the number of valid samples, then the samples.
 * thresholds[b-1] (b = 1 .. world-1, ascending, values in [0, 2^32]) is the smallest key of rank `rank` that goes to
 * bucket b or higher: bucket(key) = number of thresholds <= key; 2^32 = no key of this rank does. */
LSDSORT_API int lsdsort_sharded_thresholds(const uint32_t* gathered, int world, int samples_per_rank, int rank,
                                           uint64_t* thresholds);
/* The same cut into `parts` (1..8) parts instead of `world`: parts = world x sub-buckets; thresholds[parts - 1]. */
LSDSORT_API int lsdsort_sharded_thresholds_parts(const uint32_t* gathered, int world, int samples_per_rank, int rank,
                                                 int parts, uint64_t* thresholds);
/* After the stream has drained: the fault words of the step's two chained kernels sequences (partition pass, local
 * sort) in a workspace last used with these sizes; LSDSORT_OK or LSDSORT_ERR_DEVICE_FAULT.  Synchronises hip_stream. */
LSDSORT_API int lsdsort_sharded_check_device(void* d_workspace, size_t n_local, size_t out_capacity, int world,
                                             int radix_bits, void* hip_stream);
/* Text of the most recent failed RCCL call on this thread ("" if none). */
LSDSORT_API const char* lsdsort_last_comm_error(void);
/* Host-side arithmetic of step 3, exported so that it can be checked without a GPU: from the world x world
 * count matrix ([src][dst]) the element offsets of rank `rank`'s sends in its partitioned shard, of its
 * receives in its output (source-rank order keeps the exchange stable), its output size and global offset.
 * Returns LSDSORT_ERR_INVALID_ARG for a bad world / rank. */
LSDSORT_API int lsdsort_sharded_plan(const uint64_t* counts_matrix, int world, int rank, uint64_t* send_offsets,
                                     uint64_t* recv_offsets, uint64_t* n_out, uint64_t* global_offset);
/* The same with `sub` sub-buckets per rank: bucket_counts is [src][world * sub] (bucket b belongs to rank b / sub, its
 * sub-bucket b % sub); send_offsets[world * sub] by bucket, recv_offsets[sub * world] by (sub-bucket, source) -- a rank's
 * output holds sub-bucket 0 from source 0, 1, .. then sub-bucket 1 .. --, sub_sizes[sub]. */
LSDSORT_API int lsdsort_sharded_plan_sub(const uint64_t* bucket_counts, int world, int sub, int rank, uint64_t* send_offsets,
                                         uint64_t* recv_offsets, uint64_t* sub_sizes, uint64_t* n_out, uint64_t* global_offset);

/* ---- misc ----------------------------------------------------------------------------- */
LSDSORT_API const char* lsdsort_strerror(int status);
/* hipError_t of the most recent failed HIP call on this thread (0 if none) and its text. */
LSDSORT_API int lsdsort_last_hip_error(void);
LSDSORT_API const char* lsdsort_last_hip_error_string(void);
/* "lsdsort <version> gfx950 hip <runtime>" */
LSDSORT_API const char* lsdsort_version(void);
/* Number of usable devices (gfx950) visible to this process; 0 if none.  Never fails. */
LSDSORT_API int lsdsort_device_count(void);
/* One-time per-device set-up (architecture check + the LDS-atomic probe below); implicit in
 * the first compute call, explicit here so that it can be kept out of a graph capture. */
LSDSORT_API int lsdsort_prepare_device(void);
/* How the rank-and-scatter kernel ranks a key among the same-digit keys of its wavefront:
 *   0  peer masks (wave ballots for digits <= 4 bits, a wave-private LDS OR for 8 bits):
 *      correct on any hardware;
 *   2  one returning LDS add per key, which needs the LDS to serve the colliding lanes of one
 *      wave instruction in lane order; the library verifies that on the device (a probe
 *      kernel at set-up, in the workgroup sizes and LDS footprints of the sort's own kernels) and silently
 *      uses form 0 if it does not hold;
 *  -1  (default) form 2 for 4- and 8-bit digits when the probe passes, form 0 otherwise.
 * lsdsort_rank_method reports the form a sort with this radix will use on the current device. */
LSDSORT_API int lsdsort_set_rank_method(int method);
LSDSORT_API int lsdsort_rank_method(int radix_bits);
/* XCD affinity of the rank-and-scatter kernel: C consecutive tiles are claimed by workgroups of
 * one XCD so that neighbouring runs merge in one L2; 0 disables, default 16,
 * at most 64.  Speed only: results and forward progress never depend on it. */
LSDSORT_API int lsdsort_set_xcd_chunk(int chunk);
/* Dead passes.  A pass whose digit is the same for every key (small key ranges, dead high bits, constant input) is the
 * identity; the device sees that in the digit counts of the upfront read and skips it (its workgroups leave at once), and
 * one copy brings the keys back into the caller's buffer if an odd number of passes ran.  No host round trip, graph-
 * capturable, results identical.  On by default for the uint32 sorts (keys and pairs, default algorithm); the typed
 * sorts and 1-bit digits always run every pass.  0 switches it off (every pass runs, as the reference's do). */
LSDSORT_API int lsdsort_set_pass_skipping(int on);
/* The hybrid form (lsdradixsort_amd/csrc/hybrid.hip, local_sort.hip; no reference counterpart -- its every pass goes through
 * global memory, .cu:844-905).  Sorts of 3.8e7 (pairs 2.2e7, 4-bit digits 2^24) .. 9.6e8 keys or key/value pairs (uint32, int32, float32, either order) with 8- or
 * 4-bit digits: bits 16-31 are sorted first by ordinary global passes (LSD order; two passes at 8-bit digits, four at 4-bit),
 * which leaves the array sorted by its top 16 bits; every bucket of equal top-15-bit value (top 14 bits while uniform keys still fit the local stage: up to about 2^27 items; top 16
 * from 4.8e8) is
 * then finished inside one CU's LDS (bits 0-8, then 9-16) and stored once: at 8-bit digits 4 + 8 + 8 + 8 = 28 bytes per key of
 * memory traffic instead of 4 + 4 x 8 = 36 (pairs: 52 instead of 68), at 4-bit digits 44 instead of 68.  Valid only if every
 * bucket fits the local stage (16384 keys), which depends on the keys: the upfront read counts the buckets exactly and the
 * DEVICE decides before a key is moved; otherwise the ordinary passes run (after their own upfront read: such keys pay about
 * 10 % for the attempt, 1 % where a 65536-key sample already shows it).  Same result either way.  On by default; 0 = always
 * the ordinary passes, the reference's structure. */
LSDSORT_API int lsdsort_set_hybrid(int on);
/* Sorts of up to 16384 keys or pairs (uint32, default algorithm and rank form) are ONE launch: one workgroup sorts them inside
 * its LDS in four 8-bit digit passes (lsdradixsort_amd/csrc/local_sort.hip) instead of clear + stage 1 + stage 2 + 32 / r passes,
 * whose own latencies are all there is at this size (12 us instead of 39).  Same result.  On by default; 0 = the chained form at
 * every size (the reference's stage structure). */
LSDSORT_API int lsdsort_set_small_sort(int on);
/* Which form the last sort queued on `hip_stream` in this workspace ran: *hybrid = 1 the hybrid form, 0 the ordinary passes (also
 * where the hybrid form was not tried).  Reads the device's verdict back: synchronises the stream. */
LSDSORT_API int lsdsort_workspace_form(const void* d_workspace, void* hip_stream, int* hybrid);
/* lsdsort_u32_device for a SHARD of a range-partitioned array: every key is expected to agree with the others on its top
 * `common_prefix_bits` bits (0 .. 8) -- what a rank holds after the MSB-bucket exchange of the multi-GPU sort (north_star;
 * csrc/sharded.hip calls this).  The hybrid form takes its buckets BELOW a key prefix (with the prefix inside them a shard's 2^15
 * buckets would be 2^(15 - prefix) non-empty ones, each 2^prefix times too large for the local stage, and the form would be
 * refused).  Since the device finds the prefix itself -- from its 65536-key sample, for every sort (keys below 2^31, non-negative
 * int32 keys, shards ...), checked against every key by the upfront read, 0 .. 7 bits -- the argument is only validated; the entry
 * is kept for callers that say what they know.  The result is the sorted array whatever the keys are. */
LSDSORT_API int lsdsort_u32_device_prefixed(uint32_t* d_keys, void* d_workspace, size_t workspace_bytes, size_t n, int radix_bits,
                                            int common_prefix_bits, void* hip_stream);
/* Runtime tuning knob for experiments: selects among the compiled tile shapes (see
 * DESIGN.md); -1 restores the default.  Returns LSDSORT_ERR_INVALID_ARG if unknown. */
LSDSORT_API int lsdsort_set_tile_config(int radix_bits, int config_id);

#ifdef __cplusplus
}
#endif
#endif /* LSDSORT_H */
