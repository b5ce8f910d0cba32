// sharded.hip -- the multi-GPU sort behind the C-ABI (include/lsdsort.h, "multi-GPU sort over RCCL / xGMI").
//
// New work: the reference is single-GPU (BASELINE.json configs[3] is the target).  Host
// code only; every kernel it runs belongs to the single-GPU path (MSB histogram + partition pass, then the
// ordinary LSD sort).  RCCL is loaded with dlopen on first use, so liblsdsort.so itself carries no dependency
// on it and the single-GPU entries work on machines without librccl.
//
// One step (lsdsort_sharded_u32_device), rank r of W, b = log2 W:
//   main stream : clear | histogram of the top b bits | scan | counts -> d_vec | EVENT | partition pass -> send buffer
//   side stream :                                                     wait EVENT | capacity -> d_vec[W] |
//                 ncclAllGather (W+1 words per rank) | copy to pinned host memory           <- the host waits HERE only,
//                                                                                              while the partition runs
//   host        : offsets from the count matrix (lsdsort_sharded_plan); every rank checks every rank's capacity
//   main stream : ONE ncclGroupStart .. ncclGroupEnd with a send and a receive per peer (all links at once; xGMI is
//                 point-to-point, a ring would be per-link bound) | own bucket by hipMemcpyAsync | lsdsort_u32_device
// Buckets arrive in source-rank order, so the exchange keeps equal keys in (source rank, source position) order.
#define LSDSORT_BUILD 1
#include "../../include/lsdsort.h"

#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

#include "lsd_kernels.hpp"

namespace lsd {
// lsdsort_api.hip
int partition_with_event(const uint32_t* d_in, uint32_t* d_out, size_t n, int msb_bits, uint64_t* d_counts,
                         void* d_workspace, size_t workspace_bytes, hipStream_t stream, hipEvent_t counts_ready);
int threshold_partition_with_event(const uint32_t* d_in, uint32_t* d_out, size_t n, int log2_buckets, const uint64_t* thresholds,
                                   uint64_t* d_counts, void* d_workspace, size_t workspace_bytes, hipStream_t stream,
                                   hipEvent_t counts_ready);
void set_last_hip_error(hipError_t e);
}  // namespace lsd

namespace {

constexpr size_t kAlign = 256;
size_t align_up(size_t x) { return (x + kAlign - 1) / kAlign * kAlign; }

thread_local char t_comm_error[256] = "";

// ---- librccl, loaded on demand -------------------------------------------------------------------------------
struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};

Rccl& rccl()
{
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            r.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (r.handle) break;
        }
        if (!r.handle) return;
        bool all = true;
        auto sym = [&](auto& fn, const char* name) {
            fn = reinterpret_cast<std::remove_reference_t<decltype(fn)>>(dlsym(r.handle, name));
            all = all && fn != nullptr;
        };
        sym(r.GetUniqueId, "ncclGetUniqueId");
        sym(r.CommInitRank, "ncclCommInitRank");
        sym(r.CommInitAll, "ncclCommInitAll");
        sym(r.CommDestroy, "ncclCommDestroy");
        sym(r.AllGather, "ncclAllGather");
        sym(r.Send, "ncclSend");
        sym(r.Recv, "ncclRecv");
        sym(r.GroupStart, "ncclGroupStart");
        sym(r.GroupEnd, "ncclGroupEnd");
        sym(r.GetErrorString, "ncclGetErrorString");
        r.ok = all;
    });
    return r;
}

#define SH_HIP(expr)                          \
    do {                                      \
        hipError_t e__ = (expr);              \
        if (e__ != hipSuccess) {              \
            lsd::set_last_hip_error(e__);     \
            (void)hipGetLastError();          \
            return LSDSORT_ERR_HIP;           \
        }                                     \
    } while (0)

#define SH_NCCL(expr)                                                                                        \
    do {                                                                                                     \
        ncclResult_t r__ = (expr);                                                                           \
        if (r__ != ncclSuccess) {                                                                            \
            std::snprintf(t_comm_error, sizeof(t_comm_error), "%s: %s", #expr, rccl().GetErrorString(r__));  \
            return LSDSORT_ERR_COMM;                                                                         \
        }                                                                                                    \
    } while (0)

#define SH_TRY(expr)                       \
    do {                                   \
        int s__ = (expr);                  \
        if (s__ != LSDSORT_OK) return s__; \
    } while (0)

int log2_world(int world)
{
    switch (world) {
        case 1: return 0;
        case 2: return 1;
        case 4: return 2;
        case 8: return 3;
        default: return -1;
    }
}

struct ShardedLayout {
    size_t vec = 0;        // u64[W + 1]: this rank's bucket counts, then its output capacity
    size_t all = 0;        // u64[W][W + 1]: everybody's
    size_t part_ws = 0;    // workspace of the partition pass
    size_t send = 0;       // the partitioned shard: bucket 0 | bucket 1 | ...
    size_t sort_ws = 0;    // workspace of the local sort
    size_t samp = 0;       // u32[1 + S]: this rank's sample (count, then keys) for the splitter rule
    size_t samp_all = 0;   // u32[W][1 + S]: everybody's
    size_t total = 0;
    size_t part_ws_bytes = 0, sort_ws_bytes = 0;
};

ShardedLayout make_sharded_layout(size_t n_local_max, size_t out_capacity, int world, int radix_bits)
{
    ShardedLayout L;
    const int bits = log2_world(world);
    size_t off = 0;
    L.vec = off;
    off = align_up(off + (size_t)(world + 1) * sizeof(uint64_t));
    L.all = off;
    off = align_up(off + (size_t)world * (world + 1) * sizeof(uint64_t));
    L.part_ws = off;
    L.part_ws_bytes = lsdsort_msb_partition_workspace_bytes(n_local_max, bits);
    off = align_up(off + L.part_ws_bytes);
    L.send = off;
    off = align_up(off + n_local_max * sizeof(uint32_t));
    L.sort_ws = off;
    L.sort_ws_bytes = lsdsort_workspace_bytes(out_capacity, radix_bits, 0);
    off = align_up(off + L.sort_ws_bytes);
    L.samp = off;
    off = align_up(off + (size_t)(1 + LSDSORT_SPLITTER_SAMPLES) * sizeof(uint32_t));
    L.samp_all = off;
    off = align_up(off + (size_t)world * (1 + LSDSORT_SPLITTER_SAMPLES) * sizeof(uint32_t));
    L.total = off;
    return L;
}

}  // namespace

struct lsdsort_comm {
    ncclComm_t comm = nullptr;
    int world = 1, rank = 0, device = 0;
    hipStream_t side = nullptr;
    hipEvent_t counts_ready = nullptr;
    hipEvent_t sample_ready = nullptr;
    uint64_t* h_all = nullptr;       // pinned: [W][W + 1]
    uint32_t* h_samples = nullptr;   // pinned: [W][1 + S]
};

namespace {

int finish_comm(lsdsort_comm* c)
{
    SH_HIP(hipGetDevice(&c->device));
    SH_HIP(hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking));
    SH_HIP(hipEventCreateWithFlags(&c->counts_ready, hipEventDisableTiming));
    SH_HIP(hipEventCreateWithFlags(&c->sample_ready, hipEventDisableTiming));
    SH_HIP(hipHostMalloc(reinterpret_cast<void**>(&c->h_all), (size_t)c->world * (c->world + 1) * sizeof(uint64_t), hipHostMallocDefault));
    SH_HIP(hipHostMalloc(reinterpret_cast<void**>(&c->h_samples), (size_t)c->world * (1 + LSDSORT_SPLITTER_SAMPLES) * sizeof(uint32_t),
                         hipHostMallocDefault));
    return LSDSORT_OK;
}

}  // namespace

extern "C" {

const char* lsdsort_last_comm_error(void) { return t_comm_error; }

int lsdsort_comm_unique_id(void* id_out)
{
    if (!id_out) return LSDSORT_ERR_INVALID_ARG;
    if (!rccl().ok) return LSDSORT_ERR_UNSUPPORTED;
    static_assert(sizeof(ncclUniqueId) == LSDSORT_COMM_ID_BYTES, "id size is part of the ABI");
    ncclUniqueId id;
    SH_NCCL(rccl().GetUniqueId(&id));
    std::memcpy(id_out, &id, sizeof(id));
    return LSDSORT_OK;
}

int lsdsort_comm_create(const void* id, int world, int rank, lsdsort_comm** out)
{
    if (!id || !out || log2_world(world) < 0 || rank < 0 || rank >= world) return LSDSORT_ERR_INVALID_ARG;
    *out = nullptr;
    SH_TRY(lsdsort_prepare_device());          // gfx950 check + probe, on the current device
    if (!rccl().ok) return LSDSORT_ERR_UNSUPPORTED;
    lsdsort_comm* c = new lsdsort_comm;
    c->world = world;
    c->rank = rank;
    ncclUniqueId nid;
    std::memcpy(&nid, id, sizeof(nid));
    ncclResult_t r = rccl().CommInitRank(&c->comm, world, nid, rank);
    if (r != ncclSuccess) {
        std::snprintf(t_comm_error, sizeof(t_comm_error), "ncclCommInitRank: %s", rccl().GetErrorString(r));
        delete c;
        return LSDSORT_ERR_COMM;
    }
    const int st = finish_comm(c);
    if (st != LSDSORT_OK) {
        (void)lsdsort_comm_destroy(c);
        return st;
    }
    *out = c;
    return LSDSORT_OK;
}

int lsdsort_comm_destroy(lsdsort_comm* c)
{
    if (!c) return LSDSORT_OK;
    int prev = 0;
    const bool switched = hipGetDevice(&prev) == hipSuccess && prev != c->device && hipSetDevice(c->device) == hipSuccess;
    if (c->h_all) (void)hipHostFree(c->h_all);
    if (c->h_samples) (void)hipHostFree(c->h_samples);
    if (c->counts_ready) (void)hipEventDestroy(c->counts_ready);
    if (c->sample_ready) (void)hipEventDestroy(c->sample_ready);
    if (c->side) (void)hipStreamDestroy(c->side);
    if (c->comm && rccl().ok) (void)rccl().CommDestroy(c->comm);
    if (switched) (void)hipSetDevice(prev);
    (void)hipGetLastError();
    delete c;
    return LSDSORT_OK;
}

int lsdsort_comm_world(const lsdsort_comm* c) { return c ? c->world : LSDSORT_ERR_INVALID_ARG; }
int lsdsort_comm_rank(const lsdsort_comm* c) { return c ? c->rank : LSDSORT_ERR_INVALID_ARG; }

size_t lsdsort_sharded_workspace_bytes(size_t n_local_max, size_t out_capacity, int world, int radix_bits)
{
    if (log2_world(world) < 0 || n_local_max > LSDSORT_MAX_KEYS || out_capacity > LSDSORT_MAX_KEYS) return 0;
    if (lsdsort_workspace_bytes(out_capacity, radix_bits, 0) == 0 && out_capacity > 0) return 0;
    if (lsdsort_workspace_bytes(1, radix_bits, 0) == 0) return 0;   // bad radix
    return make_sharded_layout(n_local_max, out_capacity, world, radix_bits).total;
}

int lsdsort_sharded_plan(const uint64_t* m, int world, int rank, uint64_t* send_offsets, uint64_t* recv_offsets,
                         uint64_t* n_out, uint64_t* global_offset)
{
    if (!m || log2_world(world) < 0 || rank < 0 || rank >= world) return LSDSORT_ERR_INVALID_ARG;
    uint64_t s = 0, r = 0, before = 0;
    for (int p = 0; p < world; p++) {
        if (send_offsets) send_offsets[p] = s;       // bucket p of my partitioned shard
        s += m[(size_t)rank * world + p];
        if (recv_offsets) recv_offsets[p] = r;       // what source p sends me, in source-rank order
        r += m[(size_t)p * world + rank];
        for (int dst = 0; dst < rank; dst++) before += m[(size_t)p * world + dst];   // everything owned by lower ranks
    }
    if (n_out) *n_out = r;
    if (global_offset) *global_offset = before;
    return LSDSORT_OK;
}

int lsdsort_sharded_thresholds(const uint32_t* gathered, int world, int samples_per_rank, int rank, uint64_t* thresholds)
{
    if (!gathered || log2_world(world) < 0 || samples_per_rank < 1 || rank < 0 || rank >= world || (world > 1 && !thresholds))
        return LSDSORT_ERR_INVALID_ARG;
    // every sampled key as (key, source rank): equal keys are told apart by where they came from, so a run of one
    // value longer than a bucket can still be cut (between ranks), and the cut keeps the exchange stable
    std::vector<uint64_t> tuples;
    tuples.reserve((size_t)world * samples_per_rank);
    for (int src = 0; src < world; src++) {
        const uint32_t* row = gathered + (size_t)src * (1 + samples_per_rank);
        if (row[0] > (uint32_t)samples_per_rank) return LSDSORT_ERR_INVALID_ARG;
        for (uint32_t i = 0; i < row[0]; i++) tuples.push_back(((uint64_t)row[1 + i] << 8) | (uint64_t)src);
    }
    std::sort(tuples.begin(), tuples.end());
    const size_t total = tuples.size();
    for (int b = 1; b < world; b++) {
        if (total == 0) {   // nothing to sort anywhere: any rule does
            thresholds[b - 1] = 1ull << 32;
            continue;
        }
        const uint64_t cut = tuples[(size_t)(((unsigned __int128)b * total) / world)];
        const uint64_t key = cut >> 8;
        const int from = (int)(cut & 0xFF);
        // (k, rank) >= (key, from)  <=>  k > key, or k == key and rank >= from  <=>  k >= key + (rank < from)
        thresholds[b - 1] = key + (rank < from ? 1u : 0u);
    }
    return LSDSORT_OK;
}

int lsdsort_sharded_u32_device(lsdsort_comm* c, const uint32_t* d_keys_in, size_t n_local, uint32_t* d_out,
                               size_t out_capacity, size_t* n_out, uint64_t* global_offset, uint64_t* counts_matrix,
                               void* d_workspace, size_t workspace_bytes, int radix_bits, void* hip_stream)
{
    return lsdsort_sharded_u32_device_ex(c, d_keys_in, n_local, d_out, out_capacity, n_out, global_offset, counts_matrix, d_workspace,
                                         workspace_bytes, radix_bits, LSDSORT_PARTITION_MSB, hip_stream);
}

int lsdsort_sharded_u32_device_ex(lsdsort_comm* c, const uint32_t* d_keys_in, size_t n_local, uint32_t* d_out,
                                  size_t out_capacity, size_t* n_out, uint64_t* global_offset, uint64_t* counts_matrix,
                                  void* d_workspace, size_t workspace_bytes, int radix_bits, int partition, void* hip_stream)
{
    if (!c || !n_out || !global_offset) return LSDSORT_ERR_INVALID_ARG;
    if (partition != LSDSORT_PARTITION_MSB && partition != LSDSORT_PARTITION_SPLITTERS) return LSDSORT_ERR_INVALID_ARG;
    if (n_local > LSDSORT_MAX_KEYS || out_capacity > LSDSORT_MAX_KEYS) return LSDSORT_ERR_TOO_LARGE;
    if ((n_local > 0 && !d_keys_in) || (out_capacity > 0 && !d_out)) return LSDSORT_ERR_INVALID_ARG;
    if (lsdsort_workspace_bytes(1, radix_bits, 0) == 0) return LSDSORT_ERR_INVALID_ARG;
    int dev = -1;
    SH_HIP(hipGetDevice(&dev));
    if (dev != c->device) return LSDSORT_ERR_INVALID_ARG;       // the communicator lives on the device it was made on
    const int W = c->world, bits = log2_world(W);
    const ShardedLayout L = make_sharded_layout(n_local, out_capacity, W, radix_bits);
    if (!d_workspace || (reinterpret_cast<uintptr_t>(d_workspace) & (kAlign - 1)) || workspace_bytes < L.total) return LSDSORT_ERR_WORKSPACE;
    hipStream_t stream = static_cast<hipStream_t>(hip_stream);
    char* ws = static_cast<char*>(d_workspace);
    uint64_t* d_vec = reinterpret_cast<uint64_t*>(ws + L.vec);
    uint64_t* d_all = reinterpret_cast<uint64_t*>(ws + L.all);
    uint32_t* d_send = reinterpret_cast<uint32_t*>(ws + L.send);
    Rccl& R = rccl();

    // 0.  splitter rule only: a regular sample of every shard to every rank (one more host wait, ahead of the partition);
    //     each rank then cuts the sorted (key, source rank) sample into W equal parts and derives ITS thresholds
    uint64_t thresholds[8] = {};
    if (partition == LSDSORT_PARTITION_SPLITTERS) {
        constexpr int S = LSDSORT_SPLITTER_SAMPLES;
        uint32_t* d_samp = reinterpret_cast<uint32_t*>(ws + L.samp);
        uint32_t* d_samp_all = reinterpret_cast<uint32_t*>(ws + L.samp_all);
        SH_HIP(lsd::launch_sample_keys(d_keys_in, (uint32_t)n_local, (uint32_t)S, d_samp, stream));
        SH_HIP(hipEventRecord(c->sample_ready, stream));
        SH_HIP(hipStreamWaitEvent(c->side, c->sample_ready, 0));
        SH_NCCL(R.AllGather(d_samp, d_samp_all, (size_t)(1 + S), ncclUint32, c->comm, c->side));
        SH_HIP(hipMemcpyAsync(c->h_samples, d_samp_all, (size_t)W * (1 + S) * sizeof(uint32_t), hipMemcpyDeviceToHost, c->side));
        SH_HIP(hipStreamSynchronize(c->side));
        SH_TRY(lsdsort_sharded_thresholds(c->h_samples, W, S, c->rank, thresholds));
    }

    // 1 + 2.  main stream: counts, EVENT, partition pass.  side stream: count exchange while the partition runs.
    if (partition == LSDSORT_PARTITION_SPLITTERS)
        SH_TRY(lsd::threshold_partition_with_event(d_keys_in, d_send, n_local, bits, thresholds, d_vec, ws + L.part_ws, L.part_ws_bytes,
                                                   stream, c->counts_ready));
    else
        SH_TRY(lsd::partition_with_event(d_keys_in, d_send, n_local, bits, d_vec, ws + L.part_ws, L.part_ws_bytes, stream, c->counts_ready));
    SH_HIP(hipStreamWaitEvent(c->side, c->counts_ready, 0));
    SH_HIP(lsd::launch_store_u64(d_vec + W, (uint64_t)out_capacity, c->side));
    SH_NCCL(R.AllGather(d_vec, d_all, (size_t)(W + 1), ncclUint64, c->comm, c->side));
    SH_HIP(hipMemcpyAsync(c->h_all, d_all, (size_t)W * (W + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost, c->side));
    SH_HIP(hipStreamSynchronize(c->side));                      // the step's only host wait

    // host: the plan, identical on every rank; so is the verdict on everybody's capacity
    uint64_t m[64], send_off[8], recv_off[8], total = 0, offset = 0;
    bool fits = true;
    for (int src = 0; src < W; src++)
        for (int dst = 0; dst < W; dst++) m[src * W + dst] = c->h_all[(size_t)src * (W + 1) + dst];
    for (int dst = 0; dst < W; dst++) {
        uint64_t recv = 0;
        for (int src = 0; src < W; src++) recv += m[src * W + dst];
        if (recv > c->h_all[(size_t)dst * (W + 1) + W]) fits = false;
    }
    SH_TRY(lsdsort_sharded_plan(m, W, c->rank, send_off, recv_off, &total, &offset));
    if (counts_matrix) std::memcpy(counts_matrix, m, (size_t)W * W * sizeof(uint64_t));
    *n_out = (size_t)total;
    *global_offset = offset;
    if (!fits) return LSDSORT_ERR_TOO_LARGE;                   // every rank returns this, none has posted a send

    // 3.  one grouped exchange, every peer at once; my own bucket stays on the device
    SH_NCCL(R.GroupStart());
    for (int step = 1; step < W; step++) {
        const int to = (c->rank + step) % W, from = (c->rank - step + W) % W;   // a different partner pair per step
        const uint64_t ns = m[c->rank * W + to], nr = m[from * W + c->rank];
        if (ns) SH_NCCL(R.Send(d_send + send_off[to], (size_t)ns, ncclUint32, to, c->comm, stream));
        if (nr) SH_NCCL(R.Recv(d_out + recv_off[from], (size_t)nr, ncclUint32, from, c->comm, stream));
    }
    SH_NCCL(R.GroupEnd());
    const uint64_t mine = m[c->rank * W + c->rank];
    if (mine)
        SH_HIP(hipMemcpyAsync(d_out + recv_off[c->rank], d_send + send_off[c->rank], (size_t)mine * sizeof(uint32_t), hipMemcpyDeviceToDevice, stream));

    // 4.  the local LSD passes (the top bits are constant within a rank; all 32 bits are still sorted)
    if (total == 0) SH_HIP(hipMemsetAsync(ws + L.sort_ws, 0, sizeof(uint32_t), stream));   // an empty sort never touches its fault word
    return lsdsort_u32_device(d_out, ws + L.sort_ws, L.sort_ws_bytes, (size_t)total, radix_bits, stream);
}

int lsdsort_sharded_check_device(void* d_workspace, size_t n_local, size_t out_capacity, int world, int radix_bits, void* hip_stream)
{
    if (!d_workspace) return LSDSORT_ERR_WORKSPACE;
    if (log2_world(world) < 0 || lsdsort_workspace_bytes(1, radix_bits, 0) == 0) return LSDSORT_ERR_INVALID_ARG;
    const ShardedLayout L = make_sharded_layout(n_local, out_capacity, world, radix_bits);
    int status = lsdsort_check_device(static_cast<char*>(d_workspace) + L.part_ws, hip_stream);   // the partition pass is chained too
    if (status == LSDSORT_OK) status = lsdsort_check_device(static_cast<char*>(d_workspace) + L.sort_ws, hip_stream);
    return status;
}

}  // extern "C"

// ---- lsdsort_u32_ex(keys, n, radix_bits, num_gpus > 1): one process, one host thread per device ------------------
namespace lsd {

namespace {
struct CommSet {
    std::vector<lsdsort_comm*> comms;
};
std::mutex g_sets_mutex;
CommSet* g_sets[9] = {};   // by device count; made once, kept for the life of the process

int comm_set(int ndev, CommSet** out)
{
    std::lock_guard<std::mutex> lock(g_sets_mutex);
    if (g_sets[ndev]) {
        *out = g_sets[ndev];
        return LSDSORT_OK;
    }
    if (!rccl().ok) return LSDSORT_ERR_UNSUPPORTED;
    std::vector<ncclComm_t> raw(ndev);
    std::vector<int> devs(ndev);
    for (int i = 0; i < ndev; i++) devs[i] = i;
    SH_NCCL(rccl().CommInitAll(raw.data(), ndev, devs.data()));
    CommSet* set = new CommSet;
    int prev = 0;
    (void)hipGetDevice(&prev);
    int status = LSDSORT_OK;
    for (int i = 0; i < ndev && status == LSDSORT_OK; i++) {
        lsdsort_comm* c = new lsdsort_comm;
        c->comm = raw[i];
        c->world = ndev;
        c->rank = i;
        set->comms.push_back(c);
        if (hipSetDevice(i) != hipSuccess) status = LSDSORT_ERR_NO_DEVICE;
        else status = finish_comm(c);
    }
    (void)hipSetDevice(prev);
    if (status != LSDSORT_OK) {
        for (lsdsort_comm* c : set->comms) (void)lsdsort_comm_destroy(c);
        delete set;
        return status;
    }
    g_sets[ndev] = set;
    *out = set;
    return LSDSORT_OK;
}

int sort_shard(lsdsort_comm* c, uint32_t* keys, size_t n, size_t begin, size_t n_local, int radix_bits)
{
    SH_HIP(hipSetDevice(c->rank));
    SH_TRY(lsdsort_prepare_device());
    const size_t cap = n;   // any distribution: in the worst case every key belongs to one rank
    const size_t ws_bytes = lsdsort_sharded_workspace_bytes(n_local, cap, c->world, radix_bits);
    uint32_t *d_in = nullptr, *d_out = nullptr;
    void* d_ws = nullptr;
    auto body = [&]() -> int {
        SH_HIP(hipMalloc(reinterpret_cast<void**>(&d_in), (n_local ? n_local : 1) * sizeof(uint32_t)));
        SH_HIP(hipMalloc(reinterpret_cast<void**>(&d_out), cap * sizeof(uint32_t)));
        SH_HIP(hipMalloc(&d_ws, ws_bytes));
        if (n_local) SH_HIP(hipMemcpy(d_in, keys + begin, n_local * sizeof(uint32_t), hipMemcpyHostToDevice));
        size_t n_out = 0;
        uint64_t offset = 0;
        SH_TRY(lsdsort_sharded_u32_device(c, d_in, n_local, d_out, cap, &n_out, &offset, nullptr, d_ws, ws_bytes, radix_bits, nullptr));
        SH_TRY(lsdsort_sharded_check_device(d_ws, n_local, cap, c->world, radix_bits, nullptr));
        // every rank's shard left the host before its sends were posted, and my receives needed everybody's sends:
        // nothing of the input array is still unread when a slice comes back into it
        if (n_out) SH_HIP(hipMemcpy(keys + offset, d_out, n_out * sizeof(uint32_t), hipMemcpyDeviceToHost));
        return LSDSORT_OK;
    };
    const int status = body();
    if (d_ws) (void)hipFree(d_ws);
    if (d_out) (void)hipFree(d_out);
    if (d_in) (void)hipFree(d_in);
    return status;
}
}  // namespace

int sort_host_multi(uint32_t* keys, size_t n, int radix_bits, int num_gpus)
{
    if (log2_world(num_gpus) < 1) return LSDSORT_ERR_INVALID_ARG;
    if (n > LSDSORT_MAX_KEYS) return LSDSORT_ERR_TOO_LARGE;
    if (lsdsort_workspace_bytes(1, radix_bits, 0) == 0) return LSDSORT_ERR_INVALID_ARG;
    if (n == 0) return LSDSORT_OK;
    if (!keys) return LSDSORT_ERR_INVALID_ARG;
    if (lsdsort_device_count() < num_gpus) return LSDSORT_ERR_NO_DEVICE;
    int prev = 0;
    (void)hipGetDevice(&prev);
    CommSet* set = nullptr;
    SH_TRY(comm_set(num_gpus, &set));
    std::vector<int> status(num_gpus, LSDSORT_OK);
    std::vector<std::thread> threads;
    const size_t per = n / num_gpus, rem = n % num_gpus;
    size_t begin = 0;
    for (int i = 0; i < num_gpus; i++) {
        const size_t n_local = per + ((size_t)i < rem ? 1 : 0);
        threads.emplace_back([&, i, begin, n_local] { status[i] = sort_shard(set->comms[i], keys, n, begin, n_local, radix_bits); });
        begin += n_local;
    }
    for (std::thread& t : threads) t.join();
    (void)hipSetDevice(prev);
    for (int s : status)
        if (s != LSDSORT_OK) return s;
    return LSDSORT_OK;
}

}  // namespace lsd
