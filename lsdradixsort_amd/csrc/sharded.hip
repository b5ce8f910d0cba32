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
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <memory>
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

constexpr int kMaxBuckets = 16;   // world x sub-buckets: the partition pass takes digits of up to four bits

struct ShardedLayout {
    size_t sticky = 0;     // u32: fault words of the step's local sorts, ORed together (they share one workspace, and every
                           // sort's opening memset clears that workspace's own word)
    size_t vec = 0;        // u64[B + 1]: this rank's bucket counts, then its output capacity   (B <= kMaxBuckets)
    size_t all = 0;        // u64[W][B + 1]: everybody's
    size_t part_ws = 0;    // workspace of the partition pass
    size_t send = 0;       // the partitioned shard: bucket 0 | bucket 1 | ...
    size_t sort_ws = 0;    // workspace of the local sorts
    size_t samp = 0;       // u32[1 + S]: this rank's sample (count, then keys) for the splitter rule
    size_t samp_all = 0;   // u32[W][1 + S]: everybody's
    size_t total = 0;
    size_t part_ws_bytes = 0, sort_ws_bytes = 0;
};

// The layout does not depend on the number of sub-buckets a step uses: every table is sized for kMaxBuckets.
ShardedLayout make_sharded_layout(size_t n_local_max, size_t out_capacity, int world, int radix_bits)
{
    ShardedLayout L;
    size_t off = 0;
    L.sticky = off;
    off += kAlign;
    L.vec = off;
    off = align_up(off + (size_t)(kMaxBuckets + 1) * sizeof(uint64_t));
    L.all = off;
    off = align_up(off + (size_t)world * (kMaxBuckets + 1) * sizeof(uint64_t));
    L.part_ws = off;
    for (int bits = 0; bits <= 4; bits++) {   // whichever digit width the step's bucket count asks for
        const size_t need = lsdsort_msb_partition_workspace_bytes(n_local_max, bits);
        if (need > L.part_ws_bytes) L.part_ws_bytes = need;
    }
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


// ---- the transport seam ---------------------------------------------------------------------------------------
// Everything the step needs from the fabric: an all-gather of a few words per rank, and one grouped exchange of
// sends and receives.  Two implementations: RCCL (one process per GPU, or one process with a host thread per
// device), and a LOOPBACK -- W virtual ranks in one process on ONE device, each driven by its own host thread on its
// own streams, exchanging by device copies -- so that the whole step (offsets, ordering, capacities, error paths) runs
// with W in {2, 4, 8} where only one GPU exists (tests/test_sharded_loopback.py).  Collective calls are made by every
// rank, in the same order; sizes in bytes.
struct Transport {
    virtual ~Transport() {}
    virtual int all_gather(const void* send, void* recv, size_t bytes_per_rank, hipStream_t stream) = 0;
    virtual int group_start() = 0;
    virtual int send(const void* src, size_t bytes, int peer, hipStream_t stream) = 0;
    virtual int recv(void* dst, size_t bytes, int peer, hipStream_t stream) = 0;
    virtual int group_end(hipStream_t stream) = 0;   // ALWAYS called after group_start, also when a send or receive failed: it reports the first error
    // A rank that leaves a collective step early (a local error) tells the others, where the transport can: peers blocked
    // in a collective of the loopback return LSDSORT_ERR_COMM instead of waiting for ever.  (An RCCL peer cannot be reached
    // this way: there the ranks agree on a status BEFORE the first collective -- sort_host_multi -- or the launcher
    // tears the job down.)
    virtual void abort() {}
};

struct RcclTransport final : Transport {
    ncclComm_t comm = nullptr;
    int first_error = LSDSORT_OK;
    int note(ncclResult_t r, const char* what)
    {
        if (r == ncclSuccess) return LSDSORT_OK;
        std::snprintf(t_comm_error, sizeof(t_comm_error), "%s: %s", what, rccl().GetErrorString(r));
        return LSDSORT_ERR_COMM;
    }
    int all_gather(const void* send, void* recv, size_t bytes_per_rank, hipStream_t stream) override
    {
        return note(rccl().AllGather(send, recv, bytes_per_rank, ncclUint8, comm, stream), "ncclAllGather");
    }
    int group_start() override
    {
        first_error = LSDSORT_OK;
        return note(rccl().GroupStart(), "ncclGroupStart");
    }
    int send(const void* src, size_t bytes, int peer, hipStream_t stream) override
    {
        if (first_error != LSDSORT_OK) return first_error;   // the group is closed by group_end either way
        return first_error = note(rccl().Send(src, bytes, ncclUint8, peer, comm, stream), "ncclSend");
    }
    int recv(void* dst, size_t bytes, int peer, hipStream_t stream) override
    {
        if (first_error != LSDSORT_OK) return first_error;
        return first_error = note(rccl().Recv(dst, bytes, ncclUint8, peer, comm, stream), "ncclRecv");
    }
    int group_end(hipStream_t) override
    {
        // the group is closed whatever happened inside it: an open group would swallow every later call of this thread
        const int closed = note(rccl().GroupEnd(), "ncclGroupEnd");
        return first_error != LSDSORT_OK ? first_error : closed;
    }
    ~RcclTransport() override
    {
        if (comm && rccl().ok) (void)rccl().CommDestroy(comm);
    }
};

// The loopback world: state shared by the W virtual ranks of one process.
struct LoopbackWorld {
    int world = 1;
    std::mutex m;
    std::condition_variable cv;
    int arrived = 0;
    unsigned long long generation = 0;
    bool aborted = false;
    struct Op {
        const void* src;
        void* dst;
        size_t bytes;
        int peer;
    };
    struct Post {
        const void* send = nullptr;   // all-gather
        size_t bytes = 0;
        std::vector<Op> sends, recvs; // grouped exchange
        hipEvent_t ready = nullptr;   // "what I posted may be read" (recorded on my stream)
        hipEvent_t done = nullptr;    // "I have queued my reads of everybody's data" (recorded on my stream)
    } post[8];
    // host barrier of the W rank threads; LSDSORT_ERR_COMM once any rank has aborted
    int barrier()
    {
        std::unique_lock<std::mutex> lock(m);
        if (aborted) return LSDSORT_ERR_COMM;
        const unsigned long long gen = generation;
        if (++arrived == world) {
            arrived = 0;
            generation++;
            cv.notify_all();
            return LSDSORT_OK;
        }
        cv.wait(lock, [&] { return generation != gen || aborted; });
        return generation != gen ? LSDSORT_OK : LSDSORT_ERR_COMM;
    }
    void abort()
    {
        std::lock_guard<std::mutex> lock(m);
        aborted = true;
        cv.notify_all();
    }
};

struct LoopbackTransport final : Transport {
    std::shared_ptr<LoopbackWorld> w;
    int rank = 0;
    bool in_group = false;
    int first_error = LSDSORT_OK;
    int fail(const char* what)
    {
        std::snprintf(t_comm_error, sizeof(t_comm_error), "loopback: %s", what);
        return LSDSORT_ERR_COMM;
    }
    // Stream-ordered like the real thing: data is read once its owner's stream has reached the collective (ready), and
    // an owner's later work waits until every reader has queued behind it (done).
    int all_gather(const void* send, void* recv, size_t bytes_per_rank, hipStream_t stream) override
    {
        LoopbackWorld::Post& mine = w->post[rank];
        mine.send = send;
        mine.bytes = bytes_per_rank;
        SH_HIP(hipEventRecord(mine.ready, stream));
        if (w->barrier() != LSDSORT_OK) return fail("a rank left the all-gather");
        for (int p = 0; p < w->world; p++) {
            if (w->post[p].bytes != bytes_per_rank) return fail("all-gather sizes differ between ranks");
            SH_HIP(hipStreamWaitEvent(stream, w->post[p].ready, 0));
            SH_HIP(hipMemcpyAsync(static_cast<char*>(recv) + (size_t)p * bytes_per_rank, w->post[p].send, bytes_per_rank,
                                  hipMemcpyDeviceToDevice, stream));
        }
        SH_HIP(hipEventRecord(mine.done, stream));
        if (w->barrier() != LSDSORT_OK) return fail("a rank left the all-gather");
        for (int p = 0; p < w->world; p++) SH_HIP(hipStreamWaitEvent(stream, w->post[p].done, 0));
        return LSDSORT_OK;
    }
    int group_start() override
    {
        in_group = true;
        first_error = LSDSORT_OK;
        w->post[rank].sends.clear();
        w->post[rank].recvs.clear();
        return LSDSORT_OK;
    }
    int send(const void* src, size_t bytes, int peer, hipStream_t) override
    {
        if (!in_group || peer < 0 || peer >= w->world || peer == rank) return first_error = fail("send outside a group or to a bad peer");
        w->post[rank].sends.push_back({src, nullptr, bytes, peer});
        return LSDSORT_OK;
    }
    int recv(void* dst, size_t bytes, int peer, hipStream_t) override
    {
        if (!in_group || peer < 0 || peer >= w->world || peer == rank) return first_error = fail("receive outside a group or from a bad peer");
        w->post[rank].recvs.push_back({nullptr, dst, bytes, peer});
        return LSDSORT_OK;
    }
    int group_end(hipStream_t stream) override
    {
        in_group = false;
        if (first_error != LSDSORT_OK) {   // nothing of this group can be trusted: the peers are told, nobody waits for this rank
            w->abort();
            return first_error;
        }
        LoopbackWorld::Post& mine = w->post[rank];
        SH_HIP(hipEventRecord(mine.ready, stream));
        if (w->barrier() != LSDSORT_OK) return fail("a rank left the exchange");
        // my k-th receive from peer p takes p's k-th send to me
        int status = LSDSORT_OK;
        for (size_t i = 0; i < mine.recvs.size() && status == LSDSORT_OK; i++) {
            const LoopbackWorld::Op& r = mine.recvs[i];
            size_t nth = 0;
            for (size_t j = 0; j < i; j++) nth += mine.recvs[j].peer == r.peer ? 1 : 0;
            const LoopbackWorld::Op* match = nullptr;
            for (const LoopbackWorld::Op& s : w->post[r.peer].sends)
                if (s.peer == rank && nth-- == 0) { match = &s; break; }
            if (!match || match->bytes != r.bytes) { status = fail("a receive without a matching send of the same size"); break; }
            if (hipStreamWaitEvent(stream, w->post[r.peer].ready, 0) != hipSuccess ||
                hipMemcpyAsync(r.dst, match->src, r.bytes, hipMemcpyDeviceToDevice, stream) != hipSuccess) {
                (void)hipGetLastError();
                status = fail("device copy failed");
            }
        }
        if (status == LSDSORT_OK && hipEventRecord(mine.done, stream) != hipSuccess) status = fail("event record failed");
        if (status != LSDSORT_OK) {
            w->abort();
            return status;
        }
        if (w->barrier() != LSDSORT_OK) return fail("a rank left the exchange");
        for (int p = 0; p < w->world; p++) SH_HIP(hipStreamWaitEvent(stream, w->post[p].done, 0));
        return LSDSORT_OK;
    }
    void abort() override { w->abort(); }
    ~LoopbackTransport() override
    {
        LoopbackWorld::Post& mine = w->post[rank];
        if (mine.ready) (void)hipEventDestroy(mine.ready);
        if (mine.done) (void)hipEventDestroy(mine.done);
        mine.ready = mine.done = nullptr;
    }
};

}  // namespace

struct lsdsort_comm {
    Transport* transport = nullptr;
    int world = 1, rank = 0, device = 0;
    int sub_buckets = 1;             // lsdsort_comm_set_sub_buckets
    hipStream_t side = nullptr;      // count / sample exchange beside the partition pass
    hipStream_t sorter = nullptr;    // local sorts of sub-buckets that have arrived, beside the exchange of the next ones
    hipEvent_t counts_ready = nullptr;
    hipEvent_t sample_ready = nullptr;
    hipEvent_t arrived[4] = {};      // sub-bucket j is complete in the output buffer
    hipEvent_t sorted_all = nullptr; // the last local sort has finished
    uint64_t* h_all = nullptr;       // pinned: [W][kMaxBuckets + 1]
    uint32_t* h_samples = nullptr;   // pinned: [W][1 + S]
};

namespace {

int finish_comm(lsdsort_comm* c)
{
    SH_HIP(hipGetDevice(&c->device));
    SH_HIP(hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking));
    SH_HIP(hipStreamCreateWithFlags(&c->sorter, hipStreamNonBlocking));
    SH_HIP(hipEventCreateWithFlags(&c->counts_ready, hipEventDisableTiming));
    SH_HIP(hipEventCreateWithFlags(&c->sample_ready, hipEventDisableTiming));
    SH_HIP(hipEventCreateWithFlags(&c->sorted_all, hipEventDisableTiming));
    for (hipEvent_t& e : c->arrived) SH_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    SH_HIP(hipHostMalloc(reinterpret_cast<void**>(&c->h_all), (size_t)c->world * (kMaxBuckets + 1) * sizeof(uint64_t), hipHostMallocDefault));
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
    RcclTransport* t = new RcclTransport;
    c->transport = t;
    ncclResult_t r = rccl().CommInitRank(&t->comm, world, nid, rank);
    if (r != ncclSuccess) {
        std::snprintf(t_comm_error, sizeof(t_comm_error), "ncclCommInitRank: %s", rccl().GetErrorString(r));
        t->comm = nullptr;
        delete t;
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

int lsdsort_comm_create_loopback(int world, lsdsort_comm** out)
{
    if (!out || log2_world(world) < 0) return LSDSORT_ERR_INVALID_ARG;
    for (int i = 0; i < world; i++) out[i] = nullptr;
    SH_TRY(lsdsort_prepare_device());
    auto shared = std::make_shared<LoopbackWorld>();
    shared->world = world;
    int status = LSDSORT_OK;
    for (int i = 0; i < world && status == LSDSORT_OK; i++) {
        lsdsort_comm* c = new lsdsort_comm;
        c->world = world;
        c->rank = i;
        LoopbackTransport* t = new LoopbackTransport;
        t->w = shared;
        t->rank = i;
        c->transport = t;
        out[i] = c;
        if (hipEventCreateWithFlags(&shared->post[i].ready, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&shared->post[i].done, hipEventDisableTiming) != hipSuccess) {
            (void)hipGetLastError();
            status = LSDSORT_ERR_HIP;
        } else {
            status = finish_comm(c);
        }
    }
    if (status != LSDSORT_OK) {
        for (int i = 0; i < world; i++) {
            (void)lsdsort_comm_destroy(out[i]);
            out[i] = nullptr;
        }
    }
    return status;
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
    if (c->sorted_all) (void)hipEventDestroy(c->sorted_all);
    for (hipEvent_t e : c->arrived)
        if (e) (void)hipEventDestroy(e);
    if (c->side) (void)hipStreamDestroy(c->side);
    if (c->sorter) (void)hipStreamDestroy(c->sorter);
    delete c->transport;
    if (switched) (void)hipSetDevice(prev);
    (void)hipGetLastError();
    delete c;
    return LSDSORT_OK;
}

int lsdsort_comm_set_sub_buckets(lsdsort_comm* c, int sub_buckets)
{
    if (!c || (sub_buckets != 1 && sub_buckets != 2 && sub_buckets != 4) || c->world * sub_buckets > kMaxBuckets) return LSDSORT_ERR_INVALID_ARG;
    c->sub_buckets = sub_buckets;
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

int lsdsort_sharded_plan_sub(const uint64_t* m, int world, int sub, int rank, uint64_t* send_offsets, uint64_t* recv_offsets,
                             uint64_t* sub_sizes, uint64_t* n_out, uint64_t* global_offset)
{
    if (!m || log2_world(world) < 0 || (sub != 1 && sub != 2 && sub != 4) || world * sub > kMaxBuckets || rank < 0 || rank >= world)
        return LSDSORT_ERR_INVALID_ARG;
    const int B = world * sub;
    uint64_t s = 0;
    for (int b = 0; b < B; b++) {          // bucket b of my partitioned shard: destination b / sub, its sub-bucket b % sub
        if (send_offsets) send_offsets[b] = s;
        s += m[(size_t)rank * B + b];
    }
    uint64_t r = 0;
    for (int j = 0; j < sub; j++) {        // my output: sub-bucket 0 from source 0, 1, .. | sub-bucket 1 from source 0, 1, .. | ...
        uint64_t size = 0;
        for (int src = 0; src < world; src++) {
            if (recv_offsets) recv_offsets[j * world + src] = r;
            r += m[(size_t)src * B + rank * sub + j];
            size += m[(size_t)src * B + rank * sub + j];
        }
        if (sub_sizes) sub_sizes[j] = size;
    }
    uint64_t before = 0;
    for (int src = 0; src < world; src++)
        for (int b = 0; b < rank * sub; b++) before += m[(size_t)src * B + b];   // everything owned by lower ranks
    if (n_out) *n_out = r;
    if (global_offset) *global_offset = before;
    return LSDSORT_OK;
}

int lsdsort_sharded_thresholds(const uint32_t* gathered, int world, int samples_per_rank, int rank, uint64_t* thresholds)
{
    return lsdsort_sharded_thresholds_parts(gathered, world, samples_per_rank, rank, world, thresholds);
}

int lsdsort_sharded_thresholds_parts(const uint32_t* gathered, int world, int samples_per_rank, int rank, int parts, uint64_t* thresholds)
{
    if (!gathered || log2_world(world) < 0 || samples_per_rank < 1 || rank < 0 || rank >= world || parts < 1 || parts > 8 ||
        (parts > 1 && !thresholds))
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
    for (int b = 1; b < parts; b++) {
        if (total == 0) {   // nothing to sort anywhere: any rule does
            thresholds[b - 1] = 1ull << 32;
            continue;
        }
        const uint64_t cut = tuples[(size_t)(((unsigned __int128)b * total) / parts)];
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

static int sharded_step(lsdsort_comm* c, const uint32_t* d_keys_in, size_t n_local, uint32_t* d_out,
                        size_t out_capacity, size_t* n_out, uint64_t* global_offset, uint64_t* counts_matrix,
                        void* d_workspace, size_t workspace_bytes, int radix_bits, int partition, void* hip_stream);

int lsdsort_sharded_u32_device_ex(lsdsort_comm* c, const uint32_t* d_keys_in, size_t n_local, uint32_t* d_out,
                                  size_t out_capacity, size_t* n_out, uint64_t* global_offset, uint64_t* counts_matrix,
                                  void* d_workspace, size_t workspace_bytes, int radix_bits, int partition, void* hip_stream)
{
    if (!c) return LSDSORT_ERR_INVALID_ARG;
    const int status = sharded_step(c, d_keys_in, n_local, d_out, out_capacity, n_out, global_offset, counts_matrix, d_workspace,
                                    workspace_bytes, radix_bits, partition, hip_stream);
    // LSDSORT_ERR_CAPACITY is the one error every rank returns together (decided from the gathered capacities, before the
    // exchange).  Any other failure is this rank's alone: it leaves the step while its peers may be inside a collective.
    if (status != LSDSORT_OK && status != LSDSORT_ERR_CAPACITY) c->transport->abort();
    return status;
}

static int sharded_step(lsdsort_comm* c, const uint32_t* d_keys_in, size_t n_local, uint32_t* d_out,
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
    const int W = c->world, S = c->sub_buckets, B = W * S;
    int bits = 0;
    while ((1 << bits) < B) bits++;
    if (partition == LSDSORT_PARTITION_SPLITTERS && B > 8) return LSDSORT_ERR_UNSUPPORTED;   // the value partition cuts into eight at most
    const ShardedLayout L = make_sharded_layout(n_local, out_capacity, W, radix_bits);
    if (!d_workspace || (reinterpret_cast<uintptr_t>(d_workspace) & (kAlign - 1)) || workspace_bytes < L.total) return LSDSORT_ERR_WORKSPACE;
    hipStream_t stream = static_cast<hipStream_t>(hip_stream);
    char* ws = static_cast<char*>(d_workspace);
    uint64_t* d_vec = reinterpret_cast<uint64_t*>(ws + L.vec);
    uint64_t* d_all = reinterpret_cast<uint64_t*>(ws + L.all);
    uint32_t* d_send = reinterpret_cast<uint32_t*>(ws + L.send);
    uint32_t* sticky = reinterpret_cast<uint32_t*>(ws + L.sticky);
    Transport& T = *c->transport;
    SH_HIP(hipMemsetAsync(sticky, 0, sizeof(uint32_t), stream));

    // 0.  splitter rule only: a regular sample of every shard to every rank (one more host wait, ahead of the partition);
    //     each rank then cuts the sorted (key, source rank) sample into B equal parts and derives ITS thresholds
    uint64_t thresholds[8] = {};
    if (partition == LSDSORT_PARTITION_SPLITTERS) {
        constexpr int NS = LSDSORT_SPLITTER_SAMPLES;
        uint32_t* d_samp = reinterpret_cast<uint32_t*>(ws + L.samp);
        uint32_t* d_samp_all = reinterpret_cast<uint32_t*>(ws + L.samp_all);
        SH_HIP(lsd::launch_sample_keys(d_keys_in, (uint32_t)n_local, (uint32_t)NS, d_samp, stream));
        SH_HIP(hipEventRecord(c->sample_ready, stream));
        SH_HIP(hipStreamWaitEvent(c->side, c->sample_ready, 0));
        SH_TRY(T.all_gather(d_samp, d_samp_all, (size_t)(1 + NS) * sizeof(uint32_t), c->side));
        SH_HIP(hipMemcpyAsync(c->h_samples, d_samp_all, (size_t)W * (1 + NS) * sizeof(uint32_t), hipMemcpyDeviceToHost, c->side));
        SH_HIP(hipStreamSynchronize(c->side));
        SH_TRY(lsdsort_sharded_thresholds_parts(c->h_samples, W, NS, c->rank, B, thresholds));
    }

    // 1 + 2.  main stream: counts, EVENT, partition pass into B buckets (destination rank b / S, its sub-bucket b % S).
    //         side stream: count exchange while the partition runs.
    if (partition == LSDSORT_PARTITION_SPLITTERS)
        SH_TRY(lsd::threshold_partition_with_event(d_keys_in, d_send, n_local, bits, thresholds, d_vec, ws + L.part_ws, L.part_ws_bytes,
                                                   stream, c->counts_ready));
    else
        SH_TRY(lsd::partition_with_event(d_keys_in, d_send, n_local, bits, d_vec, ws + L.part_ws, L.part_ws_bytes, stream, c->counts_ready));
    SH_HIP(hipStreamWaitEvent(c->side, c->counts_ready, 0));
    SH_HIP(lsd::launch_store_u64(d_vec + B, (uint64_t)out_capacity, c->side));
    SH_TRY(T.all_gather(d_vec, d_all, (size_t)(B + 1) * sizeof(uint64_t), c->side));
    SH_HIP(hipMemcpyAsync(c->h_all, d_all, (size_t)W * (B + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost, c->side));
    SH_HIP(hipStreamSynchronize(c->side));                      // the step's only host wait

    // host: the plan, identical on every rank; so is the verdict on everybody's capacity
    uint64_t m[8 * kMaxBuckets], send_off[kMaxBuckets], recv_off[4 * 8], sub_size[4], total = 0, offset = 0;
    bool fits = true;
    for (int src = 0; src < W; src++)
        for (int b = 0; b < B; b++) m[src * B + b] = c->h_all[(size_t)src * (B + 1) + b];
    for (int dst = 0; dst < W; dst++) {
        uint64_t recv = 0;
        for (int src = 0; src < W; src++)
            for (int j = 0; j < S; j++) recv += m[src * B + dst * S + j];
        if (recv > c->h_all[(size_t)dst * (B + 1) + B]) fits = false;
    }
    SH_TRY(lsdsort_sharded_plan_sub(m, W, S, c->rank, send_off, recv_off, sub_size, &total, &offset));
    if (counts_matrix)                                          // [src][dst], whole ranks
        for (int src = 0; src < W; src++)
            for (int dst = 0; dst < W; dst++) {
                uint64_t sum = 0;
                for (int j = 0; j < S; j++) sum += m[src * B + dst * S + j];
                counts_matrix[src * W + dst] = sum;
            }
    *n_out = (size_t)total;
    *global_offset = offset;
    if (!fits) return LSDSORT_ERR_CAPACITY;                    // every rank returns this, none has posted a send

    // 3 + 4.  Per sub-bucket j: one grouped exchange, every peer at once (my own part stays on the device), then its local
    //         sort -- on the `sorter` stream, so that sub-bucket j is being sorted while j + 1 is still on the links (a rank's
    //         sub-buckets are consecutive key ranges: sorted one by one they are the sorted slice).  A group is closed whatever
    //         happens inside it (an open RCCL group would swallow every later call of this thread); group_end reports the
    //         first error of the group.
    uint64_t sub_begin = 0;
    for (int j = 0; j < S; j++) {
        SH_TRY(T.group_start());
        for (int step = 1; step < W; step++) {
            const int to = (c->rank + step) % W, from = (c->rank - step + W) % W;   // a different partner pair per step
            const uint64_t ns = m[c->rank * B + to * S + j], nr = m[from * B + c->rank * S + j];
            if (ns && T.send(d_send + send_off[to * S + j], (size_t)ns * sizeof(uint32_t), to, stream) != LSDSORT_OK) break;
            if (nr && T.recv(d_out + recv_off[j * W + from], (size_t)nr * sizeof(uint32_t), from, stream) != LSDSORT_OK) break;
        }
        SH_TRY(T.group_end(stream));
        const uint64_t mine = m[c->rank * B + c->rank * S + j];
        if (mine)
            SH_HIP(hipMemcpyAsync(d_out + recv_off[j * W + c->rank], d_send + send_off[c->rank * S + j], (size_t)mine * sizeof(uint32_t),
                                  hipMemcpyDeviceToDevice, stream));
        // the local LSD passes of what has arrived (the top bits are constant within a sub-bucket; all 32 bits are still sorted)
        hipStream_t sort_on = S > 1 ? c->sorter : stream;
        if (S > 1) {
            SH_HIP(hipEventRecord(c->arrived[j], stream));
            SH_HIP(hipStreamWaitEvent(c->sorter, c->arrived[j], 0));
        }
        if (sub_size[j] == 0) SH_HIP(hipMemsetAsync(ws + L.sort_ws, 0, sizeof(uint32_t), sort_on));   // an empty sort never touches its fault word
        // under the MSB partition the keys of sub-bucket j of rank r all carry the top `bits` bits (r, j): the hybrid form plans
        // its buckets below such a prefix (hybrid.hip; the device finds it itself -- the argument says what this caller knows)
        SH_TRY(lsdsort_u32_device_prefixed(d_out + sub_begin, ws + L.sort_ws, L.sort_ws_bytes, (size_t)sub_size[j], radix_bits,
                                           partition == LSDSORT_PARTITION_MSB ? bits : 0, sort_on));
        SH_HIP(lsd::launch_keep_fault(sticky, reinterpret_cast<const uint32_t*>(ws + L.sort_ws), sort_on));
        sub_begin += sub_size[j];
    }
    if (S > 1) {   // the caller's stream sees the step complete
        SH_HIP(hipEventRecord(c->sorted_all, c->sorter));
        SH_HIP(hipStreamWaitEvent(stream, c->sorted_all, 0));
    }
    return LSDSORT_OK;
}

int lsdsort_sharded_check_device(void* d_workspace, size_t n_local, size_t out_capacity, int world, int radix_bits, void* hip_stream)
{
    if (!d_workspace) return LSDSORT_ERR_WORKSPACE;
    if (log2_world(world) < 0 || lsdsort_workspace_bytes(1, radix_bits, 0) == 0) return LSDSORT_ERR_INVALID_ARG;
    const ShardedLayout L = make_sharded_layout(n_local, out_capacity, world, radix_bits);
    int status = lsdsort_check_device(static_cast<char*>(d_workspace) + L.part_ws, hip_stream);   // the partition pass is chained too
    // the local sorts' fault words, kept in the sticky word (the sorts share a workspace; lsdsort_check_device reads the first
    // word of what it is given)
    if (status == LSDSORT_OK) status = lsdsort_check_device(static_cast<char*>(d_workspace) + L.sticky, hip_stream);
    return status;
}

}  // extern "C"

// ---- lsdsort_u32_ex(keys, n, radix_bits, num_gpus > 1): one process, one host thread per device ------------------
// Also lsdsort_u32_loopback: the same code with `num_gpus` VIRTUAL ranks on the current device (loopback transport), so
// that threads, set-up agreement, capacity retry, step and copy back run where only one GPU exists.
namespace lsd {

namespace {
// Host-side agreement of the set's rank threads: everybody contributes a status and gets the worst one back.  Used in
// front of the first collective of a call, so that a rank whose set-up failed (no memory, a bad device) takes the others
// out with it instead of leaving them blocked in an all-gather it will never join.
struct HostAgreement {
    std::mutex m;
    std::condition_variable cv;
    int world = 1, arrived = 0, worst = LSDSORT_OK, agreed = LSDSORT_OK;
    unsigned long long generation = 0;
    int agree(int status)
    {
        std::unique_lock<std::mutex> lock(m);
        if (status < worst) worst = status;
        const unsigned long long gen = generation;
        if (++arrived == world) {
            arrived = 0;
            agreed = worst;
            worst = LSDSORT_OK;
            generation++;
            cv.notify_all();
        } else {
            cv.wait(lock, [&] { return generation != gen; });
        }
        return agreed;
    }
};

struct CommSet {
    std::mutex busy;                 // one call at a time per set: the communicators, side streams and pinned buffers are shared
    std::vector<lsdsort_comm*> comms;
    std::vector<int> devices;        // HIP device of each rank
    HostAgreement agreement;
};
std::mutex g_sets_mutex;
CommSet* g_sets[9] = {};        // by device count; made once, kept for the life of the process
CommSet* g_loop_sets[9] = {};   // the loopback twins, on the device current at their creation

int comm_set(int ndev, bool loopback, CommSet** out)
{
    std::lock_guard<std::mutex> lock(g_sets_mutex);
    int cur = 0;
    SH_HIP(hipGetDevice(&cur));
    CommSet*& slot = loopback ? g_loop_sets[ndev] : g_sets[ndev];
    if (slot && loopback && slot->devices[0] != cur) {   // the loopback set follows the current device
        for (lsdsort_comm* c : slot->comms) (void)lsdsort_comm_destroy(c);
        delete slot;
        slot = nullptr;
    }
    if (slot) {
        *out = slot;
        return LSDSORT_OK;
    }
    CommSet* set = new CommSet;
    set->agreement.world = ndev;
    int status = LSDSORT_OK;
    if (loopback) {
        std::vector<lsdsort_comm*> raw(ndev, nullptr);
        status = lsdsort_comm_create_loopback(ndev, raw.data());
        if (status == LSDSORT_OK) {
            set->comms = raw;
            set->devices.assign(ndev, cur);
        }
    } else {
        if (!rccl().ok) {
            delete set;
            return LSDSORT_ERR_UNSUPPORTED;
        }
        std::vector<ncclComm_t> raw(ndev);
        std::vector<int> devs(ndev);
        for (int i = 0; i < ndev; i++) devs[i] = i;
        ncclResult_t r = rccl().CommInitAll(raw.data(), ndev, devs.data());
        if (r != ncclSuccess) {
            std::snprintf(t_comm_error, sizeof(t_comm_error), "ncclCommInitAll: %s", rccl().GetErrorString(r));
            delete set;
            return LSDSORT_ERR_COMM;
        }
        for (int i = 0; i < ndev; i++) {
            lsdsort_comm* c = new lsdsort_comm;
            RcclTransport* t = new RcclTransport;
            t->comm = raw[i];
            c->transport = t;
            c->world = ndev;
            c->rank = i;
            set->comms.push_back(c);
            set->devices.push_back(i);
            if (status == LSDSORT_OK) {
                if (hipSetDevice(i) != hipSuccess) status = LSDSORT_ERR_NO_DEVICE;
                else status = finish_comm(c);
            }
        }
        (void)hipSetDevice(cur);
    }
    if (status != LSDSORT_OK) {
        for (lsdsort_comm* c : set->comms) (void)lsdsort_comm_destroy(c);
        delete set;
        return status;
    }
    slot = set;
    *out = set;
    return LSDSORT_OK;
}

// One rank's share of a call.  Phase A (local: device, buffers, upload) ends in an agreement of all ranks; only if every
// rank is ready does anybody enter phase B (the collective step).  Capacity: a share plus a quarter first; if some rank
// would receive more (skewed keys) every rank learns it together (LSDSORT_ERR_CAPACITY) and the step is repeated once with
// exact sizes -- again behind an agreement, because the larger buffers are allocated locally.
int sort_shard(CommSet* set, int rank, uint32_t* keys, size_t begin, size_t n_local, int radix_bits)
{
    lsdsort_comm* c = set->comms[rank];
    uint32_t *d_in = nullptr, *d_out = nullptr;
    void* d_ws = nullptr;
    hipStream_t stream = nullptr;   // this rank's own (virtual ranks share a device: the null stream would serialise them)
    size_t cap = n_local + n_local / 4 + 4096, ws_bytes = 0;
    if (cap > LSDSORT_MAX_KEYS) cap = LSDSORT_MAX_KEYS;
    auto release = [&]() {
        if (d_ws) (void)hipFree(d_ws);
        if (d_out) (void)hipFree(d_out);
        d_ws = nullptr;
        d_out = nullptr;
    };
    auto reserve = [&]() -> int {
        ws_bytes = lsdsort_sharded_workspace_bytes(n_local, cap, c->world, radix_bits);
        if (ws_bytes == 0) return LSDSORT_ERR_TOO_LARGE;
        SH_HIP(hipMalloc(reinterpret_cast<void**>(&d_out), (cap ? cap : 1) * sizeof(uint32_t)));
        SH_HIP(hipMalloc(&d_ws, ws_bytes));
        return LSDSORT_OK;
    };
    auto set_up = [&]() -> int {
        SH_HIP(hipSetDevice(set->devices[rank]));
        SH_TRY(lsdsort_prepare_device());
        SH_HIP(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
        SH_HIP(hipMalloc(reinterpret_cast<void**>(&d_in), (n_local ? n_local : 1) * sizeof(uint32_t)));
        if (n_local) SH_HIP(hipMemcpy(d_in, keys + begin, n_local * sizeof(uint32_t), hipMemcpyHostToDevice));
        return reserve();
    };
    int status = set->agreement.agree(set_up());
    size_t n_out = 0;
    uint64_t offset = 0;
    for (int attempt = 0; attempt < 2 && status == LSDSORT_OK; attempt++) {
        status = lsdsort_sharded_u32_device(c, d_in, n_local, d_out, cap, &n_out, &offset, nullptr, d_ws, ws_bytes, radix_bits, stream);
        if (status != LSDSORT_ERR_CAPACITY || attempt == 1) break;
        // collective verdict: every rank is here.  Exact sizes, local allocation, agreement, once more.
        (void)hipStreamSynchronize(stream);
        release();
        cap = n_out;
        status = set->agreement.agree(reserve());
    }
    if (status == LSDSORT_OK) status = lsdsort_sharded_check_device(d_ws, n_local, cap, c->world, radix_bits, stream);
    // every rank's shard left the host before its sends were posted, and my receives needed everybody's sends:
    // nothing of the input array is still unread when a slice comes back into it
    if (status == LSDSORT_OK && n_out && hipMemcpy(keys + offset, d_out, n_out * sizeof(uint32_t), hipMemcpyDeviceToHost) != hipSuccess) {
        (void)hipGetLastError();
        status = LSDSORT_ERR_HIP;
    }
    if (stream) {
        (void)hipStreamSynchronize(stream);
        (void)hipStreamSynchronize(c->side);
        (void)hipStreamDestroy(stream);
    }
    release();
    if (d_in) (void)hipFree(d_in);
    (void)hipGetLastError();
    return status;
}
}  // namespace

int sort_host_multi(uint32_t* keys, size_t n, int radix_bits, int num_gpus, bool loopback)
{
    if (log2_world(num_gpus) < 1) return LSDSORT_ERR_INVALID_ARG;
    if (n > LSDSORT_MAX_KEYS) return LSDSORT_ERR_TOO_LARGE;
    if (lsdsort_workspace_bytes(1, radix_bits, 0) == 0) return LSDSORT_ERR_INVALID_ARG;
    if (n == 0) return LSDSORT_OK;
    if (!keys) return LSDSORT_ERR_INVALID_ARG;
    if (lsdsort_device_count() < (loopback ? 1 : num_gpus)) return LSDSORT_ERR_NO_DEVICE;
    int prev = 0;
    (void)hipGetDevice(&prev);
    CommSet* set = nullptr;
    SH_TRY(comm_set(num_gpus, loopback, &set));
    std::lock_guard<std::mutex> busy(set->busy);
    std::vector<int> status(num_gpus, LSDSORT_OK);
    std::vector<std::thread> threads;
    const size_t per = n / num_gpus, rem = n % num_gpus;
    size_t begin = 0;
    for (int i = 0; i < num_gpus; i++) {
        const size_t n_local = per + ((size_t)i < rem ? 1 : 0);
        threads.emplace_back([&, i, begin, n_local] { status[i] = sort_shard(set, i, keys, begin, n_local, radix_bits); });
        begin += n_local;
    }
    for (std::thread& t : threads) t.join();
    (void)hipSetDevice(prev);
    for (int s : status)
        if (s != LSDSORT_OK) return s;
    return LSDSORT_OK;
}

}  // namespace lsd
