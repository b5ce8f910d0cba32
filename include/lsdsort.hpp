// lsdsort.hpp -- C++ face of liblsdsort.so: namespace lsd { sort(...) }.
//
// `lsd::sort(uint32_t* keys, size_t n)` is the host-side entry point BASELINE.json's north_star
// names.  The reference has no such symbol; it is defined as the body of
// TestGPULSDRadixSort between LSDRadixSort/LSDRadixSort.cu:1001 and :1005 (H2D, GPULSDRadixSort,
// D2H).  Header-only wrappers over include/lsdsort.h; a non-zero status becomes an exception (the
// reference crashes instead: MYCRASH, Utils.h:6-15).
#pragma once
#include <cstddef>
#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "lsdsort.h"

namespace lsd {

class sort_error : public std::runtime_error {
public:
    sort_error(int status, const char* where)
        : std::runtime_error(std::string(where) + ": " + lsdsort_strerror(status)), status_(status) {}
    int status() const noexcept { return status_; }

private:
    int status_;
};

inline void check(int status, const char* where)
{
    if (status != LSDSORT_OK) throw sort_error(status, where);
}

// Host array, in place, ascending.  Blocking.
inline void sort(uint32_t* keys, size_t n) { check(lsdsort_u32(keys, n), "lsdsort_u32"); }
inline void sort(uint32_t* keys, size_t n, int radix_bits) { check(lsdsort_u32_ex(keys, n, radix_bits, 1), "lsdsort_u32_ex"); }

// Host array over several GPUs of this node (one process; RCCL over xGMI; BASELINE.json configs[3]).
inline void sort(uint32_t* keys, size_t n, int radix_bits, int num_gpus) { check(lsdsort_u32_ex(keys, n, radix_bits, num_gpus), "lsdsort_u32_ex"); }

// Host key/value arrays, stable by key.
inline void sort_pairs(uint32_t* keys, uint32_t* vals, size_t n) { check(lsdsort_pairs_u32(keys, vals, n), "lsdsort_pairs_u32"); }

// Device-resident sort, stream-ordered: the counterpart of GPULSDRadixSort(a, b, h, ...), .cu:839.
inline size_t workspace_bytes(size_t n, int radix_bits = 8, bool pairs = false) { return lsdsort_workspace_bytes(n, radix_bits, pairs ? 1 : 0); }
inline void sort_device(uint32_t* d_keys, void* d_workspace, size_t workspace_bytes_, size_t n, int radix_bits = 8,
                        void* hip_stream = nullptr)
{
    check(lsdsort_u32_device(d_keys, d_workspace, workspace_bytes_, n, radix_bits, hip_stream), "lsdsort_u32_device");
}
inline void sort_pairs_device(uint32_t* d_keys, uint32_t* d_vals, void* d_workspace, size_t workspace_bytes_, size_t n,
                              int radix_bits = 8, void* hip_stream = nullptr)
{
    check(lsdsort_pairs_u32_device(d_keys, d_vals, d_workspace, workspace_bytes_, n, radix_bits, hip_stream),
          "lsdsort_pairs_u32_device");
}

// Other 32-bit key types and orders (no reference counterpart): the overload picks the key type.
inline void sort_device(int32_t* d_keys, void* d_workspace, size_t workspace_bytes_, size_t n, bool descending = false,
                        uint32_t* d_vals = nullptr, int radix_bits = 8, void* hip_stream = nullptr)
{
    check(lsdsort_keys_device(d_keys, d_vals, d_workspace, workspace_bytes_, n, radix_bits, LSDSORT_KEY_I32, descending ? 1 : 0,
                              hip_stream), "lsdsort_keys_device");
}
inline void sort_device(float* d_keys, void* d_workspace, size_t workspace_bytes_, size_t n, bool descending = false,
                        uint32_t* d_vals = nullptr, int radix_bits = 8, void* hip_stream = nullptr)
{
    check(lsdsort_keys_device(d_keys, d_vals, d_workspace, workspace_bytes_, n, radix_bits, LSDSORT_KEY_F32, descending ? 1 : 0,
                              hip_stream), "lsdsort_keys_device");
}
inline void sort_device_descending(uint32_t* d_keys, void* d_workspace, size_t workspace_bytes_, size_t n,
                                   uint32_t* d_vals = nullptr, int radix_bits = 8, void* hip_stream = nullptr)
{
    check(lsdsort_keys_device(d_keys, d_vals, d_workspace, workspace_bytes_, n, radix_bits, LSDSORT_KEY_U32, 1, hip_stream),
          "lsdsort_keys_device");
}

// A shard of a range-partitioned array: keys expected to share their top `common_prefix_bits` bits (a hint; the device checks)
inline void sort_shard_device(uint32_t* d_keys, void* d_workspace, size_t workspace_bytes_, size_t n, int common_prefix_bits,
                              int radix_bits = 8, void* hip_stream = nullptr)
{
    check(lsdsort_u32_device_prefixed(d_keys, d_workspace, workspace_bytes_, n, radix_bits, common_prefix_bits, hip_stream),
          "lsdsort_u32_device_prefixed");
}
// Did the last sort queued in this workspace run the hybrid form (lsdsort_set_hybrid)?  Synchronises the stream.
inline bool ran_hybrid_form(const void* d_workspace, void* hip_stream = nullptr)
{
    int hybrid = 0;
    check(lsdsort_workspace_form(d_workspace, hip_stream, &hybrid), "lsdsort_workspace_form");
    return hybrid != 0;
}

// One rank of a multi-GPU sort (one process per GPU): RAII over lsdsort_comm_*.  Rank 0 calls unique_id() and ships
// the 128 bytes to the other ranks over the launcher's own channel (MPI_Bcast, a file, torch.distributed).
struct comm_id {
    unsigned char bytes[LSDSORT_COMM_ID_BYTES];
};
inline comm_id unique_id()
{
    comm_id id;
    check(lsdsort_comm_unique_id(id.bytes), "lsdsort_comm_unique_id");
    return id;
}
class communicator {
public:
    communicator(const comm_id& id, int world, int rank) { check(lsdsort_comm_create(id.bytes, world, rank, &c_), "lsdsort_comm_create"); }
    // `world` VIRTUAL ranks on the current device (lsdsort_comm_create_loopback): element r is rank r's communicator; each is to
    // be driven by its own host thread.  For one-GPU machines: the step then runs with world > 1 (tests/cpp/test_sharded.cpp).
    static std::vector<std::unique_ptr<communicator>> loopback(int world)
    {
        std::vector<lsdsort_comm*> raw((size_t)(world > 0 ? world : 1), nullptr);
        check(lsdsort_comm_create_loopback(world, raw.data()), "lsdsort_comm_create_loopback");
        std::vector<std::unique_ptr<communicator>> out;
        for (lsdsort_comm* c : raw) out.emplace_back(new communicator(c));
        return out;
    }
    ~communicator() { lsdsort_comm_destroy(c_); }
    communicator(const communicator&) = delete;
    communicator& operator=(const communicator&) = delete;
    int world() const { return lsdsort_comm_world(c_); }
    int rank() const { return lsdsort_comm_rank(c_); }
    size_t workspace_bytes(size_t n_local_max, size_t out_capacity, int radix_bits = 8) const
    {
        return lsdsort_sharded_workspace_bytes(n_local_max, out_capacity, world(), radix_bits);
    }
    struct slice {
        size_t n;                 // keys of d_out that are this rank's part of the result
        uint64_t global_offset;   // index of the first of them in the global order
    };
    // Collective.  d_keys_in is left untouched; the exchange and the local sort stay queued on hip_stream.
    // partition: LSDSORT_PARTITION_MSB (top key bits; uniform keys) or LSDSORT_PARTITION_SPLITTERS (sampled; any keys).
    slice sort_device(const uint32_t* d_keys_in, size_t n_local, uint32_t* d_out, size_t out_capacity, void* d_workspace,
                      size_t workspace_bytes_, int radix_bits = 8, void* hip_stream = nullptr, uint64_t* counts_matrix = nullptr,
                      int partition = LSDSORT_PARTITION_MSB)
    {
        slice s{0, 0};
        check(lsdsort_sharded_u32_device_ex(c_, d_keys_in, n_local, d_out, out_capacity, &s.n, &s.global_offset, counts_matrix,
                                            d_workspace, workspace_bytes_, radix_bits, partition, hip_stream),
              "lsdsort_sharded_u32_device_ex");
        return s;
    }

private:
    explicit communicator(lsdsort_comm* adopted) : c_(adopted) {}
    lsdsort_comm* c_ = nullptr;
};

}  // namespace lsd
