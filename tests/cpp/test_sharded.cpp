// C++ test of the multi-GPU entry points on ONE GPU: a communicator of one rank exercises every RCCL call of the step
// (ncclGetUniqueId, ncclCommInitRank, ncclAllGather of the counts, the grouped exchange -- empty with one rank --, the
// device copy of the own bucket) and the whole host-side sequencing of lsdsort_sharded_u32_device.  N > 1 needs an
// 8-GPU node (the driver's scaling run); lsdsort_u32_ex(..., 2) must answer LSDSORT_ERR_NO_DEVICE here, not hang.
// Built with hipcc (host code only) and run by tests/test_cpp_harness.py.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <random>
#include <vector>

#include "lsdsort.hpp"

#define HIP_OK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main()
{
    try {
        const size_t n = (1u << 21) + 77;
        std::vector<uint32_t> keys(n), expect;
        std::mt19937 gen(4242);
        for (auto& k : keys) k = (uint32_t)gen();
        expect = keys;
        std::sort(expect.begin(), expect.end());

        lsd::communicator comm(lsd::unique_id(), /*world*/ 1, /*rank*/ 0);
        if (comm.world() != 1 || comm.rank() != 0) return 1;
        const size_t cap = n + 1000;
        const size_t ws_bytes = comm.workspace_bytes(n, cap, 8);
        uint32_t *d_in = nullptr, *d_out = nullptr;
        void* d_ws = nullptr;
        HIP_OK(hipMalloc((void**)&d_in, n * 4));
        HIP_OK(hipMalloc((void**)&d_out, cap * 4));
        HIP_OK(hipMalloc(&d_ws, ws_bytes));
        HIP_OK(hipMemcpy(d_in, keys.data(), n * 4, hipMemcpyHostToDevice));
        hipStream_t stream;
        HIP_OK(hipStreamCreate(&stream));
        for (int rep = 0; rep < 3; rep++) {                 // the communicator and the workspace serve every step
            uint64_t matrix[1] = {0};
            // the last repetition takes the sampled-splitter rule (one more gather and host wait; no thresholds for one rank)
            const auto s = comm.sort_device(d_in, n, d_out, cap, d_ws, ws_bytes, 8, stream, matrix,
                                            rep == 2 ? LSDSORT_PARTITION_SPLITTERS : LSDSORT_PARTITION_MSB);
            HIP_OK(hipStreamSynchronize(stream));
            if (s.n != n || s.global_offset != 0 || matrix[0] != n) { std::fprintf(stderr, "slice %zu at %llu, matrix %llu\n", s.n, (unsigned long long)s.global_offset, (unsigned long long)matrix[0]); return 1; }
            std::vector<uint32_t> got(n), in_after(n);
            HIP_OK(hipMemcpy(got.data(), d_out, n * 4, hipMemcpyDeviceToHost));
            HIP_OK(hipMemcpy(in_after.data(), d_in, n * 4, hipMemcpyDeviceToHost));
            if (got != expect) { std::fprintf(stderr, "sharded sort of one rank differs from std::sort (rep %d)\n", rep); return 1; }
            if (in_after != keys) { std::fprintf(stderr, "the input shard was modified\n"); return 1; }
        }
        // a rank that cannot hold what it would receive: every rank is told before anything is exchanged
        {
            size_t n_out = 0;
            uint64_t off = 0;
            lsdsort_comm* raw = nullptr;
            const auto id = lsd::unique_id();
            lsd::check(lsdsort_comm_create(id.bytes, 1, 0, &raw), "lsdsort_comm_create");
            const int st = lsdsort_sharded_u32_device(raw, d_in, n, d_out, n - 1, &n_out, &off, nullptr, d_ws, ws_bytes, 8, stream);
            HIP_OK(hipStreamSynchronize(stream));
            lsdsort_comm_destroy(raw);
            if (st != LSDSORT_ERR_CAPACITY || n_out != n) { std::fprintf(stderr, "expected LSDSORT_ERR_CAPACITY, got %d\n", st); return 1; }
        }
        // the one-process form: more GPUs than this box has is an error, not a hang
        {
            std::vector<uint32_t> h = keys;
            int count = 0;
            HIP_OK(hipGetDeviceCount(&count));
            if (count < 2) {
                const int st = lsdsort_u32_ex(h.data(), h.size(), 8, 2);
                if (st != LSDSORT_ERR_NO_DEVICE) { std::fprintf(stderr, "lsdsort_u32_ex(.., 2) on %d GPU(s): %d\n", count, st); return 1; }
            } else {
                lsd::sort(h.data(), h.size(), 8, 2);       // a multi-GPU box: the real thing
                if (h != expect) { std::fprintf(stderr, "two-GPU sort differs from std::sort\n"); return 1; }
            }
        }
        hipFree(d_ws); hipFree(d_out); hipFree(d_in);
    } catch (const std::exception& e) {
        std::fprintf(stderr, "exception: %s\n", e.what());
        return 1;
    }
    std::printf("sharded cpp test ok\n");
    return 0;
}
