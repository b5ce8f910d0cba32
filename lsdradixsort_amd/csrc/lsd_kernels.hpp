// lsd_kernels.hpp -- host-visible launch interface of the gfx950 kernels.
//
// The C-ABI layer (lsdsort_api.hip) sequences these; each launcher picks the template
// instantiation for (radix_bits, tile shape) and returns the hipError_t of the launch.
#pragma once
#include <hip/hip_runtime.h>
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

// Everything one rank-and-scatter launch needs.
struct PassParams {
    const uint32_t* in;
    uint32_t* out;
    const uint32_t* vals_in;   // null: keys only
    uint32_t* vals_out;
    uint32_t n;
    uint32_t shift;            // bit_group * radix_bits
    uint32_t num_tiles;
    // chained (onesweep) form
    const uint32_t* digit_base;  // [2^R] exclusive scan of the global digit counts of this pass
    uint32_t* status;            // [num_tiles][2^R] tile-status words (lsd_device.hpp)
    uint32_t* tile_counter;      // dynamic tile id dispenser for this pass (zeroed)
    uint32_t parity;             // pass parity for the status codes
    // staged form
    const uint32_t* global_off;  // [num_tiles][2^R] digit-major exclusive scan, block-major
    uint32_t* fault;             // workspace fault word
};

// Stage 3.  chained=true: tile bases by decoupled look-back; false: read from global_off.
hipError_t launch_rank_scatter(int radix_bits, const TileShape& shape, bool chained,
                               const PassParams& p, hipStream_t stream);

// Stage 1, onesweep: all `groups` digit histograms (digit g at bit shift0 + g*radix_bits) in
// one read; hist[g][d] must be zero on entry.
hipError_t launch_digit_histograms(int radix_bits, int groups, uint32_t shift0, const uint32_t* keys,
                                   uint32_t n, uint32_t* hist, hipStream_t stream);

// Stage 2, onesweep: base[g][d] = exclusive scan over d of hist[g][d], for every group.
hipError_t launch_scan_digit_counts(int radix_bits, int groups, const uint32_t* hist, uint32_t* base,
                                    hipStream_t stream);

// Stage 1, staged: hist[t][d] per tile (BuildHistogramsKernel, .cu:660-702).
hipError_t launch_tile_histograms(int radix_bits, const TileShape& shape, const uint32_t* keys,
                                  uint32_t n, uint32_t shift, uint32_t* hist, hipStream_t stream);

// Stage 2, staged: local[t][d] (per-tile exclusive scan) and global[t][d] (digit-major
// exclusive scan) from hist[t][d]; either output may be null.  `scratch` holds
// tile_offsets_scratch_words(tiles, radix_bits) words.
size_t tile_offsets_scratch_words(size_t tiles, int radix_bits);
hipError_t launch_tile_offsets(int radix_bits, const uint32_t* hist, uint32_t* local, uint32_t* global,
                               uint32_t tiles, uint32_t* scratch, hipStream_t stream);

// counts64[b] = hist32[b], b < bins (multi-GPU bucket sizes as uint64).
hipError_t launch_widen_counts(const uint32_t* hist32, uint64_t* counts64, int bins, hipStream_t stream);

// *out = value (stream-ordered).
hipError_t launch_store_u64(uint64_t* out, uint64_t value, hipStream_t stream);

}  // namespace lsd
