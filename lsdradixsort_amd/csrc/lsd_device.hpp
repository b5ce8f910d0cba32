// lsd_device.hpp -- device-side building blocks shared by the gfx950 kernels.
//
// Wavefront = 64 lanes everywhere (CDNA4).  All arithmetic is unsigned 32-bit, like the
// reference's.  Nothing here is derived from the reference's CUDA
// kernels; the file:line citations say which reference stage a piece stands in for.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace lsd {

constexpr int kWave = 64;

// LDS-qualified scalar types (address space 3), for pointers that must stay ds_* accesses.
typedef __attribute__((address_space(3))) uint32_t lds_u32;
typedef __attribute__((address_space(3))) uint64_t lds_u64;

// ---- tile-status words of the chained scan --------------------------------------------
// One 32-bit word carries value and state together, so a single relaxed agent-scope store
// publishes it and a single relaxed agent-scope load observes it (no fences: the data is the
// flag).  State = the low two bits:  0 = stale (not written in this pass), 1 = tile aggregate,
// 2 = inclusive prefix.  A sort keeps TWO status arrays: a pass works in one while its workgroups
// zero the other for the pass after it (rank_scatter.hpp, clear_next; the sort's opening memset
// covers the first), so every word a pass reads is either zero or was written by THAT pass -- no
// word outlives its pass (a single never-cleared, parity-coded array is
// what let round 1's prototype read an old prefix as a current one).  The codes still take a
// `parity` argument from that scheme; every launch passes 0.
__device__ __forceinline__ constexpr uint32_t code_aggregate(uint32_t parity) { return parity ? 3u : 1u; }
__device__ __forceinline__ constexpr uint32_t code_prefix(uint32_t parity) { return parity ? 0u : 2u; }
__device__ __forceinline__ constexpr uint32_t code_stale(uint32_t parity) { return parity ? 2u : 0u; }

__device__ __forceinline__ uint32_t load_status(const uint32_t* p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void store_status(uint32_t* p, uint32_t v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// a1 -- GET_R_BITS (LSDRadixSort/Utils.h:22) with the shift precomputed: one v_bfe_u32.
template <int R>
__device__ __forceinline__ uint32_t digit_at(uint32_t key, uint32_t shift)
{
    return __builtin_amdgcn_ubfe(key, shift, (uint32_t)R);
}

__device__ __forceinline__ uint32_t lane_index()
{
    return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
}

// number of set bits of `m` below this lane, plus `base`
__device__ __forceinline__ uint32_t mbcnt_add(uint64_t m, uint32_t base)
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, base));
}

__device__ __forceinline__ uint32_t popc64_add(uint64_t m, uint32_t base)
{
    return __builtin_popcount((uint32_t)m) + __builtin_popcount((uint32_t)(m >> 32)) + base;
}

// Peers of this lane: the 64-bit mask of lanes whose R-bit digit equals ours, from R
// wave-wide ballots.  All 64 lanes must be active.
template <int R>
__device__ __forceinline__ uint64_t match_ballot(uint32_t d)
{
    uint64_t m = ~0ull;
#pragma unroll
    for (int b = 0; b < R; b++) {
        const bool bit = (d >> b) & 1u;
        const uint64_t bal = __ballot(bit);
        m &= bit ? bal : ~bal;
    }
    return m;
}

// Inclusive scan across the 64 lanes of a wave, on the DPP network (no LDS round trips): row_shr 1/2/4/8 scans the four
// rows of 16 lanes, row_bcast15 carries row 0's total into row 1 and row 2's into row 3, row_bcast31 carries the first
// half's total into the second (the sequence LLVM's atomic optimiser emits for wave64 on GFX9-family targets).
// -DLSD_SCAN_SHFL builds the ds_bpermute (__shfl_up) form this replaced (A/B in DESIGN.md section 4.8).
__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v, uint32_t lane)
{
#ifdef LSD_SCAN_SHFL
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
        const uint32_t up = __shfl_up(v, off, kWave);
        if (lane >= (uint32_t)off) v += up;
    }
    return v;
#else
    (void)lane;
    // update_dpp(old, src, dpp_ctrl, row_mask, bank_mask, bound_ctrl): lanes without a source keep `old` = 0
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);   // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);   // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);   // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);   // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);   // row_bcast:15 -> rows 1 and 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);   // row_bcast:31 -> rows 2 and 3
    return v;
#endif
}

// Number of LDS replicas of a small histogram so that 64 lanes do not pile onto a handful
// of words (2^R < 64).  Lane l uses replica l % copies.
template <int R>
__host__ __device__ constexpr int hist_copies()
{
    return (1 << R) >= 64 ? 1 : ((1 << R) >= 16 ? 8 : 32);
}

}  // namespace lsd
