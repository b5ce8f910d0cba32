/*
 * lsd_oracle.c -- CPU restatement of the reference's LSD radix sort path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only
 * as the checker.  The product (lsdradixsort_amd/csrc) never links or calls it.
 *
 * Every function states the reference lines (relative to /root/reference/) it follows.
 * ".cu" = LSDRadixSort/LSDRadixSort.cu.  The text below is a fresh restatement of the
 * algorithm, not a copy: same arithmetic, same visit order, same outputs.
 *
 * Parity pin: this restatement is checked (tests/test_oracle.py) against
 *   - oracle/_ref/libref_lsd.so, the reference's own CPU functions compiled in place from
 *     /root/reference (oracle/Makefile, target `ref`), when that tree is present, and
 *   - tests/golden/ fixtures generated from that same reference build + std::sort.
 */
#include <stdint.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

#define ORACLE_API __attribute__((visibility("default")))

/* a1 -- GET_R_BITS(n, r, i), LSDRadixSort/Utils.h:22: the i-th group of r bits of n. */
static inline uint32_t digit_of(uint32_t key, int r, int bit_group)
{
    return (key >> (bit_group * r)) & ((1u << r) - 1u);
}

ORACLE_API uint32_t oracle_get_r_bits(uint32_t key, int r, int bit_group)
{
    return digit_of(key, r, bit_group);
}

/*
 * a9 -- one counting-sort pass, .cu:25-54.
 *   .cu:27      clear the 2^r counters
 *   .cu:30-35   count digits front to back
 *   .cu:38-41   inclusive running sum over the counters
 *   .cu:44-50   walk the input BACKWARDS, pre-decrement the digit's counter, place the key
 *               there (this is what makes the pass stable)
 *   .cu:53      copy the output over the input
 * The reference indexes with `int`; counts here are size_t so the oracle can also serve
 * sizes the reference cannot (n >= 2^31), with identical results where both are defined.
 */
ORACLE_API void oracle_lsd_pass(uint32_t* in, uint32_t* out, size_t count, uint32_t* histogram,
                                int r, int bit_group)
{
    const size_t bins = (size_t)1 << r;
    for (size_t b = 0; b < bins; b++) histogram[b] = 0;
    for (size_t j = 0; j < count; j++) histogram[digit_of(in[j], r, bit_group)] += 1;
    for (size_t b = 1; b < bins; b++) histogram[b] += histogram[b - 1];
    for (size_t j = count; j-- > 0;) {
        const uint32_t key = in[j];
        const uint32_t d = digit_of(key, r, bit_group);
        histogram[d] -= 1;
        out[histogram[d]] = key;
    }
    memcpy(in, out, count * sizeof(uint32_t));
}

/* a9 -- the full sort, .cu:62-69: 32/r passes, least significant group first.  Both `in`
 * and `out` hold the sorted keys afterwards (the per-pass copy-back at .cu:53). */
ORACLE_API void oracle_lsd_sort(uint32_t* in, uint32_t* out, size_t count, uint32_t* histogram, int r)
{
    const int groups = 32 / r;
    for (int g = 0; g < groups; g++) oracle_lsd_pass(in, out, count, histogram, r, g);
}

/*
 * Key/value form of the same pass.  The reference sorts keys only;
 * this extends .cu:25-54 in the one obvious way -- the payload follows its key through the
 * same backward stable placement -- and is itself checked against std::stable_sort by key
 * (oracle_std_stable_sort_pairs in std_sort.cpp).
 */
ORACLE_API void oracle_lsd_pass_pairs(uint32_t* kin, uint32_t* vin, uint32_t* kout, uint32_t* vout,
                                      size_t count, uint32_t* histogram, int r, int bit_group)
{
    const size_t bins = (size_t)1 << r;
    for (size_t b = 0; b < bins; b++) histogram[b] = 0;
    for (size_t j = 0; j < count; j++) histogram[digit_of(kin[j], r, bit_group)] += 1;
    for (size_t b = 1; b < bins; b++) histogram[b] += histogram[b - 1];
    for (size_t j = count; j-- > 0;) {
        const uint32_t d = digit_of(kin[j], r, bit_group);
        histogram[d] -= 1;
        kout[histogram[d]] = kin[j];
        vout[histogram[d]] = vin[j];
    }
    memcpy(kin, kout, count * sizeof(uint32_t));
    memcpy(vin, vout, count * sizeof(uint32_t));
}

ORACLE_API void oracle_lsd_sort_pairs(uint32_t* kin, uint32_t* vin, uint32_t* kout, uint32_t* vout,
                                      size_t count, uint32_t* histogram, int r)
{
    const int groups = 32 / r;
    for (int g = 0; g < groups; g++)
        oracle_lsd_pass_pairs(kin, vin, kout, vout, count, histogram, r, g);
}

/* Exclusive running sum in place, .cu:128-139 (PrefixSum): {3,1,4,1,5} -> {0,3,4,8,9}. */
ORACLE_API void oracle_exclusive_scan(uint32_t* a, size_t count)
{
    uint32_t running = 0;
    for (size_t i = 0; i < count; i++) {
        const uint32_t v = a[i];
        a[i] = running;
        running += v;
    }
}

/*
 * a2 -- per-tile digit counts, block-major h[G][H], following BuildHistogramsCPU .cu:643-658
 * (and what BuildHistogramsKernel .cu:660-702 leaves in global memory).  The reference
 * accumulates into whatever h held (.cu:655, a latent harness bug); this restatement zeroes
 * h first, which is what the kernel's output actually is.  A ragged last tile (count not a
 * multiple of tile) simply counts fewer keys; the reference never runs that case.
 */
ORACLE_API void oracle_tile_histograms(const uint32_t* a, uint32_t* h, size_t count, size_t tile,
                                       int r, int bit_group)
{
    const size_t bins = (size_t)1 << r;
    const size_t tiles = (count + tile - 1) / tile;
    memset(h, 0, tiles * bins * sizeof(uint32_t));
    for (size_t i = 0; i < count; i++) h[(i / tile) * bins + digit_of(a[i], r, bit_group)] += 1;
}

/* a4 -- local offsets: exclusive scan of each tile's H counts, in place on a [G][H] array.
 * BlockPrefixSumKernel launched with H threads per block, .cu:869 (def .cu:180-207). */
ORACLE_API void oracle_local_offsets(uint32_t* h, size_t tiles, int r)
{
    const size_t bins = (size_t)1 << r;
    for (size_t t = 0; t < tiles; t++) oracle_exclusive_scan(h + t * bins, bins);
}

/*
 * a3+a5+a6 -- global offsets.  The reference copies the histograms (.cu:862), transposes
 * them to digit-major [H][G] (.cu:885), runs one flat exclusive scan over all G*H words
 * (GPUPrefixSum .cu:887 -> .cu:286-302) and transposes back (.cu:894).  Net effect restated
 * directly: g[t][d] = (number of keys with digit < d anywhere) + (number with digit d in
 * tiles before t), written block-major.  `hist` is the [G][H] counts, `g` the output.
 */
ORACLE_API void oracle_global_offsets(const uint32_t* hist, uint32_t* g, size_t tiles, int r)
{
    const size_t bins = (size_t)1 << r;
    uint32_t running = 0;
    for (size_t d = 0; d < bins; d++) {
        for (size_t t = 0; t < tiles; t++) {
            g[t * bins + d] = running;
            running += hist[t * bins + d];
        }
    }
}

/*
 * a7 -- rank-and-scatter, LSDRadixSortKernel .cu:795-837: each tile is stably sorted on the
 * current digit (SMEMLSDBinaryRadixSort .cu:373-402 is r stable one-bit splits = a stable
 * sort on the r-bit digit), then the key at sorted position `tid` goes to
 *     dst = tid - local[d] + global[d]                              (.cu:833)
 * `local`/`global` are the [G][H] arrays from the two functions above.
 */
ORACLE_API void oracle_rank_scatter(const uint32_t* a, uint32_t* b, const uint32_t* local,
                                    const uint32_t* global, size_t count, size_t tile, int r,
                                    int bit_group)
{
    const size_t bins = (size_t)1 << r;
    uint32_t* sorted = (uint32_t*)malloc(tile * sizeof(uint32_t));
    uint32_t* cursor = (uint32_t*)malloc(bins * sizeof(uint32_t));
    const size_t tiles = (count + tile - 1) / tile;
    for (size_t t = 0; t < tiles; t++) {
        const size_t base = t * tile;
        const size_t len = (count - base < tile) ? count - base : tile;
        const uint32_t* l = local + t * bins;
        const uint32_t* g = global + t * bins;
        /* stable in-tile sort on the digit, using the tile's own local offsets as bases */
        for (size_t d = 0; d < bins; d++) cursor[d] = l[d];
        for (size_t i = 0; i < len; i++) {
            const uint32_t d = digit_of(a[base + i], r, bit_group);
            sorted[cursor[d]++] = a[base + i];
        }
        for (size_t tid = 0; tid < len; tid++) {
            const uint32_t d = digit_of(sorted[tid], r, bit_group);
            const uint32_t dst = (uint32_t)((int64_t)tid - (int64_t)l[d] + (int64_t)g[d]);
            b[dst] = sorted[tid];
        }
    }
    free(cursor);
    free(sorted);
}

/*
 * a8 -- the staged GPU driver's data flow, GPULSDRadixSort .cu:839-910, on the CPU: per
 * pass histogram -> local offsets -> global offsets -> rank-and-scatter, ping-pong a/b.
 * Result lands in `a` when the pass count is even (.cu:905, .cu:1005), as the reference
 * relies on; the function returns the buffer that holds it (always `a` for r in {1,2,4,8,16}).
 * `h` must hold 2*G*H words (local offsets, global offsets).
 */
ORACLE_API uint32_t* oracle_staged_sort(uint32_t* a, uint32_t* b, uint32_t* h, size_t count,
                                        size_t tile, int r)
{
    const size_t bins = (size_t)1 << r;
    const size_t tiles = (count + tile - 1) / tile;
    uint32_t* local = h;
    uint32_t* global = h + tiles * bins;
    const int groups = 32 / r;
    for (int grp = 0; grp < groups; grp++) {
        oracle_tile_histograms(a, local, count, tile, r, grp);
        oracle_global_offsets(local, global, tiles, r);
        oracle_local_offsets(local, tiles, r);
        oracle_rank_scatter(a, b, local, global, count, tile, r, grp);
        uint32_t* tmp = a; a = b; b = tmp;
    }
    return a;
}

/*
 * All 32/r digit histograms of the whole array in one read: out[g][d] = number of keys whose
 * g-th r-bit group equals d.  Equals the column sums of oracle_tile_histograms for each g,
 * i.e. the counters LSDRadixSortPass builds at .cu:30-35 for every pass at once (legal
 * because a pass permutes keys and never changes them).
 */
ORACLE_API void oracle_digit_histograms(const uint32_t* a, size_t count, int r, uint64_t* out)
{
    const size_t bins = (size_t)1 << r;
    const int groups = 32 / r;
    memset(out, 0, (size_t)groups * bins * sizeof(uint64_t));
    for (size_t i = 0; i < count; i++)
        for (int g = 0; g < groups; g++) out[(size_t)g * bins + digit_of(a[i], r, g)] += 1;
}

/*
 * Multi-GPU first step (new work; no reference counterpart): stable
 * partition of a shard by its top `msb_bits` bits.  counts[b] = keys in bucket b; `out`
 * holds bucket 0, then bucket 1, ... each in original order.  It is one counting-sort pass
 * (.cu:25-50 without the copy-back) on the digit (key >> (32 - msb_bits)).
 */
ORACLE_API void oracle_msb_partition(const uint32_t* in, uint32_t* out, size_t count, int msb_bits,
                                     uint64_t* counts)
{
    const size_t bins = (size_t)1 << msb_bits;
    const int shift = 32 - msb_bits;
    uint64_t* cursor = (uint64_t*)calloc(bins, sizeof(uint64_t));
    for (size_t b = 0; b < bins; b++) counts[b] = 0;
    if (msb_bits == 0) {
        counts[0] = count;
        memcpy(out, in, count * sizeof(uint32_t));
        free(cursor);
        return;
    }
    for (size_t i = 0; i < count; i++) counts[in[i] >> shift] += 1;
    uint64_t running = 0;
    for (size_t b = 0; b < bins; b++) { cursor[b] = running; running += counts[b]; }
    for (size_t i = 0; i < count; i++) out[cursor[in[i] >> shift]++] = in[i];
    free(cursor);
}

/* CheckArrays, LSDRadixSort/Utils.cpp:62-68, made reportable: index of the first mismatch,
 * or `count` when the arrays agree (the reference crashes instead). */
ORACLE_API size_t oracle_first_mismatch(const uint32_t* a, const uint32_t* b, size_t count)
{
    for (size_t i = 0; i < count; i++)
        if (a[i] != b[i]) return i;
    return count;
}
