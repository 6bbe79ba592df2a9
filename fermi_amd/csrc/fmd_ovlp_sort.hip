// fmd_ovlp_sort.hip -- the order of a sorted overlap job (fmd_ovlp_sorted_dev, fmd_ovlp.hip).
//
// Two strands whose last bases lie d positions apart on the same strand of the genome walk through the same rank blocks, d steps
// apart: the interval of "g[a, e)" contains the interval of "g[a, e + d)" and both are a few dozen positions wide.  In id order
// (= the order the reads came off the sequencer) such strands are never in flight together, so each of those visits is a DRAM
// miss; next to each other in a wave, the second visit is an L2 hit.  Nobody knows the layout before the overlaps are found, but
// strands of one genomic window share k-mers: after the first FMD_WALK_SPLIT bases of every walk (k_ovl_walk<WALK_HEAD>) the
// strands are sorted by the MINIMIZER of those bases -- the 16-mer with the smallest hash -- with the offset of that 16-mer from
// the strand's end as the low bits of the key, so that strands sharing a minimizer are in genome order among themselves.
// 3.5 strands per key at 30x; measured on the 50 M-read set: 100 M strands in 269 ms in this order, 254 ms in true genome order,
// 367 ms in id order (profiles/r3_locality).
#include "fmd_prim.h"
#include "fmd_kernel_common.h"

#define PARK_K 16u                                   // bases per k-mer (2 bits each: one 32-bit word)
#define PARK_NK (FMD_WALK_SPLIT - PARK_K + 1)        // k-mers in the window (17: the offset takes 5 bits of the key)
static_assert(PARK_NK <= 32, "the offset of the minimizer takes the low five bits of the key");

__device__ __forceinline__ uint32_t park_hash(uint32_t v)
{
    v *= 0x9E3779B1u; v ^= v >> 15; v *= 0x85EBCA77u; v ^= v >> 13;
    return v;
}

// one thread per parked strand: key = hash of the minimizer (27 bits) | its offset (5 bits); strands that ended inside the head
// sort to the end (key ~0) -- the second pass skips them
__global__ void k_ovl_park_keys(size_t n, const FmdWalkPark *__restrict__ park, uint32_t *__restrict__ keys, uint32_t *__restrict__ vals)
{
    const size_t step = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += step) {
        const uint4 *p = (const uint4 *)(park + i);
        const uint4 a = p[0];
        uint32_t key = 0xffffffffu;
        if (((uint64_t)a.y << 32 | a.x) != ~0ull) {
            const uint4 hb = p[2];
            const uint32_t w[4] = {hb.x, hb.y, hb.z, hb.w};   // FmdWalkPark::bases: one nt6 code per 4 bits, last base of the strand first
            uint32_t v = 0, bad = 0, best = 0xffffffffu;
#pragma unroll
            for (uint32_t j = 0; j < FMD_WALK_SPLIT; ++j) {
                const uint32_t c = (w[j >> 3] >> (4 * (j & 7))) & 0xfu;
                v = v << 2 | ((c - 1u) & 3u);
                bad = bad << 1 | (uint32_t)(c < 1u || c > 4u);                          // a k-mer over an ambiguous base is nobody's minimizer
                if (j + 1 >= PARK_K && (bad & 0xffffu) == 0) {
                    const uint32_t hsh = (park_hash(v) & ~31u) | (j + 1 - PARK_K);
                    best = hsh < best ? hsh : best;
                }
            }
            key = best == 0xffffffffu ? 0xfffffffeu : best;
        }
        keys[i] = key;
        vals[i] = (uint32_t)i;
    }
}

#ifndef FMD_PARK_SORT_FROM_DEFAULT
#define FMD_PARK_SORT_FROM_DEFAULT 0
#endif
static unsigned park_sort_from(void)
{
    const char *e = getenv("FMD_PARK_SORT_FROM");
    const int v = e ? atoi(e) : FMD_PARK_SORT_FROM_DEFAULT;
    return v < 0 ? 0u : v > 31 ? 31u : (unsigned)v;
}

size_t fmd_park_sort_temp_bytes(size_t n)
{
    size_t tb = 0;
    if (fmd_sort_pairs(nullptr, tb, (const uint32_t *)nullptr, (uint32_t *)nullptr, (const uint32_t *)nullptr, (uint32_t *)nullptr, n, 0, 32, (hipStream_t)0) != hipSuccess) {
        (void)hipGetLastError();
        tb = 16 * n + (1u << 20);   // more than any version of the sort has asked for
    }
    return tb;
}

int fmd_park_sort(hipStream_t st, size_t n, const FmdWalkPark *park, uint32_t *keys_a, uint32_t *keys_b, uint32_t *vals_a, uint32_t *vals_b, void *tmp, size_t tmp_bytes)
{
    size_t blocks = (n + 255) / 256;
    if (blocks > (1u << 20)) blocks = 1u << 20;
    k_ovl_park_keys<<<(unsigned)blocks, 256, 0, st>>>(n, park, keys_a, vals_a);
    // All 32 bits: the order is for the caches only (any order gives the same rows), but dropping radix passes does not pay -- with the sort looking at
    // key bits b .. 31 the whole job took 254.4 (b = 0), 259.5 (8), 274.1 (12), 276.2 (16), 292.7 (20), 322.5 (24) ms per 10^8 strands
    // (profiles/r5_sortbits, FMD_PARK_SORT_FROM=b): already the five offset bits that put the strands of ONE minimizer in genome order are worth 2 %.
    FMD_HIP_TRY(fmd_sort_pairs(tmp, tmp_bytes, (const uint32_t *)keys_a, keys_b, (const uint32_t *)vals_a, vals_b, n, park_sort_from(), 32, st));
    return FMD_OK;
}
