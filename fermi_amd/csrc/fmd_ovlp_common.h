// fmd_ovlp_common.h -- helpers shared by the overlap-discovery kernels
#pragma once
#include "fmd_internal.h"

#define NONE64 (~0ull)
#define FMD_SZ_MASK 0xffffffffffffull

__device__ __forceinline__ int comp6(int c) { return (c >= 1 && c <= 4) ? 5 - c : c; }

template <class T>
__device__ __forceinline__ T sel6(int c, T a0, T a1, T a2, T a3, T a4, T a5)
{
    T r = a0;
    r = c == 1 ? a1 : r; r = c == 2 ? a2 : r; r = c == 3 ? a3 : r; r = c == 4 ? a4 : r; r = c == 5 ? a5 : r;
    return r;
}

// queue refill shared by the persistent kernels: returns the item index for lanes that asked
__device__ __forceinline__ size_t fmd_queue_take(uint32_t *queue, bool want)
{
    const uint64_t m = __ballot(want);
    if (m == 0) return (size_t)-1;
    uint32_t first = 0;
    if (fmd_lane() == 0) first = atomicAdd(queue, (uint32_t)__popcll(m));
    first = (uint32_t)__builtin_amdgcn_readfirstlane((int)first);
    return want ? (size_t)first + __popcll(m & ((1ull << fmd_lane()) - 1)) : (size_t)-1;
}

__device__ __forceinline__ void load_entry(const fmd_intv_t *e, uint64_t &x0, uint64_t &x1, uint64_t &sz, uint64_t &info)
{
    const uint4 *q = (const uint4 *)e;
    const uint4 a = q[0], b = q[1];
    x0 = (uint64_t)a.y << 32 | a.x; x1 = (uint64_t)a.w << 32 | a.z;
    sz = (uint64_t)b.y << 32 | b.x; info = (uint64_t)b.w << 32 | b.z;
}
__device__ __forceinline__ void store_entry(fmd_intv_t *e, uint64_t x0, uint64_t x1, uint64_t sz, uint64_t info)
{
    uint4 *q = (uint4 *)e;
    q[0] = make_uint4((uint32_t)x0, (uint32_t)(x0 >> 32), (uint32_t)x1, (uint32_t)(x1 >> 32));
    q[1] = make_uint4((uint32_t)sz, (uint32_t)(sz >> 32), (uint32_t)info, (uint32_t)(info >> 32));
}

// work list of strand indices for one get_nei kernel class: [0] = count, then the indices
struct FmdOvlClasses {
    uint32_t *n16, *l16;     // strands with <= 16 candidates and small intervals
    uint32_t *n32, *l32;     // 17..32 candidates
    uint32_t *nslow, *lslow; // everything else, plus strands the group kernels hand back
};
