// fmd_pair.hip -- the two-base blocks of an index (FmdIndexView::pair, fmd_wave.h): built on the device from the rank blocks, on first use.
//
// Below min_match a walk pushes nothing (overlap_intv, unitig.c:47-59: a candidate needs depth >= min), so all it wants from a step is where its row
// and its interval stand one base further -- and that composes: two steps are ONE look-up in a structure that holds, per position p, the PAIR
// (BWT[p], BWT[LF(p)]) and pair counts.  A 128-byte random line costs this memory system what a 64-byte one does (profiles/r6_probe: 48 G lines/s
// either way), so the pair block takes the room it needs: a 128-byte block every 32 positions, 32 bits per symbol beside the 8 of the rank blocks.  Results are the same bits: the pair
// step is two exact LF / extension steps.
#include "fmd_kernel_common.h"
#include "fmd_prim.h"
#include <stdlib.h>

static inline unsigned pnblk(uint64_t n, unsigned t)
{
    const uint64_t b = (n + t - 1) / t, cap = (1ull << 31) / t;
    return (unsigned)(b < cap ? (b ? b : 1) : cap);
}

// symbol at position q, straight from the rank blocks in HBM
__device__ __forceinline__ int pair_sym_at(const uint4 *__restrict__ blocks, uint64_t q)
{
    const uint4 v = blocks[(q >> 6) * FMD_BLK_U4 + ((q >> 5) & 1)];
    const uint32_t bit = (uint32_t)q & 31;
    return (int)(((v.x >> bit) & 1) | ((v.y >> bit) & 1) << 1 | ((v.z >> bit) & 1) << 2);
}
// absolute count of symbol c (1..4) before block b, from the block's meta words (fmd_wave.h)
__device__ __forceinline__ uint64_t pair_abs(const uint4 *__restrict__ blk, int c)
{
    const uint4 mv = blk[3];
    const uint32_t lo = c == 1 ? blk[1].w : c == 2 ? blk[2].w : c == 3 ? mv.x : mv.y;
    const uint32_t hi = c < 4 ? (mv.z >> (8 * c)) & 0xff : mv.w & 0xff;
    return (uint64_t)hi << 32 | lo;
}

// One wave per RANK block b, lane i = position 64 b + i: the two pair blocks 2 b and 2 b + 1 (32 own positions each).  The positions of a rank block with
// the same first symbol c map under LF to CONSECUTIVE rows (cnt[c] + count of c before the block + rank among the block's c's), so their second symbols
// are a short run of BWT read in order.  Writes chunk 0 (the own positions) of both pair blocks and their sixteen pair counts (one byte each,
// pc[pair * n_pblocks + pblock]).
__global__ __launch_bounds__(256) void k_pair_planes(FmdIndexView ix, uint64_t n_rblocks, uint64_t n_pblocks, uint4 *__restrict__ pair, uint8_t *__restrict__ pc)
{
    const int lane = threadIdx.x & 63;
    const uint64_t w0 = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), wstep = (uint64_t)gridDim.x * (blockDim.x >> 6);
    for (uint64_t b = w0; b < n_rblocks; b += wstep) {
        const uint4 *blk = ix.blocks + b * FMD_BLK_U4;
        const uint64_t p = b * 64 + (uint64_t)lane;
        const uint4 v = blk[lane >> 5];
        const uint32_t bit = (uint32_t)lane & 31;
        int c1 = (int)(((v.x >> bit) & 1) | ((v.y >> bit) & 1) << 1 | ((v.z >> bit) & 1) << 2);
        if (p >= ix.n_sym) c1 = 0;
        int c2 = 0;
        uint64_t m1[5] = {0, 0, 0, 0, 0};
#pragma unroll
        for (int c = 1; c <= 4; ++c) m1[c] = __ballot(c1 == c);
        if (c1 >= 1 && c1 <= 4) {
            const uint64_t mc = c1 == 1 ? m1[1] : c1 == 2 ? m1[2] : c1 == 3 ? m1[3] : m1[4];
            const uint64_t q = ix.cnt[c1] + pair_abs(blk, c1) + (uint64_t)fmd_below(mc);     // LF(p) = cnt[c] + rank_c(p) - 1, rank inclusive
            c2 = pair_sym_at(ix.blocks, q);
        }
        const uint64_t s0 = __ballot(c2 & 1), s1 = __ballot(c2 & 2), s2 = __ballot(c2 & 4);
        if (lane < 2 && 2 * b + (uint64_t)lane < n_pblocks) {   // pair block 2 b + lane: its own chunk = chunk `lane` of the rank block
            uint4 *o = pair + (2 * b + (uint64_t)lane) * FMD_PAIR_U4;
            const uint4 r = blk[lane];
            const uint32_t sh = 32u * (uint32_t)lane;
            uint4 a = o[0], bb = o[3];
            a.x = r.x; a.y = r.y; a.z = r.z; a.w = (uint32_t)(s0 >> sh);
            bb.x = (uint32_t)(s1 >> sh); bb.y = (uint32_t)(s2 >> sh);
            if (b * 64 + 32u * (uint64_t)lane >= ix.n_sym) { a.x = a.y = a.z = 0; }   // positions past the end read as '$' with no second symbol
            o[0] = a; o[3] = bb;
        }
        // the sixteen pair counts of each half's 32 positions
        const uint64_t t0 = ~s2 & ~s1 & s0, t1 = ~s2 & s1 & ~s0, t2 = ~s2 & s1 & s0, t3 = s2 & ~s1 & ~s0;   // second symbol = A, C, G, T
        if (lane < 32) {
            const int pr = lane & 15, half = lane >> 4, a = pr >> 2, bq = pr & 3;
            const uint64_t ma = a == 0 ? m1[1] : a == 1 ? m1[2] : a == 2 ? m1[3] : m1[4];
            const uint64_t mb = bq == 0 ? t0 : bq == 1 ? t1 : bq == 2 ? t2 : t3;
            const uint32_t mm = (uint32_t)((ma & mb) >> (32 * half));
            if (2 * b + (uint64_t)half < n_pblocks) pc[(uint64_t)pr * n_pblocks + 2 * b + (uint64_t)half] = (uint8_t)__builtin_popcount(mm);
        }
    }
}
// the look-ahead chunks (positions 32 .. 95 of block p = the own chunks of blocks p + 1 and p + 2), planes only
__global__ void k_pair_lookahead(uint4 *__restrict__ pair, uint64_t n_pblocks)
{
    for (uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; b < n_pblocks; b += (uint64_t)gridDim.x * blockDim.x) {
        uint4 *o = pair + b * FMD_PAIR_U4;
#pragma unroll
        for (int j = 1; j <= 2; ++j) {
            uint4 a = make_uint4(0, 0, 0, 0), bb = make_uint4(0, 0, 0, 0);
            if (b + (uint64_t)j < n_pblocks) { a = pair[(b + (uint64_t)j) * FMD_PAIR_U4]; bb = pair[(b + (uint64_t)j) * FMD_PAIR_U4 + 3]; }
            uint4 x = o[j], y = o[3 + j];
            x.x = a.x; x.y = a.y; x.z = a.z; x.w = a.w; y.x = bb.x; y.y = bb.y;
            o[j] = x; o[3 + j] = y;
        }
    }
}
// count word t (0..13) of a pair block lives at: u4[3 + j].z / .w for t = 2j, 2j + 1 (j < 3), u4[6 + (t - 6) / 4] component (t - 6) % 4 beyond
__device__ __forceinline__ uint32_t *pair_cw(uint4 *blk, int t)
{
    uint32_t *w = (uint32_t *)blk;
    return t < 6 ? w + 4 * (3 + (t >> 1)) + 2 + (t & 1) : w + 4 * 6 + (t - 6);
}
// acc[b] = pairs `pr` before block b (absolute): the 28-bit field of every block, and the superblock's row of the table
__global__ void k_pair_counts(uint4 *__restrict__ pair, uint64_t n_blocks, const uint64_t *__restrict__ acc, int pr, uint64_t k2, unsigned long long *__restrict__ tab)
{
    for (uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; b < n_blocks; b += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t sb = b >> FMD_PAIR_SB_SHIFT, base = acc[sb << FMD_PAIR_SB_SHIFT];
        const uint32_t rel = (uint32_t)(acc[b] - base);          // < 2^28: a superblock has 2^28 positions
        if ((b & ((1ull << FMD_PAIR_SB_SHIFT) - 1)) == 0) tab[sb * 16 + (uint64_t)pr] = k2 + base;
        uint4 *blk = pair + b * FMD_PAIR_U4;
        const int bitpos = 28 * pr, t = bitpos >> 5, sh = bitpos & 31;
        uint32_t *w0 = pair_cw(blk, t);
        *w0 = (*w0 & ~(0x0fffffffu << sh)) | rel << sh;
        if (sh > 4) { uint32_t *w1 = pair_cw(blk, t + 1); *w1 = (*w1 & ~(0x0fffffffu >> (32 - sh))) | rel >> (32 - sh); }
    }
}
// K2[c1][c2] = cnt[c2] + #{c2 in BWT[0, cnt[c1])}
__global__ void k_pair_k2(FmdIndexView ix, unsigned long long *__restrict__ k2)
{
    const int i = threadIdx.x;
    if (i >= 16) return;
    const int c1 = (i >> 2) + 1, c2 = (i & 3) + 1;
    const uint64_t pos = ix.cnt[c1];          // count c2 in [0, pos): rank inclusive of pos - 1 (pos >= cnt[1] >= 1 in any index with a sequence)
    uint64_t r = 0;
    if (pos) { uint32_t b, o; fmd_split(pos - 1, b, o); r = fmd_block_rank1(ix.blocks + (size_t)b * FMD_BLK_U4, 0, o + 1, c2, b); }
    k2[i] = ix.cnt[c2] + r;
}

struct PairWiden { __host__ __device__ uint64_t operator()(uint8_t v) const { return (uint64_t)v; } };

// Build the two-base blocks of `h` (idempotent): 32 bits per symbol beside the index's 8, ~1 s per 10^10 symbols.  They pay where an index serves many
// passes (pass 1 of the sorted overlap job 47.8 -> 36.3 ms per 10^8 strands, profiles/r6_pair) and never within ONE pass, so nothing builds them unasked:
// force = fmd_dev_build_pairs (a caller that keeps the index), FMD_PAIR=1 the same from the first sorted job of any caller (A/B, tests), FMD_PAIR=0 never.
// Built only where they fit with room for a job beside them (force = 2 / FMD_PAIR=2: whenever the allocation succeeds).  FMD_OK either way: h->pair says.
int fmd_pairs_ensure(fmd_dev *h, int force)
{
    if (h->pair) return FMD_OK;
    const char *e = getenv("FMD_PAIR");
    if (e && atoi(e) == 0) return FMD_OK;
    const int want = force > 0 ? force : (e ? atoi(e) : 0);
    if (want < 1 || h->pair_tried >= want) return FMD_OK;
    h->pair_tried = want;
    if (hipSetDevice(h->device) != hipSuccess) { (void)hipGetLastError(); return FMD_OK; }
    const uint64_t nrb = h->n_blocks, nb = 2 * h->n_blocks;      // pair blocks: one per 32 positions
    if (nb >= 0xffffffffull) return FMD_OK;                      // (32-bit block numbers in the gather)
    const size_t need = (size_t)nb * FMD_PAIR_BYTES, temp = (size_t)nb * (16 + 8) + (64u << 20);
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); return FMD_OK; }
    // by default only where the job that follows still finds its room: the blocks, their construction, and as much again as the index itself
    if (want < 2 && need + temp + h->bytes + ((size_t)8 << 30) > free_b) return FMD_OK;
    uint4 *pair = nullptr; uint8_t *pc = nullptr; uint64_t *acc = nullptr; unsigned long long *tab = nullptr, *k2d = nullptr;
    const uint64_t n_sb = ((nb - 1) >> FMD_PAIR_SB_SHIFT) + 1;
    bool ok = hipMalloc((void **)&pair, need) == hipSuccess && hipMalloc((void **)&pc, (size_t)nb * 16) == hipSuccess && hipMalloc((void **)&acc, (size_t)nb * 8) == hipSuccess &&
              hipMalloc((void **)&tab, n_sb * 16 * 8) == hipSuccess && hipMalloc((void **)&k2d, 16 * 8) == hipSuccess;
    unsigned long long k2[16];
    void *tmp = nullptr; size_t tmp_bytes = 0;
    if (ok) {
        const FmdIndexView ix = fmd_view(h);
        ok = hipMemsetAsync(pair, 0, need, 0) == hipSuccess;
        k_pair_k2<<<1, 64>>>(ix, k2d);
        k_pair_planes<<<pnblk(nrb, 4), 256>>>(ix, nrb, nb, pair, pc);
        k_pair_lookahead<<<pnblk(nb, 256), 256>>>(pair, nb);
        ok = ok && hipMemcpy(k2, k2d, sizeof(k2), hipMemcpyDeviceToHost) == hipSuccess;
        rocprim::transform_iterator<const uint8_t *, PairWiden, uint64_t> in0(pc, PairWiden());
        ok = ok && fmd_exclusive_sum(nullptr, tmp_bytes, in0, acc, (size_t)nb, 0) == hipSuccess && hipMalloc(&tmp, tmp_bytes ? tmp_bytes : 16) == hipSuccess;
        for (int pr = 0; pr < 16 && ok; ++pr) {
            rocprim::transform_iterator<const uint8_t *, PairWiden, uint64_t> in(pc + (size_t)pr * nb, PairWiden());
            ok = fmd_exclusive_sum(tmp, tmp_bytes, in, acc, (size_t)nb, 0) == hipSuccess;
            k_pair_counts<<<pnblk(nb, 256), 256>>>(pair, nb, acc, pr, k2[pr], tab);
        }
        ok = ok && hipDeviceSynchronize() == hipSuccess;
    }
    (void)hipGetLastError();
    hipFree(tmp); hipFree(pc); hipFree(acc); hipFree(k2d);
    if (!ok) { hipFree(pair); hipFree(tab); (void)hipGetLastError(); return FMD_OK; }
    h->pair = pair; h->pair_tab = tab;
    h->pair_bytes = need + n_sb * 128;
    h->bytes += h->pair_bytes;
    if (getenv("FMD_DEBUG_PAIR")) fprintf(stderr, "[M::fmd_pairs_ensure] two-base blocks: %.2f GB, %llu superblocks\n", need / 1e9, (unsigned long long)n_sb);
    return FMD_OK;
}

extern "C" int fmd_dev_build_pairs(fmd_dev_t *h, int *built)
{
    if (!h) return FMD_E_ARG;
    if (fmd_device_count() <= 0) return FMD_E_NODEV;
    const int rc = fmd_pairs_ensure(h, 1);
    if (built) *built = h->pair != nullptr;
    return rc;
}

// ---- self-check: the pair step against two single steps, for every row (tests; `chkbwt -r` calls it when the blocks exist)
__global__ void k_pair_check(FmdIndexView ix, unsigned long long *__restrict__ bad)
{
    for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < ix.n_sym; k += (uint64_t)gridDim.x * blockDim.x) {
        // two single LF steps
        uint32_t b, o;
        fmd_split(k, b, o);
        uint64_t r[6];
        const int c1 = fmd_block_rank6<true>(ix.blocks + (size_t)b * FMD_BLK_U4, 0, o + 1, r, b);
        int c2 = 0;
        uint64_t k2 = 0;
        if (c1 >= 1 && c1 <= 4) {
            const uint64_t k1 = ix.cnt[c1] + r[c1] - 1;
            fmd_split(k1, b, o);
            c2 = fmd_block_rank6<true>(ix.blocks + (size_t)b * FMD_BLK_U4, 0, o + 1, r, b);
            if (c2 >= 1 && c2 <= 4) k2 = ix.cnt[c2] + r[c2] - 1;
        }
        // the pair block's answer
        const uint64_t pb = k >> 5;
        const uint32_t off = (uint32_t)k & 31;
        const uint4 *pq = ix.pair + pb * FMD_PAIR_U4;
        const uint4 A = pq[0], B = pq[3];
        const uint32_t bit = off;
        const int d1 = (int)(((A.x >> bit) & 1) | ((A.y >> bit) & 1) << 1 | ((A.z >> bit) & 1) << 2);
        const int d2 = (int)(((A.w >> bit) & 1) | ((B.x >> bit) & 1) << 1 | ((B.y >> bit) & 1) << 2);
        bool good = d1 == c1 && (!(c1 >= 1 && c1 <= 4) || d2 == c2);
        // the look-ahead chunks repeat the own chunks of the next two blocks
        if (good && pb >= 1) { const uint4 L1 = ix.pair[(pb - 1) * FMD_PAIR_U4 + 1], M1 = ix.pair[(pb - 1) * FMD_PAIR_U4 + 4]; good = L1.x == A.x && L1.y == A.y && L1.z == A.z && L1.w == A.w && M1.x == B.x && M1.y == B.y; }
        if (good && pb >= 2) { const uint4 L2 = ix.pair[(pb - 2) * FMD_PAIR_U4 + 2], M2 = ix.pair[(pb - 2) * FMD_PAIR_U4 + 5]; good = L2.x == A.x && L2.y == A.y && L2.z == A.z && L2.w == A.w && M2.x == B.x && M2.y == B.y; }
        if (good && c1 >= 1 && c1 <= 4 && c2 >= 1 && c2 <= 4) {
            const int pr = 4 * (c1 - 1) + (c2 - 1);
            const uint32_t m = off == 31 ? ~0u : (1u << (off + 1)) - 1u;
            const uint32_t e1 = ((c1 & 1) ? A.x : ~A.x) & ((c1 & 2) ? A.y : ~A.y) & ((c1 & 4) ? A.z : ~A.z);
            const uint32_t e2 = ((c2 & 1) ? A.w : ~A.w) & ((c2 & 2) ? B.x : ~B.x) & ((c2 & 4) ? B.y : ~B.y);
            const uint32_t n = __builtin_popcount(e1 & e2 & m);
            const uint32_t *w = (const uint32_t *)pq;
            uint32_t cw[14];
            for (int t = 0; t < 6; ++t) cw[t] = w[4 * (3 + (t >> 1)) + 2 + (t & 1)];
            for (int t = 6; t < 14; ++t) cw[t] = w[4 * 6 + (t - 6)];
            const int bp = 28 * pr, t = bp >> 5, sh = bp & 31;
            uint32_t rel = cw[t] >> sh;
            if (sh > 4) rel |= cw[t + 1] << (32 - sh);
            rel &= 0x0fffffffu;
            const uint64_t got = ix.pair_tab[(pb >> FMD_PAIR_SB_SHIFT) * 16 + (uint64_t)pr] + rel + n - 1;
            good = got == k2;
        }
        if (!good && atomicAdd(bad, 1ull) == 0) bad[1] = k;
    }
}
extern "C" int fmd_dev_check_pairs(fmd_dev_t *h, uint64_t *n_bad, uint64_t *first_bad)
{
    if (!h || !n_bad) return FMD_E_ARG;
    if (!h->pair) return FMD_E_ARG;
    FMD_HIP_TRY(hipSetDevice(h->device));
    unsigned long long *bad = nullptr, host[2] = {0, 0};
    FMD_HIP_TRY(hipMalloc((void **)&bad, 16));
    FMD_HIP_TRY(hipMemset(bad, 0, 16));
    k_pair_check<<<h->n_cu * 16, 256>>>(fmd_view(h), bad);
    FMD_HIP_TRY(hipDeviceSynchronize());
    FMD_HIP_TRY(hipMemcpy(host, bad, 16, hipMemcpyDeviceToHost));
    hipFree(bad);
    *n_bad = host[0];
    if (first_bad) *first_bad = host[1];
    return FMD_OK;
}
