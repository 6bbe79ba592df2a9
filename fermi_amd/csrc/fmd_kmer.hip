// fmd_kmer.hip -- the k-mer harvest of `fermi correct`: fm6_traverse (exact.c:141-171) +
// ec_collect (correct.c:35-87), 90 % of `correct`'s CPU time.
//
// The reference walks the k-mer trie depth-first with an explicit stack per suffix bucket.  The
// table it builds -- per bucket, key = the remaining bases + best next base, val = ratio/rest
// code -- does not depend on the visiting order (only khash insertion order does), so the GPU
// expands the trie LEVEL BY LEVEL instead: frontier(d) -> one backward extension per node on the
// wave engine -> children with enough occurrences appended to frontier(d+1) by wave-aggregated
// atomics.  No per-lane stacks, every lane busy, every step the same phase.  The k-mer travels in
// the node's info word as K = sum (base_d - 1) << 2d  (base_0 = rightmost base of the k-mer):
// bucket = K mod 4^SUF_LEN (fm6_traverse's index, exact.c:160) and key = (K >> 2*SUF_LEN) << 2 |
// best (correct.c:71-73).
#include <stdlib.h>
#include <string.h>
#include "fmd_prim.h"
#include "fmd_internal.h"
#include "fmd_kernel_common.h"


// counters in device memory: [0..63] frontier sizes per depth, [64] #output, [65] overflow flag,
// [66] cnt[0] (k-mers kept), [67] cnt[1] (informative ones), correct.c:64-69, [68] extensions
#define KM_OUT 64
#define KM_OVF 65
#define KM_CNT0 66
#define KM_CNT1 67
#define KM_EXT 68      // backward extensions done (= trie nodes expanded), all levels
#define KM_WORDS 72

// One backward extension per lane.  Narrow intervals (size <= 63: every level below ~log4(n)) take the
// child sizes from one 64-position window of the lane's block image(s); the absolute rank tk[c] is then
// computed only for the children that are kept (usually one) -- or not at all (k_kmer_emit needs sizes
// only).  Wide intervals: two six-symbol block ranks.
template <bool NEED_TK>
__device__ __forceinline__ void km_extend_back(const FmdIndexView &ix, uint4 *lds, bool active, uint64_t x0, uint64_t sz,
                                               uint64_t thr, uint64_t tk[6], uint64_t s[6])
{
    FmdRank2c r = fmd_wave_rank2_fetch_compact(ix, lds, active ? x0 - 1 : NONE64, active ? x0 - 1 + sz : NONE64);
    const bool narrow = active && sz <= 63 && !(r.two_phase && r.l_sep);
    uint64_t tl[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int c = 0; c < 6; ++c) { tk[c] = 0; s[c] = 0; }
    if (active && !narrow && r.hk) fmd_block_rank6<false>(r.bk, r.t, r.nk, tk, r.blk_k);
    fmd_wave_l_ready(ix, lds, r);
    if (narrow) {
        uint4 a, b, c;
        grp_window(r.bk, r.t, r.bl, r.tl, r.blk_k, r.blk_l, r.hk, r.l_sep, r.blk_k, r.nk - 1, a, b, c); // window at x0 = (x0 - 1) + 1
        const uint32_t sh = (uint32_t)x0 & 31;
        const uint64_t m = (1ull << (int)sz) - 1;
        const uint64_t X = win64(a.x, b.x, c.x, sh), Y = win64(a.y, b.y, c.y, sh), Z = win64(a.z, b.z, c.z, sh);
        const uint64_t lo = ~Z & m, hi = Z & ~Y & m;
        s[0] = __popcll(lo & ~Y & ~X); s[1] = __popcll(lo & ~Y & X); s[2] = __popcll(lo & Y & ~X); s[3] = __popcll(lo & Y & X);
        s[4] = __popcll(hi & ~X); s[5] = __popcll(hi & X);
        if (NEED_TK) {
            uint32_t todo = (s[1] >= thr ? 2u : 0u) | (s[2] >= thr ? 4u : 0u) | (s[3] >= thr ? 8u : 0u) | (s[4] >= thr ? 16u : 0u);
            while (__ballot(todo != 0)) {
                const int cc = todo ? __ffs((int)todo) - 1 : 1;
                const uint64_t v = r.hk ? fmd_block_rank1(r.bk, r.t, r.nk, cc, r.blk_k) : 0;
                if (todo) { tk[1] = cc == 1 ? v : tk[1]; tk[2] = cc == 2 ? v : tk[2]; tk[3] = cc == 3 ? v : tk[3]; tk[4] = cc == 4 ? v : tk[4]; }
                todo &= todo - 1;
            }
        }
    } else if (active) {
        if (r.hl) fmd_block_rank6<false>(r.bl, r.tl, r.nl, tl, r.blk_l);
#pragma unroll
        for (int c = 0; c < 6; ++c) s[c] = tl[c] - tk[c];
    }
}

// Output space of a level is handed out in chunks of `ch` entries per wave: ONE device-wide atomic
// per chunk instead of one per 64 nodes (a single counter serialises at ~10^8 atomics/s, which
// capped the first version of this kernel at a twelfth of the gather rate).  A wave zero-fills
// what it leaves unused of its last chunk (size 0 = hole; the next level skips holes), so a
// frontier is "children + at most one partial chunk per wave".
struct KmChunk { unsigned long long base; uint32_t fill, ch; bool have; };

__device__ __forceinline__ void km_zero_fill(fmd_intv_t *out, uint64_t cap, const KmChunk &k)
{
    if (!k.have) return;
    for (uint64_t e = k.base + k.fill + fmd_lane(); e < k.base + k.ch; e += 64)
        if (e < cap) { uint4 *q = (uint4 *)(out + e); q[0] = make_uint4(0, 0, 0, 0); q[1] = make_uint4(0, 0, 0, 0); }
}

// first output slot for `tot` new entries of this wave (wave-uniform)
__device__ __forceinline__ unsigned long long km_reserve(fmd_intv_t *out, uint64_t cap, KmChunk &k, uint32_t tot, unsigned long long *ctr_next)
{
    if (!k.have || k.fill + tot > k.ch) {
        km_zero_fill(out, cap, k);
        unsigned long long first = 0;
        if (fmd_lane() == 0) first = atomicAdd(ctr_next, (unsigned long long)k.ch);
        k.base = ((unsigned long long)__builtin_amdgcn_readfirstlane((int)(first >> 32)) << 32) |
                 (unsigned int)__builtin_amdgcn_readfirstlane((int)first);
        k.fill = 0; k.have = true;
    }
    const unsigned long long o = k.base + k.fill;
    k.fill += tot;
    return o;
}

// Which frontier slots a workgroup walks.  Consecutive workgroups land on different XCDs (8 of them, each
// with its own L2), while consecutive frontier entries are children of neighbouring nodes and touch
// neighbouring rank blocks: give every XCD one contiguous eighth of the frontier so that this locality
// stays inside one L2.
struct KmRange { uint64_t beg, end, stride; };
__device__ __forceinline__ KmRange km_range(uint64_t n, int xcd_aware)
{
    KmRange r;
    if (xcd_aware && (gridDim.x & 7) == 0) {
        const uint32_t x = blockIdx.x & 7, slot = blockIdx.x >> 3, per = gridDim.x >> 3;
        const uint64_t seg = ((n + 7) / 8 + 63) & ~63ull;
        r.beg = (uint64_t)x * seg + (uint64_t)slot * 64;
        r.end = (uint64_t)(x + 1) * seg < n ? (uint64_t)(x + 1) * seg : n;
        r.stride = (uint64_t)per * 64;
    } else { r.beg = (uint64_t)blockIdx.x * 64; r.end = n; r.stride = (uint64_t)gridDim.x * 64; }
    return r;
}

// one trie level: nodes at depth d -> children at depth d+1
__global__ __launch_bounds__(64) void k_kmer_level(FmdIndexView ix, int d, int suf_len, int min_occ, const fmd_intv_t *__restrict__ in,
                                                   fmd_intv_t *__restrict__ out, uint64_t cap, uint32_t ch, unsigned long long *__restrict__ ctr,
                                                   int xcd_aware)
{
    FMD_DECLARE_COMPACT_LDS();
    const int lane = fmd_lane();
    const uint64_t n = ctr[d] < cap ? ctr[d] : cap;   // an overflowing level counted more than it stored
    const uint64_t thr = (d + 1 <= suf_len) ? 1 : (uint64_t)min_occ; // exact.c:159 vs correct.c:78
    const KmRange rg = km_range(n, xcd_aware);
    const uint64_t stride = rg.stride;
    KmChunk ck; ck.base = 0; ck.fill = 0; ck.ch = ch; ck.have = false;
    uint32_t n_ext = 0;
    uint4 na = make_uint4(0, 0, 0, 0), nb = na;       // the entry of the next iteration, loaded one gather ahead
    {
        const uint64_t i0 = rg.beg + lane;
        if (i0 < rg.end) { const uint4 *q = (const uint4 *)(in + i0); na = q[0]; nb = q[1]; }
    }
    for (uint64_t base = rg.beg; base < rg.end; base += stride) {
        const uint4 a = na, b = nb;
        const uint64_t x0 = (uint64_t)a.y << 32 | a.x, x1 = (uint64_t)a.w << 32 | a.z;
        const uint64_t sz = (uint64_t)b.y << 32 | b.x, K = (uint64_t)b.w << 32 | b.z;
        const bool act = base + lane < rg.end && sz != 0;
        {
            const uint64_t i1 = base + stride + lane;
            na = make_uint4(0, 0, 0, 0); nb = na;
            if (i1 < rg.end) { const uint4 *q = (const uint4 *)(in + i1); na = q[0]; nb = q[1]; }
        }
        const uint64_t m_act = __ballot(act);
        if (m_act == 0) continue;                     // a run of holes
        n_ext += (uint32_t)__popcll(m_act);
        uint64_t tk[6], s[6];
        km_extend_back<true>(ix, fmd_lds, act, x0, sz, thr, tk, s);
        // children c = 1..4 (ambiguous bases are skipped, correct.c:77); x[1] = running sum in
        // the order $,T,G,C,A (exact.c:81-86)
        const bool has1 = act && s[1] >= thr, has2 = act && s[2] >= thr, has3 = act && s[3] >= thr, has4 = act && s[4] >= thr;
        const uint64_t m1 = __ballot(has1), m2 = __ballot(has2), m3 = __ballot(has3), m4 = __ballot(has4);
        const uint32_t tot = (uint32_t)(__popcll(m1) + __popcll(m2) + __popcll(m3) + __popcll(m4));
        if (tot == 0) continue;
        const unsigned long long first = km_reserve(out, cap, ck, tot, &ctr[d + 1]);
        uint64_t o1 = first + fmd_below(m1);
        uint64_t o2 = first + __popcll(m1) + fmd_below(m2);
        uint64_t o3 = first + __popcll(m1) + __popcll(m2) + fmd_below(m3);
        uint64_t o4 = first + __popcll(m1) + __popcll(m2) + __popcll(m3) + fmd_below(m4);
        const uint64_t x1_4 = x1 + s[0], x1_3 = x1_4 + s[4], x1_2 = x1_3 + s[3], x1_1 = x1_2 + s[2];
#define KM_PUSH(c, has, o, x1c)                                                                   \
        if (has) {                                                                                \
            if (o < cap) {                                                                        \
                const uint64_t nx0 = ix.cnt[c] + tk[c], nk = K | (uint64_t)(c - 1) << (2 * d);    \
                uint4 *q = (uint4 *)(out + o);                                                    \
                q[0] = make_uint4((uint32_t)nx0, (uint32_t)(nx0 >> 32), (uint32_t)(x1c), (uint32_t)((x1c) >> 32)); \
                q[1] = make_uint4((uint32_t)s[c], (uint32_t)(s[c] >> 32), (uint32_t)nk, (uint32_t)(nk >> 32));     \
            } else ctr[KM_OVF] = 1;                                                               \
        }
        KM_PUSH(1, has1, o1, x1_1) KM_PUSH(2, has2, o2, x1_2) KM_PUSH(3, has3, o3, x1_3) KM_PUSH(4, has4, o4, x1_4)
#undef KM_PUSH
    }
    km_zero_fill(out, cap, ck);
    if (lane == 0 && n_ext) atomicAdd(&ctr[KM_EXT], (unsigned long long)n_ext);
}

// nodes at depth w: pick the most frequent next base and emit (bucket, key, val), correct.c:56-75.
// No atomics in the loop: node i writes slot i of the raw arrays (flag 0 = nothing kept); the host
// entry compacts them with hipCUB.  cnt[0]/cnt[1] are summed per wave and added once.
__global__ __launch_bounds__(64) void k_kmer_emit(FmdIndexView ix, int w, int suf_len, int min_occ, const fmd_intv_t *__restrict__ in,
                                                  uint32_t *__restrict__ r_bucket, uint32_t *__restrict__ r_key,
                                                  uint8_t *__restrict__ r_val, uint8_t *__restrict__ r_flag, uint64_t cap,
                                                  unsigned long long *__restrict__ ctr, int xcd_aware)
{
    FMD_DECLARE_COMPACT_LDS();
    const int lane = fmd_lane();
    const uint64_t n = ctr[w] < cap ? ctr[w] : cap;
    const KmRange rg = km_range(n, xcd_aware);
    const uint64_t stride = rg.stride;
    uint32_t n_keep = 0, n_inf = 0, n_ext = 0;
    uint4 na = make_uint4(0, 0, 0, 0), nb = na;
    {
        const uint64_t i0 = rg.beg + lane;
        if (i0 < rg.end) { const uint4 *q = (const uint4 *)(in + i0); na = q[0]; nb = q[1]; }
    }
    for (uint64_t base = rg.beg; base < rg.end; base += stride) {
        const uint64_t i = base + lane;
        const uint4 a = na, b = nb;
        const uint64_t x0 = (uint64_t)a.y << 32 | a.x;
        const uint64_t sz = (uint64_t)b.y << 32 | b.x, K = (uint64_t)b.w << 32 | b.z;
        const bool act = i < rg.end && sz != 0;
        {
            const uint64_t i1 = i + stride;
            na = make_uint4(0, 0, 0, 0); nb = na;
            if (i1 < rg.end) { const uint4 *q = (const uint4 *)(in + i1); na = q[0]; nb = q[1]; }
        }
        const uint64_t m_act = __ballot(act);
        if (m_act == 0) continue;
        n_ext += (uint32_t)__popcll(m_act);
        uint64_t tk[6], s[6];
        km_extend_back<false>(ix, fmd_lds, act, x0, sz, 0, tk, s);
        uint64_t mx = 0; int max_c = 6;
#pragma unroll
        for (int c = 1; c <= 4; ++c) if (s[c] > mx) { mx = s[c]; max_c = c; }
        const bool keep = act && mx >= (uint64_t)min_occ;
        const uint64_t rest = sz - mx - s[0] - s[5];
        double r = rest == 0 ? (double)mx : (double)mx / (double)rest;   // IEEE double divide, as on the host
        if (r > 31.) r = 31.;
        const bool informative = keep && rest <= 7 && r >= (double)min_occ;
        n_keep += (uint32_t)__popcll(__ballot(keep)); n_inf += (uint32_t)__popcll(__ballot(informative));
        if (i < rg.end) {
            r_flag[i] = keep ? 1 : 0;
            if (keep) {
                r_bucket[i] = (uint32_t)(K & ((1ull << (2 * suf_len)) - 1));
                r_key[i] = (uint32_t)(K >> (2 * suf_len)) << 2 | (uint32_t)(max_c - 1);
                r_val[i] = (uint8_t)((int)(r + .499) << 3 | (int)(rest < 7 ? rest : 7));
            }
        }
    }
    if (lane == 0 && n_ext) atomicAdd(&ctr[KM_EXT], (unsigned long long)n_ext);
    if (lane == 0 && n_keep) {
        atomicAdd(&ctr[KM_CNT0], (unsigned long long)n_keep);
        if (n_inf) atomicAdd(&ctr[KM_CNT1], (unsigned long long)n_inf);
    }
}

// ---- compaction of the raw emit arrays: tile counts -> exclusive scan -> scatter ----------------
#define KM_TILE 2048   // items per 256-thread block, 8 per thread

__device__ __forceinline__ uint32_t km_tile_prefix(uint32_t mine, uint32_t *lds, uint32_t &total) // exclusive scan over 256 threads
{
    const int tid = (int)threadIdx.x, lane = tid & 63, wv = tid >> 6;
    uint32_t v = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint32_t u = (uint32_t)__shfl_up((int)v, o); if (lane >= o) v += u; }
    if (lane == 63) lds[wv] = v;
    __syncthreads();
    uint32_t before = 0;
    for (int i = 0; i < wv; ++i) before += lds[i];
    total = lds[0] + lds[1] + lds[2] + lds[3];
    return before + v - mine;
}

__global__ __launch_bounds__(256) void k_km_count(const uint8_t *__restrict__ flag, uint64_t cap, unsigned long long *__restrict__ tile_cnt)
{
    __shared__ uint32_t lds[4];
    const uint64_t i = (uint64_t)blockIdx.x * KM_TILE + threadIdx.x * 8;
    uint32_t c = 0;
    if (i + 8 <= cap) c = (uint32_t)__popcll(*(const uint64_t *)(flag + i) & 0x0101010101010101ull);
    else for (uint64_t j = i; j < cap; ++j) c += flag[j] & 1;
    uint32_t total;
    km_tile_prefix(c, lds, total);
    if (threadIdx.x == 0) tile_cnt[blockIdx.x] = total;
}

__global__ __launch_bounds__(256) void k_km_scatter(const uint8_t *__restrict__ flag, uint64_t cap, const unsigned long long *__restrict__ tile_off,
                                                    const unsigned long long *__restrict__ tile_cnt, uint64_t n_tiles,
                                                    const uint32_t *__restrict__ r_bucket, const uint32_t *__restrict__ r_key, const uint8_t *__restrict__ r_val,
                                                    uint32_t *__restrict__ o_bucket, uint32_t *__restrict__ o_key, uint8_t *__restrict__ o_val,
                                                    unsigned long long *__restrict__ ctr)
{
    __shared__ uint32_t lds[4];
    const uint64_t i = (uint64_t)blockIdx.x * KM_TILE + threadIdx.x * 8;
    uint64_t f = 0;
    if (i + 8 <= cap) f = *(const uint64_t *)(flag + i) & 0x0101010101010101ull;
    else for (uint64_t j = i; j < cap; ++j) f |= (uint64_t)(flag[j] & 1) << (8 * (j - i));
    uint32_t total;
    uint64_t o = tile_off[blockIdx.x] + km_tile_prefix((uint32_t)__popcll(f), lds, total);
    for (int j = 0; j < 8; ++j)
        if ((f >> (8 * j)) & 1) { o_bucket[o] = r_bucket[i + j]; o_key[o] = r_key[i + j]; o_val[o] = r_val[i + j]; ++o; }
    if (blockIdx.x == n_tiles - 1 && threadIdx.x == 0) ctr[KM_OUT] = tile_off[blockIdx.x] + tile_cnt[blockIdx.x];
}

extern "C" size_t fmd_kmer_work_bytes(uint64_t cap_frontier)
{
    return 2 * cap_frontier * sizeof(fmd_intv_t) + KM_WORDS * 8 + 512 + (4u << 20);
}

// d_status (device, 4 x u64): [0] number of (bucket,key,val) triples, [1] overflow flag (then the
// result is incomplete: call again with a larger cap), [2] cnt[0], [3] cnt[1] of correct.c:64-69.
// The same over the part of the trie whose k-mers END in one of the bases of seed_mask (bit c-1 = base c): the trie is a forest
// rooted at the last base, so the four parts are disjoint, their union is the whole harvest and each needs about a quarter
// of the frontier -- what carries indexes beyond 2.5*10^10 symbols, whose full frontiers (2 x 32 bytes per distinct k-mer)
// do not fit next to the index.
extern "C" int fmd_kmer_collect_part_dev(fmd_dev_t *h, void *stream_, int w, int min_occ, int suf_len, int seed_mask, void *d_work, size_t work_bytes,
                                         uint64_t cap, uint32_t *d_bucket, uint32_t *d_key, uint8_t *d_val, uint64_t *d_status);
extern "C" int fmd_kmer_collect_dev(fmd_dev_t *h, void *stream_, int w, int min_occ, int suf_len, void *d_work, size_t work_bytes,
                                    uint64_t cap, uint32_t *d_bucket, uint32_t *d_key, uint8_t *d_val, uint64_t *d_status)
{
    return fmd_kmer_collect_part_dev(h, stream_, w, min_occ, suf_len, 0xf, d_work, work_bytes, cap, d_bucket, d_key, d_val, d_status);
}
extern "C" int fmd_kmer_collect_part_dev(fmd_dev_t *h, void *stream_, int w, int min_occ, int suf_len, int seed_mask, void *d_work, size_t work_bytes,
                                         uint64_t cap, uint32_t *d_bucket, uint32_t *d_key, uint8_t *d_val, uint64_t *d_status)
{
    if (!h || !d_work || !d_bucket || !d_key || !d_val || !d_status) return FMD_E_ARG;
    if (w < 2 || w > 27 || suf_len < 1 || suf_len >= w || min_occ < 1 || w - suf_len > 15 || cap < (1u << 16) || !(seed_mask & 0xf)) return FMD_E_ARG; // MAX_KMER 27 (correct.c:303)
    if (work_bytes < fmd_kmer_work_bytes(cap)) return FMD_E_ARG;
    FMD_HIP_TRY(hipSetDevice(h->device));
    hipStream_t st = (hipStream_t)stream_;
    unsigned long long *ctr = (unsigned long long *)d_work;
    fmd_intv_t *fa = (fmd_intv_t *)(((uintptr_t)((uint8_t *)d_work + KM_WORDS * 8) + 255) & ~(uintptr_t)255);
    fmd_intv_t *fb = fa + cap;
    uint8_t *tail = (uint8_t *)(fb + cap);                       // 4 MiB: scan scratch
    FMD_HIP_TRY(hipMemsetAsync(ctr, 0, KM_WORDS * 8, st));
    // depth 1: the four single-base intervals (exact.c:153-155: fm6_set_intv for the root)
    fmd_intv_t seed[4]; unsigned long long n1 = 0;
    for (int c = 1; c <= 4; ++c) {
        const uint64_t sz = h->cnt[c + 1] - h->cnt[c];
        if (sz == 0 || !((seed_mask >> (c - 1)) & 1)) continue;
        seed[n1].x[0] = h->cnt[c]; seed[n1].x[1] = h->cnt[5 - c]; seed[n1].x[2] = sz; seed[n1].info = (uint64_t)(c - 1);
        ++n1;
    }
    FMD_HIP_TRY(hipMemcpyAsync(fa, seed, n1 * sizeof(fmd_intv_t), hipMemcpyHostToDevice, st));
    FMD_HIP_TRY(hipMemcpyAsync(ctr + 1, &n1, 8, hipMemcpyHostToDevice, st));
    FMD_HIP_TRY(hipStreamSynchronize(st)); // seed[] is a stack buffer
    const FmdIndexView ix = fmd_view(h);
    // both kernels walk the frontier by static ranges: the grid is the smaller of the two resident sets
    static int per_cu = 0;
    if (!per_cu) {
        const int a = fmd_resident_per_cu(k_kmer_level, FMD_COMPACT_LDS_U4 * 16, 16, "k_kmer_level");
        const int b = fmd_resident_per_cu(k_kmer_emit, FMD_COMPACT_LDS_U4 * 16, 16, "k_kmer_emit");
        per_cu = a < b ? a : b;
    }
    int grid = fmd_grid_for_lds(h, cap, FMD_COMPACT_LDS_U4 * 16);
    if (grid > h->n_cu * per_cu) grid = h->n_cu * per_cu;
    // chunk of output slots a wave reserves per atomic: as large as the capacity comfortably allows
    // (at most one partial chunk per wave is wasted per level), at least one iteration's worth (256)
    static const int xcd_aware = getenv("FMD_KMER_XCD") ? atoi(getenv("FMD_KMER_XCD")) : 1;
    uint32_t ch = 1024;
    while (ch > 256 && (uint64_t)grid * ch * 8 > cap) ch >>= 1;
    fmd_intv_t *in = fa, *out = fb;
    for (int d = 1; d < w; ++d) {
        k_kmer_level<<<grid, 64, 0, st>>>(ix, d, suf_len, min_occ, in, out, cap, ch, ctr, xcd_aware);
        fmd_intv_t *t = in; in = out; out = t;
    }
    // raw (uncompacted) triples + flags live in the frontier buffer that is free now: 10 of its 32 bytes per slot
    uint32_t *r_bucket = (uint32_t *)out, *r_key = r_bucket + cap;
    uint8_t *r_val = (uint8_t *)(r_key + cap), *r_flag = r_val + cap;
    const uint64_t n_tiles = (cap + KM_TILE - 1) / KM_TILE;
    unsigned long long *tile_cnt = (unsigned long long *)(((uintptr_t)(r_flag + cap) + 255) & ~(uintptr_t)255), *tile_off = tile_cnt + n_tiles;
    if ((uint8_t *)(tile_off + n_tiles) > (uint8_t *)out + cap * sizeof(fmd_intv_t)) return FMD_E_ARG;
    FMD_HIP_TRY(hipMemsetAsync(r_flag, 0, cap, st));
    k_kmer_emit<<<grid, 64, 0, st>>>(ix, w, suf_len, min_occ, in, r_bucket, r_key, r_val, r_flag, cap, ctr, xcd_aware);
    k_km_count<<<(unsigned)n_tiles, 256, 0, st>>>(r_flag, cap, tile_cnt);
    size_t tmp_bytes = 0;
    FMD_HIP_TRY(fmd_exclusive_sum(nullptr, tmp_bytes, tile_cnt, tile_off, (size_t)n_tiles, st));
    if (tmp_bytes > (4u << 20)) return FMD_E_ARG;
    FMD_HIP_TRY(fmd_exclusive_sum(tail, tmp_bytes, tile_cnt, tile_off, (size_t)n_tiles, st));
    k_km_scatter<<<(unsigned)n_tiles, 256, 0, st>>>(r_flag, cap, tile_off, tile_cnt, n_tiles, r_bucket, r_key, r_val, d_bucket, d_key, d_val, ctr);
    FMD_HIP_TRY(hipMemcpyAsync(d_status, ctr + KM_OUT, 4 * 8, hipMemcpyDeviceToDevice, st));
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { fmd_set_hip_error(e, "kmer kernels"); return FMD_E_HIP; }
    return FMD_OK;
}

// (bucket, key) packed into one sortable word and back: the host form returns the triples sorted, so
// the consumer's per-bucket tables need no sorting of their own
__global__ void k_km_pack(uint64_t n, const uint32_t *__restrict__ bucket, const uint32_t *__restrict__ key, uint64_t *__restrict__ k64)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) k64[i] = (uint64_t)bucket[i] << 32 | key[i];
}
__global__ void k_km_unpack(uint64_t n, const uint64_t *__restrict__ k64, uint32_t *__restrict__ bucket, uint32_t *__restrict__ key)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { bucket[i] = (uint32_t)(k64[i] >> 32); key[i] = (uint32_t)k64[i]; }
}

// sorts the m triples in (db, dk, dv) by (bucket, key) in place; a no-op for m beyond hipCUB's item count
static int km_sort_triples(uint64_t m, int suf_len, uint32_t *db, uint32_t *dk, uint8_t *dv)
{
    if (m < 2 || m > 0x7fffffffull) return FMD_OK;
    uint64_t *ka = nullptr, *kb = nullptr; uint8_t *vb = nullptr; void *tmp = nullptr;
    size_t tmp_bytes = 0;
    int rc = FMD_OK;
    if (hipMalloc((void **)&ka, m * 8) != hipSuccess || hipMalloc((void **)&kb, m * 8) != hipSuccess || hipMalloc((void **)&vb, m) != hipSuccess) rc = FMD_E_NOMEM;
    if (rc == FMD_OK) {
        const unsigned nb = (unsigned)((m + 255) / 256);
        k_km_pack<<<nb, 256>>>(m, db, dk, ka);
        const int end_bit = 32 + 2 * suf_len;
        if (fmd_sort_pairs(nullptr, tmp_bytes, ka, kb, dv, vb, (int)m, 0, end_bit) != hipSuccess ||
            hipMalloc(&tmp, tmp_bytes ? tmp_bytes : 16) != hipSuccess) rc = FMD_E_NOMEM;
        else if (fmd_sort_pairs(tmp, tmp_bytes, ka, kb, dv, vb, (int)m, 0, end_bit) != hipSuccess) rc = FMD_E_HIP;
        else {
            k_km_unpack<<<nb, 256>>>(m, kb, db, dk);
            if (hipMemcpy(dv, vb, m, hipMemcpyDeviceToDevice) != hipSuccess) rc = FMD_E_HIP;
        }
    }
    hipFree(ka); hipFree(kb); hipFree(vb); hipFree(tmp);
    return rc;
}

// One part of the harvest (seed_mask) on the device with the capacity grown until nothing overflows; triples sorted by
// (bucket, key) and appended to the host arrays.
static int km_collect_part_host(fmd_dev_t *h, int w, int min_occ, int suf_len, int seed_mask, uint64_t cap0, uint32_t **bucket, uint32_t **key,
                                uint8_t **val, uint64_t *n, uint64_t *m_alloc, int64_t cnt[2])
{
    uint64_t cap = cap0, demand = 0;
    for (int attempt = 0; attempt < 24; ++attempt) {
        if (attempt) {   // grow by what the overflowed pass asked for (the widest level it counted, + a fifth), at least twice
            const uint64_t by_demand = demand + demand / 5 + 1024;
            cap = by_demand > 2 * cap ? by_demand : 2 * cap;
        }
        void *work = nullptr, *db = nullptr, *dk = nullptr, *dv = nullptr, *ds = nullptr;
        const size_t wb = fmd_kmer_work_bytes(cap);
        int rc = FMD_OK;
        if (hipMalloc(&work, wb) != hipSuccess || hipMalloc(&db, cap * 4) != hipSuccess || hipMalloc(&dk, cap * 4) != hipSuccess ||
            hipMalloc(&dv, cap) != hipSuccess || hipMalloc(&ds, 32) != hipSuccess) { (void)hipGetLastError(); rc = FMD_E_NOMEM; }
        uint64_t status[4] = {0, 0, 0, 0};
        if (rc == FMD_OK) rc = fmd_kmer_collect_part_dev(h, nullptr, w, min_occ, suf_len, seed_mask, work, wb, cap, (uint32_t *)db, (uint32_t *)dk, (uint8_t *)dv, (uint64_t *)ds);
        if (rc == FMD_OK && hipMemcpy(status, ds, 32, hipMemcpyDeviceToHost) != hipSuccess) rc = FMD_E_HIP;
        if (rc == FMD_OK && status[1] != 0) {   // overflowed: the level counters (exact up to the first level that did not fit) say how much is needed
            unsigned long long lv[KM_WORDS];
            if (hipMemcpy(lv, work, sizeof(lv), hipMemcpyDeviceToHost) == hipSuccess) { demand = 0; for (int d = 1; d <= w && d < KM_WORDS; ++d) if (lv[d] > demand) demand = lv[d]; }
        }
        if (rc == FMD_OK && status[1] == 0) {
            hipFree(work); work = nullptr;   // the frontier buffers are not needed any more; the sort wants the room
            rc = km_sort_triples(status[0], suf_len, (uint32_t *)db, (uint32_t *)dk, (uint8_t *)dv);
        }
        if (rc == FMD_OK && status[1] == 0) {
            const uint64_t m = status[0];
            if (*n + m + 4 > *m_alloc) {
                const uint64_t want = (*n + m) + (*n ? (*n + m) / 2 : 0) + 4;     // later parts are about as large as this one
                uint32_t *nb = (uint32_t *)realloc(*bucket, want * 4), *nk = nb ? (uint32_t *)realloc(*key, want * 4) : nullptr;
                if (nb) *bucket = nb;
                if (nk) *key = nk;
                uint8_t *nv = nk ? (uint8_t *)realloc(*val, want) : nullptr;
                if (nv) *val = nv;
                if (!nb || !nk || !nv) rc = FMD_E_NOMEM; else *m_alloc = want;
            }
            if (rc == FMD_OK && m && (hipMemcpy(*bucket + *n, db, m * 4, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(*key + *n, dk, m * 4, hipMemcpyDeviceToHost) != hipSuccess ||
                                      hipMemcpy(*val + *n, dv, m, hipMemcpyDeviceToHost) != hipSuccess)) rc = FMD_E_HIP;
            if (rc == FMD_OK) { *n += m; cnt[0] += (int64_t)status[2]; cnt[1] += (int64_t)status[3]; }
        }
        hipFree(work); hipFree(db); hipFree(dk); hipFree(dv); hipFree(ds);
        if (rc != FMD_OK) return rc;   // FMD_E_NOMEM: the caller cuts the harvest into more parts
        if (status[1] == 0) return FMD_OK;
    }
    return FMD_E_OVERFLOW;
}

// Host form; outputs are malloc'ed (fmd_host_free).  One pass over the whole trie when its frontiers fit next to the index
// (triples sorted by (bucket, key)), else the four parts by last base one after the other (each part sorted, parts
// concatenated: the consumers -- fmd_ectab_build, multiset comparisons -- do not depend on the order).
extern "C" int fmd_kmer_collect(fmd_dev_t *h, int w, int min_occ, int suf_len, uint32_t **bucket, uint32_t **key, uint8_t **val,
                                uint64_t *n, int64_t cnt[2])
{
    if (!h || !bucket || !key || !val || !n || !cnt) return FMD_E_ARG;
    FMD_HIP_TRY(hipSetDevice(h->device));
    *bucket = nullptr; *key = nullptr; *val = nullptr; *n = 0; cnt[0] = cnt[1] = 0;
    // distinct k-mers of both strands ~ symbols / 30 at 30x; two frontiers of 32 bytes each + 9 bytes of output per slot
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) free_b = (size_t)64 << 30;
    const double est = (double)h->mcnt[0] / 20.0;
    int parts = est * 73.0 > 0.8 * (double)free_b ? 4 : 1;
    { const char *e = getenv("FMD_KMER_PARTS"); if (e && (atoi(e) == 1 || atoi(e) == 4)) parts = atoi(e); }
    uint64_t m_alloc = 0;
    int rc = FMD_OK;
    for (;;) {
        for (int p = 0; p < parts && rc == FMD_OK; ++p)
            rc = km_collect_part_host(h, w, min_occ, suf_len, parts == 1 ? 0xf : 1 << p, 1u << 22, bucket, key, val, n, &m_alloc, cnt);
        if (rc != FMD_E_NOMEM || parts != 1) break;
        // the frontiers of the whole trie did not fit beside the index after all: start over, a quarter at a time
        free(*bucket); free(*key); free(*val); *bucket = nullptr; *key = nullptr; *val = nullptr; *n = 0; m_alloc = 0; cnt[0] = cnt[1] = 0;
        parts = 4; rc = FMD_OK;
    }
    if (rc != FMD_OK) { free(*bucket); free(*key); free(*val); *bucket = nullptr; *key = nullptr; *val = nullptr; *n = 0; }
    else if (!*bucket) { *bucket = (uint32_t *)malloc(4); *key = (uint32_t *)malloc(4); *val = (uint8_t *)malloc(4); }
    return rc;
}

// The part of the harvest whose k-mers END in one of the bases of seed_mask (bit c-1 = base c), host form: what one GPU of several
// computes when `fermi-amd correct -g a,b,..` shards the harvest -- the reference hands suffix buckets to its workers
// (correct.c:346-356); here the trie is a forest rooted at the last base, so the shards are whole trees.  Same outputs as
// fmd_kmer_collect (malloc'ed, fmd_host_free; cnt[] = the shard's own counts); the union over disjoint masks that cover 0xf is the
// whole harvest.
extern "C" int fmd_kmer_collect_seeds(fmd_dev_t *h, int w, int min_occ, int suf_len, int seed_mask, uint32_t **bucket, uint32_t **key, uint8_t **val,
                                      uint64_t *n, int64_t cnt[2])
{
    if (!h || !bucket || !key || !val || !n || !cnt || (seed_mask & ~0xf)) return FMD_E_ARG;
    FMD_HIP_TRY(hipSetDevice(h->device));
    *bucket = nullptr; *key = nullptr; *val = nullptr; *n = 0; cnt[0] = cnt[1] = 0;
    uint64_t m_alloc = 0;
    int rc = FMD_OK;
    for (int p = 0; p < 4 && rc == FMD_OK; ++p)     // one tree at a time: a quarter of the frontier memory of the whole harvest
        if ((seed_mask >> p) & 1) rc = km_collect_part_host(h, w, min_occ, suf_len, 1 << p, 1u << 22, bucket, key, val, n, &m_alloc, cnt);
    if (rc != FMD_OK) { free(*bucket); free(*key); free(*val); *bucket = nullptr; *key = nullptr; *val = nullptr; *n = 0; }
    else if (!*bucket) { *bucket = (uint32_t *)malloc(4); *key = (uint32_t *)malloc(4); *val = (uint8_t *)malloc(4); }
    return rc;
}
