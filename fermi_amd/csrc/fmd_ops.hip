// fmd_ops.hip -- batched rank / extend / backward-search / retrieve kernels on the wave engine
// of fmd_wave.h, and their C-ABI entry points (include/fmd_hip.h).
//
// Every kernel runs 64-thread workgroups (one wavefront, 16 KiB LDS); lane = one query/search.
#include <stdlib.h>
#include <string.h>
#include "fmd_kernel_common.h"

#define NONE64 (~0ull)

// ------------------------------------------------------------------------------ rank kernels
// rld_rank1a (rld.c:424-446)
__global__ __launch_bounds__(64) void k_rank1a(FmdIndexView ix, size_t n, const uint64_t *__restrict__ d_k,
                                               uint64_t *__restrict__ d_ok, int8_t *__restrict__ d_sym)
{
    FMD_DECLARE_WAVE_LDS();
    const int lane = fmd_lane();
    const size_t stride = (size_t)gridDim.x * 64;
    for (size_t base = (size_t)blockIdx.x * 64; base < n; base += stride) {
        const size_t i = base + lane;
        const uint64_t k = i < n ? d_k[i] : NONE64;
        const FmdRank2 r = fmd_wave_rank2_fetch(ix, fmd_lds, k, NONE64);
        if (i < n) {
            uint64_t ok[6] = {0, 0, 0, 0, 0, 0};
            int sym = -1;
            if (r.hk) sym = fmd_block_rank6<true>(r.bk, r.t, r.nk, ok, r.blk_k);
#pragma unroll
            for (int c = 0; c < 6; ++c) d_ok[i * 6 + c] = ok[c];
            if (d_sym) d_sym[i] = (int8_t)sym;
        }
    }
}

// rld_rank2a (rld.c:457-492)
__global__ __launch_bounds__(64) void k_rank2a(FmdIndexView ix, size_t n, const uint64_t *__restrict__ d_k,
                                               const uint64_t *__restrict__ d_l, uint64_t *__restrict__ d_ok,
                                               uint64_t *__restrict__ d_ol)
{
    FMD_DECLARE_WAVE_LDS();
    const int lane = fmd_lane();
    const size_t stride = (size_t)gridDim.x * 64;
    for (size_t base = (size_t)blockIdx.x * 64; base < n; base += stride) {
        const size_t i = base + lane;
        const uint64_t k = i < n ? d_k[i] : NONE64, l = i < n ? d_l[i] : NONE64;
        const FmdRank2 r = fmd_wave_rank2_fetch(ix, fmd_lds, k, l);
        if (i < n) {
            uint64_t ok[6] = {0, 0, 0, 0, 0, 0}, ol[6] = {0, 0, 0, 0, 0, 0};
            if (r.hk) fmd_block_rank6<false>(r.bk, r.t, r.nk, ok, r.blk_k);
            if (r.hl) fmd_block_rank6<false>(r.bl, r.tl, r.nl, ol, r.blk_l);
#pragma unroll
            for (int c = 0; c < 6; ++c) { d_ok[i * 6 + c] = ok[c]; d_ol[i * 6 + c] = ol[c]; }
        }
    }
}

// fm6_extend (exact.c:72-88): one rank2a on strand o = !is_back, then the running sum over the
// other strand in the fixed order $,T,G,C,A,N.
__device__ __forceinline__ void fmd_extend_finish(const FmdIndexView &ix, const uint64_t x[3], int is_back,
                                                  const uint64_t tk[6], const uint64_t tl[6], fmd_intv_t ok[6])
{
    uint64_t a[6], b[6], s[6];   // a: coordinate on the searched strand, b: on the other strand
#pragma unroll
    for (int c = 0; c < 6; ++c) { a[c] = ix.cnt[c] + tk[c]; s[c] = tl[c] - tk[c]; }
    uint64_t acc = is_back ? x[1] : x[0];
    b[0] = acc; acc += s[0];
    b[4] = acc; acc += s[4];
    b[3] = acc; acc += s[3];
    b[2] = acc; acc += s[2];
    b[1] = acc; acc += s[1];
    b[5] = acc;
#pragma unroll
    for (int c = 0; c < 6; ++c) {
        ok[c].x[0] = is_back ? a[c] : b[c];
        ok[c].x[1] = is_back ? b[c] : a[c];
        ok[c].x[2] = s[c];
        ok[c].info = 0;
    }
}

__global__ __launch_bounds__(64) void k_extend(FmdIndexView ix, size_t n, const fmd_intv_t *__restrict__ d_ik,
                                               const uint8_t *__restrict__ d_is_back, fmd_intv_t *__restrict__ d_ok)
{
    FMD_DECLARE_WAVE_LDS();
    const int lane = fmd_lane();
    const size_t stride = (size_t)gridDim.x * 64;
    for (size_t base = (size_t)blockIdx.x * 64; base < n; base += stride) {
        const size_t i = base + lane;
        uint64_t x[3] = {0, 0, 0};
        int is_back = 0;
        uint64_t k = NONE64, l = NONE64;
        if (i < n) {
            x[0] = d_ik[i].x[0]; x[1] = d_ik[i].x[1]; x[2] = d_ik[i].x[2];
            is_back = d_is_back[i] != 0;
            const uint64_t a = is_back ? x[0] : x[1];
            k = a - 1; l = a - 1 + x[2];
        }
        const FmdRank2 r = fmd_wave_rank2_fetch(ix, fmd_lds, k, l);
        if (i < n) {
            uint64_t tk[6] = {0, 0, 0, 0, 0, 0}, tl[6] = {0, 0, 0, 0, 0, 0};
            if (r.hk) fmd_block_rank6<false>(r.bk, r.t, r.nk, tk, r.blk_k);
            if (r.hl) fmd_block_rank6<false>(r.bl, r.tl, r.nl, tl, r.blk_l);
            fmd_intv_t ok[6];
            fmd_extend_finish(ix, x, is_back, tk, tl, ok);
            uint4 *dst = (uint4 *)(d_ok + i * 6);
            const uint4 *src = (const uint4 *)ok;
#pragma unroll
            for (int c = 0; c < 12; ++c) dst[c] = src[c];
        }
    }
}

// ------------------------------------------------------------------- fm_backward_search
// exact.c:7-23.  Persistent waves; a lane that finishes its read (hit or early miss) pulls the
// next read index from a global queue: ballot -> one atomicAdd per wave -> prefix popcount, so the
// wave keeps 64 live SA intervals.
// MODE 0: the whole search.  MODE 1 (an index with two-base blocks, fmd_pair.hip): a search is HANDED OVER to k_bsearch_pair as soon as its interval is
// narrower than 65 and an even number of bases is left -- d_cnt = BS_HANDED, d_beg = k, d_end = bases left << 32 | size -- and goes on there two bases per
// request.  MODE 2: only the reads k_bsearch_pair gave back (d_cnt = BS_AGAIN: a base that is not A/C/G/T among those left), the whole search.
#define BS_HANDED (~0ull)
#define BS_AGAIN (~1ull)
template <int MODE>
__global__ __launch_bounds__(64) void k_bsearch(FmdIndexView ix, size_t n, const uint8_t *__restrict__ seqs,
                                                const uint64_t *__restrict__ off, uint64_t *__restrict__ d_cnt,
                                                uint64_t *__restrict__ d_beg, uint64_t *__restrict__ d_end,
                                                uint32_t *__restrict__ queue, const uint32_t *__restrict__ again_n = nullptr)
{
    if (MODE == 2 && again_n && *again_n == 0) return;   // nobody was given back (reads without an N: every run on real reads): 2 of 13.5 ms per 10^7 reads went into looking
    FMD_DECLARE_COMPACT_LDS();

    size_t rid = (size_t)-1;      // read being searched by this lane
    uint64_t sbase = 0;           // off[rid]
    int pos = -1;                 // next base to prepend
    uint64_t k = 0, l = 0;
    // 16 bases of the read around pos: the four dwords of the 16-byte block of the read buffer that holds base pos, fetched together.  A lane
    // reads its read backwards, one base per step; fetched one dword every fourth step (round 1-3), each of a line's 16 dwords was a request of
    // its own, four steps after the last -- long enough for the random block traffic of the other waves to have pushed the line out of L2:
    // the kernel fetched 1.18 x the bytes it asked for, 1 KB per read of refetched read lines (PMC).  Only dwords at or below
    // pos are loaded (the bases above it are behind us), so nothing beyond what the contract makes readable is touched.
    uint4 cq = make_uint4(0, 0, 0, 0);
    bool live = false, exhausted = false;
#define BS_LOAD16(at_)                                                                                           \
    do {                                                                                                         \
        const uint64_t a_ = (at_), b_ = a_ & ~15ull, top_ = a_ & ~3ull;                                          \
        const uint32_t *w_ = (const uint32_t *)(seqs + b_);                                                      \
        cq.x = w_[0];                                                                                            \
        cq.y = b_ + 4 <= top_ ? w_[1] : 0u; cq.z = b_ + 8 <= top_ ? w_[2] : 0u; cq.w = b_ + 12 <= top_ ? w_[3] : 0u; \
    } while (0)

    FmdTickets tk_;
    fmd_tickets_init(tk_, queue, 64, n);   // guided chunks (fmd_wave.h)
    for (;;) {
        // ---- refill finished lanes from the queue
        {
            const size_t my = fmd_tickets_take(tk_, queue, !live && !exhausted, n);
            if (!live && !exhausted) {
                if (my < n && !(MODE == 2 && d_cnt[my] != BS_AGAIN)) {
                    rid = my; sbase = off[my];
                    const int len = (int)(off[my + 1] - sbase);
                    if (len <= 0) { d_cnt[my] = 0; d_beg[my] = 0; d_end[my] = 0; }
                    else {
                        // the last ptab_d bases in one table look-up (the top of the search tree, two lines
                        // per step while the interval is wider than a block), when they are all A/C/G/T
                        const int D = ix.ptab_d;
                        bool from_table = false;
                        if (ix.ptab && len >= D) {
                            const uint64_t beg = sbase + (uint64_t)(len - D), end = sbase + (uint64_t)len;
                            uint64_t idx = 0; bool acgt = true;
                            for (uint64_t a = beg & ~3ull; a < end; a += 4) {
                                const uint32_t w = *(const uint32_t *)(seqs + a);
#pragma unroll
                                for (int b = 0; b < 4; ++b) {
                                    const uint32_t c = (w >> (8 * b)) & 0xff;
                                    if (a + b >= beg && a + b < end) { acgt = acgt && c >= 1 && c <= 4; idx = idx << 2 | ((c - 1) & 3); }
                                }
                            }
                            if (acgt) {
                                const uint4 e = ix.ptab[idx];
                                fmd_count_lane(ix, 1, 1);
                                k = (uint64_t)e.y << 32 | e.x; l = (uint64_t)e.w << 32 | e.z;
                                pos = len - D - 1;
                                from_table = true;
                                if (k > l) { d_cnt[my] = 0; d_beg[my] = 0; d_end[my] = 0; } // already a miss
                                else live = true;
                            }
                        }
                        if (!from_table) {
                            const int c = seqs[sbase + len - 1];
                            k = ix.cnt[c]; l = ix.cnt[c + 1] - 1;
                            pos = len - 2;
                            live = true;
                        }
                        if (live && pos >= 0) BS_LOAD16(sbase + pos);
                    }
                } else if (my >= n) exhausted = true;
            }
        }
        if (__ballot(live) == 0) { if (__ballot(!exhausted) == 0) break; else continue; }   // (a wave whose lanes all drew reads that need nothing draws again)

        // ---- retire lanes that have consumed their whole read (len == 1 lands here directly)
        if (live && pos < 0) {
            const bool hit = k <= l;
            d_cnt[rid] = hit ? l - k + 1 : 0; d_beg[rid] = hit ? k : 0; d_end[rid] = hit ? l : 0;
            live = false;
        }
        // ---- MODE 1: narrow, and an even number of bases left: the search goes on in k_bsearch_pair
        if (MODE == 1 && live && pos >= 1 && (pos & 1) && l - k < 64) {
            d_cnt[rid] = BS_HANDED; d_beg[rid] = k; d_end[rid] = (uint64_t)(uint32_t)(pos + 1) << 32 | (l - k + 1);
            live = false;
        }
        // ---- one backward step for every live lane: rank21(k-1, l, c)
        int c = 0;
        uint64_t qk = NONE64, ql = NONE64;
        if (live) {
            const uint64_t a = sbase + pos;
            const uint32_t wq = (uint32_t)(a >> 2) & 3u, cw_ = wq == 0 ? cq.x : wq == 1 ? cq.y : wq == 2 ? cq.z : cq.w;
            c = (int)((cw_ >> (8 * (a & 3))) & 0xff);
            qk = k - 1; ql = l;
        }
        FmdRank2c r = fmd_wave_rank2_fetch_compact(ix, fmd_lds, qk, ql);
        const uint64_t ok = (live && r.hk) ? fmd_block_rank1(r.bk, r.t, r.nk, c, r.blk_k) : 0;
        fmd_wave_l_ready(ix, fmd_lds, r); // only while the intervals are wide (more than 32 lanes straddle)
        if (live) {
            const uint64_t ol = fmd_block_rank1(r.bl, r.tl, r.nl, c, r.blk_l);
            k = ix.cnt[c] + ok;
            l = ix.cnt[c] + ol - 1;
            --pos;
            if (k > l || pos < 0) {
                const bool hit = k <= l;
                d_cnt[rid] = hit ? l - k + 1 : 0; d_beg[rid] = hit ? k : 0; d_end[rid] = hit ? l : 0;
                live = false;
            } else if (((sbase + pos) & 15) == 15) {
                BS_LOAD16(sbase + pos);
            }
        }
    }
#undef BS_LOAD16
}

// fm_backward_search two bases per request (round 6): for a read handed over by k_bsearch<1>, the interval [k, k + size) and `left` bases (even) to go.
// A step takes the read's next two bases (c1 nearest the interval, then c2): the rows of the interval with BWT[p] = c1 and BWT[LF(p)] = c2 are, in order,
// the interval two bases on; its start is one pair count (fmd_wave.h: the block's 28 bits + ix.pair_tab, which holds K2[c1][c2] too), its size a
// popcount.  A pair block starts every 32 positions and describes 96: an interval of up to 64 positions lies inside the block of its first position.
// One 8 KiB landing slot, ~60 registers: twenty waves per CU.  A read with a base that is not A/C/G/T among those left is given back (BS_AGAIN).
__global__ __launch_bounds__(64, 5) void k_bsearch_pair(FmdIndexView ix, size_t n, const uint8_t *__restrict__ seqs, const uint64_t *__restrict__ off,
                                                      uint64_t *__restrict__ d_cnt, uint64_t *__restrict__ d_beg, uint64_t *__restrict__ d_end, uint32_t *__restrict__ queue, uint32_t *__restrict__ again_n)
{
    __shared__ uint4 pair_lds[FMD_PAIR_SLOT_U4];
    const int q_ = fmd_lane(), px = fmd_pair_xor(q_);
    const uint4 *img = pair_lds + fmd_pair_base(q_);
    const uint32_t *iw = (const uint32_t *)img;
    size_t rid = 0;
    uint64_t k = 0, sbase = 0, hk = 0, he = 0, hc = 0, ho = 0;
    uint32_t size = 0;
    int pos = -1, st = 0;                 // st: 0 idle, 1 the hand-over record on its way, 2 running
    uint4 cq = make_uint4(0, 0, 0, 0), cp = cq;      // the 16 bases around pos, and the 16 below them
    bool exhausted = false;
#define BSP_LOAD16(dst_, at_, top_at_)                                                                           \
    do {                                                                                                         \
        const uint64_t b_ = (at_) & ~15ull, top_ = (top_at_) & ~3ull;                                            \
        const uint32_t *w_ = (const uint32_t *)(seqs + b_);                                                      \
        dst_.x = w_[0];                                                                                          \
        dst_.y = b_ + 4 <= top_ ? w_[1] : 0u; dst_.z = b_ + 8 <= top_ ? w_[2] : 0u; dst_.w = b_ + 12 <= top_ ? w_[3] : 0u; \
    } while (0)
#define BSP_BASE(a_, win_) ({ const uint32_t wq_ = (uint32_t)((a_) >> 2) & 3u, cw_ = wq_ == 0 ? win_.x : wq_ == 1 ? win_.y : wq_ == 2 ? win_.z : win_.w; (int)((cw_ >> (8 * ((a_) & 3))) & 0xff); })
    FmdTickets tk_;
    fmd_tickets_init(tk_, queue, 64, n);
    for (;;) {
        const size_t my = fmd_tickets_take(tk_, queue, st == 0 && !exhausted, n);
        if (st == 0 && !exhausted) {
            if (my < n) { rid = my; hc = d_cnt[my]; hk = d_beg[my]; he = d_end[my]; ho = off[my]; st = 1; }
            else exhausted = true;
        }
        if (__ballot(st != 0) == 0) break;
        fmd_pair_fetch(ix, pair_lds, (uint32_t)(k >> 5), st == 2);
        fmd_fetch_wait();
        if (st == 1) {
            st = 0;
            if (hc == BS_HANDED) {
                k = hk; size = (uint32_t)he; pos = (int)(he >> 32) - 1; sbase = ho;
                const uint64_t a = sbase + (uint64_t)pos;
                BSP_LOAD16(cq, a, a);
                if ((a & ~15ull) > (sbase & ~15ull)) BSP_LOAD16(cp, (a & ~15ull) - 16, (a & ~15ull) - 1); // (never below the read's own first block)
                st = 2;
            }
            continue;
        }
        if (st != 2) continue;
        const uint64_t a1 = sbase + (uint64_t)pos, a2 = a1 - 1;
        const int c1 = BSP_BASE(a1, cq);
        const int c2 = (a2 & ~15ull) == (a1 & ~15ull) ? BSP_BASE(a2, cq) : BSP_BASE(a2, cp);
        if (c1 < 1 || c1 > 4 || c2 < 1 || c2 > 4) { d_cnt[rid] = BS_AGAIN; atomicAdd(again_n, 1u); st = 0; continue; }
        const uint32_t offp = (uint32_t)k & 31u;
        const uint4 A0 = img[0 ^ px], A1 = img[1 ^ px], A2 = img[2 ^ px], B0 = img[3 ^ px], B1 = img[4 ^ px], B2 = img[5 ^ px];
        const uint32_t e0x = (c1 & 1) ? 0u : ~0u, e0y = (c1 & 2) ? 0u : ~0u, e0z = (c1 & 4) ? 0u : ~0u;
        const uint32_t e1x = (c2 & 1) ? 0u : ~0u, e1y = (c2 & 2) ? 0u : ~0u, e1z = (c2 & 4) ? 0u : ~0u;
        const uint32_t pm0 = (A0.x ^ e0x) & (A0.y ^ e0y) & (A0.z ^ e0z) & (A0.w ^ e1x) & (B0.x ^ e1y) & (B0.y ^ e1z);
        const uint32_t pm1 = (A1.x ^ e0x) & (A1.y ^ e0y) & (A1.z ^ e0z) & (A1.w ^ e1x) & (B1.x ^ e1y) & (B1.y ^ e1z);
        const uint32_t pm2 = (A2.x ^ e0x) & (A2.y ^ e0y) & (A2.z ^ e0z) & (A2.w ^ e1x) & (B2.x ^ e1y) & (B2.y ^ e1z);
        const uint64_t Mp = win64(pm0, pm1, pm2, offp) & bits_below((int)size);
        const uint32_t nsz = (uint32_t)__popcll(Mp);
        if (nsz == 0) { d_cnt[rid] = 0; d_beg[rid] = 0; d_end[rid] = 0; st = 0; continue; }      // a miss (exact.c:17-18: the outputs of a miss are not defined; zeros, as k_bsearch)
        const int pr = 4 * (c1 - 1) + (c2 - 1), bp = 28 * pr, tw = bp >> 5, tw1 = tw < 13 ? tw + 1 : 13;
#define WP_CW(t) iw[(((t) < 6 ? 3 + ((t) >> 1) : 6 + (((t) - 6) >> 2)) ^ px) * 4 + ((t) < 6 ? 2 + ((t) & 1) : (((t) - 6) & 3))]
        const uint32_t cwl = WP_CW(tw), cwh = WP_CW(tw1);
#undef WP_CW
        const uint32_t rel = __builtin_amdgcn_alignbit(cwh, cwl, (uint32_t)bp & 31u) & 0x0fffffffu;
        k = ix.pair_tab[(k >> (5 + FMD_PAIR_SB_SHIFT)) * 16 + (uint64_t)pr] + rel + (uint32_t)__builtin_popcount(pm0 & fmd_mask32((int)offp));
        size = nsz;
        pos -= 2;
        if (pos < 0) { d_cnt[rid] = size; d_beg[rid] = k; d_end[rid] = k + size - 1; st = 0; continue; }
        if (((sbase + (uint64_t)pos) & ~15ull) != (a1 & ~15ull)) {   // into the block below: it is here already; the one below that is asked for now
            cq = cp;
            const uint64_t nb_ = (sbase + (uint64_t)pos) & ~15ull;
            if (nb_ > (sbase & ~15ull)) BSP_LOAD16(cp, nb_ - 16, nb_ - 1);
        }
    }
#undef BSP_LOAD16
#undef BSP_BASE
}

// ------------------------------------------------------------------------------ forward reach
// For every position p of a buffer of zero-terminated nt6 sequences: the length of the longest
// prefix of seqs[p..] (up to the terminator) that occurs in the index.  This is the value
// fm6_smem1_core returns (smem.c:46: the end of its forward sweep), so x -> x + reach[x] is the chain
// fm6_miter_next / fm6_smem walk (smem.c:96-102, :404-409) -- computed here for ALL positions at once
// so that the chain itself becomes a pointer chase and every call on it an independent work item.
// A forward extension of W by c is a backward extension of revcomp(W) by comp(c): plain backward
// search on the reverse strand's interval, one single-symbol rank pair per base.
__global__ __launch_bounds__(64) void k_reach(FmdIndexView ix, size_t n, const uint8_t *__restrict__ seqs, uint32_t *__restrict__ out_len,
                                              uint32_t *__restrict__ queue)
{
    FMD_DECLARE_COMPACT_LDS();
    size_t p = 0, i = 0;
    uint64_t k = 0, l = 0;
    uint32_t cw = 0; size_t cw_at = (size_t)-1;
    bool live = false, exhausted = false;
    FmdTickets tk_;
    fmd_tickets_init(tk_, queue);
    for (;;) {
        {
            const size_t my = fmd_tickets_take(tk_, queue, !live && !exhausted);
            if (!live && !exhausted) {
                if (my < n) {
                    p = my;
                    cw_at = p & ~(size_t)3; cw = *(const uint32_t *)(seqs + cw_at);
                    const int c = (int)((cw >> (8 * (p & 3))) & 0xff);
                    const int cc = (c >= 1 && c <= 4) ? 5 - c : c;
                    if (c == 0 || c > 5) out_len[p] = 0;
                    else {
                        // first ptab_d symbols in one look-up when that many A/C/G/T follow and all of them match
                        // (the sweep consumes comp(q[p]), comp(q[p+1]), ..: the table string read backwards)
                        const int D = ix.ptab_d;
                        bool from_table = false;
                        if (ix.ptab) {
                            uint64_t idx = 0; bool acgt = true;
                            for (size_t a = p & ~(size_t)3; a < p + (size_t)D && acgt; a += 4) {
                                const uint32_t w = *(const uint32_t *)(seqs + a);
#pragma unroll
                                for (int b = 0; b < 4; ++b) {
                                    const uint32_t x = (w >> (8 * b)) & 0xff;
                                    if (a + b >= p && a + b < p + (size_t)D) { acgt = acgt && x >= 1 && x <= 4; idx |= (uint64_t)((4 - x) & 3) << (2 * (a + b - p)); }
                                }
                            }
                            if (acgt) {
                                const uint4 e = ix.ptab[idx];
                                fmd_count_lane(ix, 1, 1);
                                const uint64_t tk = (uint64_t)e.y << 32 | e.x, tl = (uint64_t)e.w << 32 | e.z;
                                if (tk <= tl) { k = tk; l = tl; i = p + (size_t)D; live = true; from_table = true; }
                            }
                        }
                        if (!from_table) {
                            k = ix.cnt[cc]; l = ix.cnt[cc + 1] - 1;
                            if (k > l) out_len[p] = 0;        // a base the index does not contain
                            else { i = p + 1; live = true; }
                        }
                    }
                } else exhausted = true;
            }
        }
        if (__ballot(live) == 0) { if (__ballot(!exhausted) == 0) break; else continue; }
        int cc = 0;
        if (live) {
            if ((i & ~(size_t)3) != cw_at) { cw_at = i & ~(size_t)3; cw = *(const uint32_t *)(seqs + cw_at); }
            const int c = (int)((cw >> (8 * (i & 3))) & 0xff);
            cc = (c >= 1 && c <= 4) ? 5 - c : c;
            if (c == 0 || c > 5) { out_len[p] = (uint32_t)(i - p); live = false; } // terminator
        }
        FmdRank2c r = fmd_wave_rank2_fetch_compact(ix, fmd_lds, live ? k - 1 : NONE64, live ? l : NONE64);
        const uint64_t ok = (live && r.hk) ? fmd_block_rank1(r.bk, r.t, r.nk, cc, r.blk_k) : 0;
        fmd_wave_l_ready(ix, fmd_lds, r);
        if (live) {
            const uint64_t ol = fmd_block_rank1(r.bl, r.tl, r.nl, cc, r.blk_l);
            k = ix.cnt[cc] + ok; l = ix.cnt[cc] + ol - 1;
            if (k > l) { out_len[p] = (uint32_t)(i - p); live = false; }
            else ++i;
        }
    }
}

// ------------------------------------------------------------------------------ fm_retrieve
// exact.c:59-70: LF-walk from row x until '$'.  One dependent rank1a per base.
__global__ __launch_bounds__(64) void k_retrieve(FmdIndexView ix, size_t n, const uint64_t *__restrict__ d_x,
                                                 uint8_t *__restrict__ d_seqs, uint32_t stride,
                                                 uint32_t *__restrict__ d_len, uint64_t *__restrict__ d_rank,
                                                 uint32_t *__restrict__ queue)
{
    FMD_DECLARE_WAVE_LDS();

    size_t rid = 0;
    uint64_t k = 0;
    uint32_t len = 0;
    bool live = false, exhausted = false;
    FmdTickets tk_;
    fmd_tickets_init(tk_, queue);
    for (;;) {
        {
            const size_t my = fmd_tickets_take(tk_, queue, !live && !exhausted);
            if (!live && !exhausted) {
                if (my < n) { rid = my; k = d_x[my]; len = 0; live = true; }
                else exhausted = true;
            }
        }
        if (__ballot(live) == 0) break;
        const FmdRank2 r = fmd_wave_rank2_fetch(ix, fmd_lds, live ? k : NONE64, NONE64);
        if (live) {
            uint64_t ok[6];
            const int c = fmd_block_rank6<true>(r.bk, r.t, r.nk, ok, r.blk_k);
            k = ix.cnt[c] + ok[c] - 1;
            if (c == 0) { d_len[rid] = len; d_rank[rid] = k; live = false; }
            else {
                if (len < stride) d_seqs[rid * (size_t)stride + len] = (uint8_t)c;
                ++len;
            }
        }
    }
}

// ------------------------------------------------------------------------------- host entry
static inline hipStream_t S(void *s) { return (hipStream_t)s; }

#define FMD_CHECK_LAUNCH()                                              \
    do {                                                                \
        hipError_t e__ = hipGetLastError();                             \
        if (e__ != hipSuccess) { fmd_set_hip_error(e__, "kernel launch"); return FMD_E_HIP; } \
    } while (0)

extern "C" int fmd_rank1a_dev(fmd_dev_t *h, void *stream, size_t n, const uint64_t *d_k, uint64_t *d_ok, int8_t *d_sym)
{
    if (!h || (n && (!d_k || !d_ok))) return FMD_E_ARG;
    if (n == 0) return FMD_OK;
    FMD_HIP_TRY(hipSetDevice(h->device));
    k_rank1a<<<fmd_grid_for(h, n), 64, 0, S(stream)>>>(fmd_view(h), n, d_k, d_ok, d_sym);
    FMD_CHECK_LAUNCH();
    return FMD_OK;
}

extern "C" int fmd_rank2a_dev(fmd_dev_t *h, void *stream, size_t n, const uint64_t *d_k, const uint64_t *d_l,
                              uint64_t *d_ok, uint64_t *d_ol)
{
    if (!h || (n && (!d_k || !d_l || !d_ok || !d_ol))) return FMD_E_ARG;
    if (n == 0) return FMD_OK;
    FMD_HIP_TRY(hipSetDevice(h->device));
    k_rank2a<<<fmd_grid_for(h, n), 64, 0, S(stream)>>>(fmd_view(h), n, d_k, d_l, d_ok, d_ol);
    FMD_CHECK_LAUNCH();
    return FMD_OK;
}

extern "C" int fmd_extend_dev(fmd_dev_t *h, void *stream, size_t n, const fmd_intv_t *d_ik, const uint8_t *d_is_back,
                              fmd_intv_t *d_ok)
{
    if (!h || (n && (!d_ik || !d_is_back || !d_ok))) return FMD_E_ARG;
    if (n == 0) return FMD_OK;
    FMD_HIP_TRY(hipSetDevice(h->device));
    k_extend<<<fmd_grid_for(h, n), 64, 0, S(stream)>>>(fmd_view(h), n, d_ik, d_is_back, d_ok);
    FMD_CHECK_LAUNCH();
    return FMD_OK;
}

extern "C" int fmd_bsearch_dev(fmd_dev_t *h, void *stream, size_t n, const uint8_t *d_seqs, const uint64_t *d_off,
                               uint64_t *d_cnt, uint64_t *d_beg, uint64_t *d_end)
{
    if (!h || (n && (!d_seqs || !d_off || !d_cnt || !d_beg || !d_end))) return FMD_E_ARG;
    if (n == 0) return FMD_OK;
    if (n >= 0xffffff00ull) return FMD_E_ARG; // 32-bit queue head
    FMD_HIP_TRY(hipSetDevice(h->device));
    uint32_t *q = fmd_next_queue(h, S(stream));
    (void)fmd_pairs_ensure(h, 0);          // (built here only where FMD_PAIR asks for it: fmd_pair.hip)
    const FmdIndexView ix = fmd_view(h);
    bool pairs = ix.pair != nullptr && ix.pair_tab != nullptr;
    { const char *e = getenv("FMD_PAIR_USE"); if (e && atoi(e) == 0) pairs = false; }   // A/B switch on a handle that has the two-base blocks
    const int grid = fmd_grid_for_lds(h, n, FMD_COMPACT_LDS_U4 * 16);
    if (pairs) {
        // the search one base at a time until the interval is narrow, then two bases per request (k_bsearch_pair), then -- one base at a time, whole -- the reads
        // that kernel gave back (an N among the bases left).  All three on the stream, no host in between: the third finds nothing to do on real reads.
        k_bsearch<1><<<grid, 64, 0, S(stream)>>>(ix, n, d_seqs, d_off, d_cnt, d_beg, d_end, q);
        uint32_t *q2 = fmd_next_queue(h, S(stream));
        int grid2 = h->n_cu * 20;
        if ((size_t)grid2 > (n + 63) / 64) grid2 = (int)((n + 63) / 64);
        uint32_t *again_n = fmd_next_queue(h, S(stream));      // (a zeroed word of the handle's: the reads given back, counted)
        k_bsearch_pair<<<grid2, 64, 0, S(stream)>>>(ix, n, d_seqs, d_off, d_cnt, d_beg, d_end, q2, again_n);
        uint32_t *q3 = fmd_next_queue(h, S(stream));
        k_bsearch<2><<<grid, 64, 0, S(stream)>>>(ix, n, d_seqs, d_off, d_cnt, d_beg, d_end, q3, again_n);
    } else
    k_bsearch<0><<<grid, 64, 0, S(stream)>>>(ix, n, d_seqs, d_off, d_cnt, d_beg, d_end, q);
    FMD_CHECK_LAUNCH();
    return FMD_OK;
}

extern "C" int fmd_reach_dev(fmd_dev_t *h, void *stream, size_t n_bytes, const uint8_t *d_seqs, uint32_t *d_len)
{
    if (!h || (n_bytes && (!d_seqs || !d_len)) || ((uintptr_t)d_seqs & 3)) return FMD_E_ARG;
    if (n_bytes == 0) return FMD_OK;
    if (n_bytes >= 0xffffff00ull) return FMD_E_ARG; // 32-bit queue head
    FMD_HIP_TRY(hipSetDevice(h->device));
    uint32_t *q = fmd_next_queue(h, S(stream));
    k_reach<<<fmd_grid_for_lds(h, n_bytes, FMD_COMPACT_LDS_U4 * 16), 64, 0, S(stream)>>>(fmd_view(h), n_bytes, d_seqs, d_len, q);
    FMD_CHECK_LAUNCH();
    return FMD_OK;
}

extern "C" int fmd_retrieve_dev(fmd_dev_t *h, void *stream, size_t n, const uint64_t *d_x, uint8_t *d_seqs, uint32_t stride,
                                uint32_t *d_len, uint64_t *d_rank)
{
    if (!h || (n && (!d_x || !d_seqs || !d_len || !d_rank || stride == 0))) return FMD_E_ARG;
    if (n == 0) return FMD_OK;
    if (n >= 0xffffff00ull) return FMD_E_ARG;
    FMD_HIP_TRY(hipSetDevice(h->device));
    uint32_t *q = fmd_next_queue(h, S(stream));
    k_retrieve<<<fmd_grid_for(h, n), 64, 0, S(stream)>>>(fmd_view(h), n, d_x, d_seqs, stride, d_len, d_rank, q);
    FMD_CHECK_LAUNCH();
    return FMD_OK;
}

// ---- host-pointer convenience forms: copy in, run, copy out, synchronise --------------------
struct DevBuf {
    void *p = nullptr;
    int alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 16) == hipSuccess ? FMD_OK : FMD_E_NOMEM; }
    ~DevBuf() { if (p) hipFree(p); }
};
#define TRY_RC(x) do { int rc__ = (x); if (rc__) return rc__; } while (0)

extern "C" int fmd_rank1a_batch(fmd_dev_t *h, size_t n, const uint64_t *k, uint64_t *ok, int8_t *sym)
{
    if (!h || (n && (!k || !ok))) return FMD_E_ARG;
    if (n == 0) return FMD_OK;
    FMD_HIP_TRY(hipSetDevice(h->device));
    DevBuf dk, dok, ds;
    TRY_RC(dk.alloc(n * 8)); TRY_RC(dok.alloc(n * 48)); TRY_RC(ds.alloc(n));
    FMD_HIP_TRY(hipMemcpy(dk.p, k, n * 8, hipMemcpyHostToDevice));
    TRY_RC(fmd_rank1a_dev(h, nullptr, n, (uint64_t *)dk.p, (uint64_t *)dok.p, (int8_t *)ds.p));
    FMD_HIP_TRY(hipMemcpy(ok, dok.p, n * 48, hipMemcpyDeviceToHost));
    if (sym) FMD_HIP_TRY(hipMemcpy(sym, ds.p, n, hipMemcpyDeviceToHost));
    return FMD_OK;
}

extern "C" int fmd_rank2a_batch(fmd_dev_t *h, size_t n, const uint64_t *k, const uint64_t *l, uint64_t *ok, uint64_t *ol)
{
    if (!h || (n && (!k || !l || !ok || !ol))) return FMD_E_ARG;
    if (n == 0) return FMD_OK;
    FMD_HIP_TRY(hipSetDevice(h->device));
    DevBuf dk, dl, dok, dol;
    TRY_RC(dk.alloc(n * 8)); TRY_RC(dl.alloc(n * 8)); TRY_RC(dok.alloc(n * 48)); TRY_RC(dol.alloc(n * 48));
    FMD_HIP_TRY(hipMemcpy(dk.p, k, n * 8, hipMemcpyHostToDevice));
    FMD_HIP_TRY(hipMemcpy(dl.p, l, n * 8, hipMemcpyHostToDevice));
    TRY_RC(fmd_rank2a_dev(h, nullptr, n, (uint64_t *)dk.p, (uint64_t *)dl.p, (uint64_t *)dok.p, (uint64_t *)dol.p));
    FMD_HIP_TRY(hipMemcpy(ok, dok.p, n * 48, hipMemcpyDeviceToHost));
    FMD_HIP_TRY(hipMemcpy(ol, dol.p, n * 48, hipMemcpyDeviceToHost));
    return FMD_OK;
}

extern "C" int fmd_extend_batch(fmd_dev_t *h, size_t n, const fmd_intv_t *ik, const uint8_t *is_back, fmd_intv_t *ok)
{
    if (!h || (n && (!ik || !is_back || !ok))) return FMD_E_ARG;
    if (n == 0) return FMD_OK;
    FMD_HIP_TRY(hipSetDevice(h->device));
    DevBuf di, db, dok;
    TRY_RC(di.alloc(n * 32)); TRY_RC(db.alloc(n)); TRY_RC(dok.alloc(n * 192));
    FMD_HIP_TRY(hipMemcpy(di.p, ik, n * 32, hipMemcpyHostToDevice));
    FMD_HIP_TRY(hipMemcpy(db.p, is_back, n, hipMemcpyHostToDevice));
    TRY_RC(fmd_extend_dev(h, nullptr, n, (fmd_intv_t *)di.p, (uint8_t *)db.p, (fmd_intv_t *)dok.p));
    FMD_HIP_TRY(hipMemcpy(ok, dok.p, n * 192, hipMemcpyDeviceToHost));
    return FMD_OK;
}

extern "C" int fmd_bsearch_batch(fmd_dev_t *h, size_t n, const uint8_t *seqs, const uint64_t *off,
                                 uint64_t *cnt, uint64_t *beg, uint64_t *end)
{
    if (!h || (n && (!seqs || !off || !cnt || !beg || !end))) return FMD_E_ARG;
    if (n == 0) return FMD_OK;
    FMD_HIP_TRY(hipSetDevice(h->device));
    const size_t total = off[n];
    DevBuf ds, doff, dc, dbg, den;
    TRY_RC(ds.alloc(total + 8)); TRY_RC(doff.alloc((n + 1) * 8)); TRY_RC(dc.alloc(n * 8)); TRY_RC(dbg.alloc(n * 8)); TRY_RC(den.alloc(n * 8));
    FMD_HIP_TRY(hipMemcpy(ds.p, seqs, total, hipMemcpyHostToDevice));
    FMD_HIP_TRY(hipMemcpy(doff.p, off, (n + 1) * 8, hipMemcpyHostToDevice));
    TRY_RC(fmd_bsearch_dev(h, nullptr, n, (uint8_t *)ds.p, (uint64_t *)doff.p, (uint64_t *)dc.p, (uint64_t *)dbg.p, (uint64_t *)den.p));
    FMD_HIP_TRY(hipMemcpy(cnt, dc.p, n * 8, hipMemcpyDeviceToHost));
    FMD_HIP_TRY(hipMemcpy(beg, dbg.p, n * 8, hipMemcpyDeviceToHost));
    FMD_HIP_TRY(hipMemcpy(end, den.p, n * 8, hipMemcpyDeviceToHost));
    return FMD_OK;
}

extern "C" int fmd_reach_batch(fmd_dev_t *h, size_t n_bytes, const uint8_t *seqs, uint32_t *len)
{
    if (!h || (n_bytes && (!seqs || !len))) return FMD_E_ARG;
    if (n_bytes == 0) return FMD_OK;
    FMD_HIP_TRY(hipSetDevice(h->device));
    DevBuf ds, dl;
    TRY_RC(ds.alloc(n_bytes + 8)); TRY_RC(dl.alloc(n_bytes * 4));
    FMD_HIP_TRY(hipMemset((uint8_t *)ds.p + (n_bytes & ~(size_t)3), 0, 8 + (n_bytes & 3))); // zero terminator after the last sequence
    FMD_HIP_TRY(hipMemcpy(ds.p, seqs, n_bytes, hipMemcpyHostToDevice));
    TRY_RC(fmd_reach_dev(h, nullptr, n_bytes, (uint8_t *)ds.p, (uint32_t *)dl.p));
    FMD_HIP_TRY(hipMemcpy(len, dl.p, n_bytes * 4, hipMemcpyDeviceToHost));
    return FMD_OK;
}

extern "C" int fmd_retrieve_batch(fmd_dev_t *h, size_t n, const uint64_t *x, uint8_t *seqs, uint32_t stride,
                                  uint32_t *len, uint64_t *rank)
{
    if (!h || (n && (!x || !seqs || !len || !rank || !stride))) return FMD_E_ARG;
    if (n == 0) return FMD_OK;
    FMD_HIP_TRY(hipSetDevice(h->device));
    DevBuf dx, ds, dl, dr;
    TRY_RC(dx.alloc(n * 8)); TRY_RC(ds.alloc(n * (size_t)stride)); TRY_RC(dl.alloc(n * 4)); TRY_RC(dr.alloc(n * 8));
    FMD_HIP_TRY(hipMemcpy(dx.p, x, n * 8, hipMemcpyHostToDevice));
    FMD_HIP_TRY(hipMemset(ds.p, 0, n * (size_t)stride));
    TRY_RC(fmd_retrieve_dev(h, nullptr, n, (uint64_t *)dx.p, (uint8_t *)ds.p, stride, (uint32_t *)dl.p, (uint64_t *)dr.p));
    FMD_HIP_TRY(hipMemcpy(seqs, ds.p, n * (size_t)stride, hipMemcpyDeviceToHost));
    FMD_HIP_TRY(hipMemcpy(len, dl.p, n * 4, hipMemcpyDeviceToHost));
    FMD_HIP_TRY(hipMemcpy(rank, dr.p, n * 8, hipMemcpyDeviceToHost));
    return FMD_OK;
}
