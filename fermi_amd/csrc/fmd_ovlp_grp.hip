// fmd_ovlp_grp.hip -- fm6_get_nei (unitig.c:93-179) with ONE LANE PER CANDIDATE INTERVAL.
//
// fm6_get_nei advances a list of candidate intervals (one per read that might overlap the strand)
// by one base per round.  The lane-per-strand kernel (k_ovl_nei, fmd_ovlp.hip) keeps those lists
// in HBM and pays for it: profiles/r1_ovlp showed 154 GB of traffic per 4 M-strand launch against
// 53 GB of rank blocks.  Here a strand owns a group of G = 16 or 32 lanes and its candidate list
// IS the group's registers:
//   * one wave step = one round of every resident strand (64/G of them): each live lane extends
//     its interval forward (rank2a on the x[1] strand) and, from the SAME step, answers the
//     backward `$` tests of unitig.c:112 and :129 for ok[0] and all four children: they are ranks
//     of '$' at the six child boundaries of the x[0] range [x0-1, x0-1+size], i.e. one more block
//     (two if the range straddles) whose address is known before the extension returns;
//   * the sequential semantics of the reference's loop (first neighbour of a category masks the
//     rest of it; children ordered by old category, base, start) are prefix computations on
//     group ballots; children are re-packed through a 2 KiB LDS staging area;
//   * the next strand of every group is prefetched (descriptor, then candidates) under the rank
//     gathers of the current one.
// Strands that do not fit the fast shape -- more than G candidates, an interval wider than a rank
// block, more neighbours than max_nei, or the fake-fork fix-up of unitig.c:158-176 -- are handed
// to k_ovl_nei through the `slow` work list; nothing is approximated.
#include "fmd_ovlp_common.h"

// LDS: 4 block slots per lane (EXT k/l, B k/l) + staging
#define GRP_SLOTS_U4 (4 * 512)
#define GRP_STAGE_U4 128

// ---------------------------------------------------------------------------- classification
// one thread per strand: work lists for the three get_nei kernels
__global__ void k_ovl_classify(size_t n, const fmd_ovlp_rec_t *__restrict__ rec, const fmd_intv_t *__restrict__ listA, uint32_t cap,
                               FmdOvlClasses cl)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const fmd_ovlp_rec_t *o = rec + i;
    if (o->status != 0 || o->n_ovlp <= 0 || (o->flags & FMD_OVLP_F_OVERFLOW)) return;
    const uint32_t m = (uint32_t)o->n_ovlp;
    // the widest candidate is the last one (shortest overlap): it must fit a rank block so that
    // the six '$' boundaries of a lane live in at most two adjacent blocks
    uint64_t x0, x1, sz, inf;
    load_entry(listA + i * (size_t)cap + (cap - 1), x0, x1, sz, inf);
    int cls = (sz >= 255 || o->len >= 65535) ? 2 : (m <= 16 ? 0 : (m <= 32 ? 1 : 2));
    uint32_t *cnt = cls == 0 ? cl.n16 : cls == 1 ? cl.n32 : cl.nslow;
    uint32_t *lst = cls == 0 ? cl.l16 : cls == 1 ? cl.l32 : cl.lslow;
    const uint32_t k = atomicAdd(cnt, 1u);
    if (cls == 2) lst[k] = (uint32_t)i;
    else { lst[2 * k] = (uint32_t)i; lst[2 * k + 1] = m | (uint32_t)o->len << 16; }
}

// ------------------------------------------------------------------------------ the kernel
__device__ __forceinline__ uint64_t bits_below(int j) { return j >= 64 ? ~0ull : ((1ull << j) - 1); }

// '$' indicator words of a block image and its absolute '$' count
__device__ __forceinline__ void dollar_words(const uint4 *blk, int t, uint32_t w[8], uint64_t &abs0)
{
    uint32_t lo = 0, hi = 0;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const uint4 v = blk[c ^ t];
        w[c] = ~v.x & ~v.y & ~v.z;
        if (c == 0) lo = v.w;
        if (c == 6) hi = v.w & 0xff;
    }
    abs0 = (uint64_t)hi << 32 | lo;
}
__device__ __forceinline__ uint64_t rank0_words(const uint32_t w[8], uint64_t abs0, uint32_t npos)
{
    uint32_t n = 0;
#pragma unroll
    for (int c = 0; c < 8; ++c) n += __builtin_popcount(w[c] & fmd_mask32((int)npos - 32 * c));
    return abs0 + n;
}

template <int G>
__global__ __launch_bounds__(64) void k_ovl_nei_grp(FmdIndexView ix, const uint32_t *__restrict__ list, const uint32_t *__restrict__ list_n,
                                                    uint32_t cap, const fmd_intv_t *__restrict__ listA, fmd_ovlp_rec_t *__restrict__ rec,
                                                    fmd_intv_t *__restrict__ nei_out, uint32_t max_nei, uint8_t *__restrict__ seq_out,
                                                    uint32_t seq_stride, uint32_t *__restrict__ slow_list, uint32_t *__restrict__ slow_n)
{
    __shared__ uint4 lds[GRP_SLOTS_U4 + GRP_STAGE_U4];
    uint4 *stage = lds + GRP_SLOTS_U4;
    constexpr int S = 64 / G;
    constexpr uint32_t GM = G == 32 ? 0xffffffffu : 0xffffu;
    const int lane = fmd_lane(), g = lane / G, j = lane % G, gbase = g * G;
    const uint32_t N = *list_n;
    const uint32_t n_groups = gridDim.x * S;
    uint32_t idx = blockIdx.x * S + g;                   // position of this group's next strand in the list

    // group-uniform strand state (identical in all lanes of the group)
    bool active = false;
    uint32_t sid = 0, n_nei = 0, flags = 0;
    int ori_l = 0, round = 0;
    uint64_t nei0_info = 0;
    // the lane's candidate
    bool alive = false;
    uint64_t x0 = 0, x1 = 0, sz = 0;
    uint32_t pos = 0; int cat = 0;
    // prefetch pipeline: 0 empty, 1 descriptor in flight, 2 candidates in flight / ready
    int pf = 0;
    uint32_t d_sid = 0, d_meta = 0;
    uint4 pa = make_uint4(0, 0, 0, 0), pb = make_uint4(0, 0, 0, 0);

    for (;;) {
        // ---- admission
        if (!active && pf == 2) {
            const uint32_t m = d_meta & 0xffff;
            sid = d_sid; ori_l = (int)(d_meta >> 16); round = 0; n_nei = 0; flags = 0; nei0_info = 0;
            alive = (uint32_t)j < m;
            x0 = (uint64_t)pa.y << 32 | pa.x; x1 = (uint64_t)pa.w << 32 | pa.z;
            sz = ((uint64_t)pb.y << 32 | pb.x) & FMD_SZ_MASK; pos = pb.z; cat = 0;
            active = true;
            pf = 0; idx += n_groups;
        }
        // ---- prefetch pipeline (loads complete under the rank gather below)
        if (pf == 1) { // descriptor has arrived: fetch this lane's candidate
            const uint32_t m = d_meta & 0xffff;
            if ((uint32_t)j < m) { const uint4 *q = (const uint4 *)(listA + d_sid * (size_t)cap + (cap - m) + j); pa = q[0]; pb = q[1]; }
            pf = 2;
        } else if (pf == 0 && idx < N) {
            d_sid = list[2 * (size_t)idx]; d_meta = list[2 * (size_t)idx + 1];
            pf = 1;
        }
        const uint64_t act_m = __ballot(active);
        if (act_m == 0) {
            if (__ballot(pf != 0 || idx < N) == 0) break;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // let the prefetches land
            continue;
        }

        // ---- one round: forward extension + the six '$' boundaries, all from one gather
        const bool live = active && alive;
        const uint64_t ke = live ? x1 - 1 : NONE64, le = live ? x1 - 1 + sz : NONE64;
        const uint64_t kb = live ? x0 - 1 : NONE64, lb = live ? x0 - 1 + sz : NONE64; // x0 >= mcnt[1] > 0 for base strings
        const uint32_t bke = (uint32_t)(ke >> FMD_BLK_SHIFT), ble = (uint32_t)(le >> FMD_BLK_SHIFT);
        const uint32_t bkb = (uint32_t)(kb >> FMD_BLK_SHIFT), blb = (uint32_t)(lb >> FMD_BLK_SHIFT);
        const bool e_sep = live && ble != bke, b_sep = live && blb != bkb;
        fmd_fetch_slot<0>(ix, lds, bke, live);
        fmd_fetch_slot<1>(ix, lds, ble, e_sep);
        fmd_fetch_slot<2>(ix, lds, bkb, live);
        fmd_fetch_slot<3>(ix, lds, blb, b_sep);
        fmd_fetch_wait();

        uint64_t s[6] = {0, 0, 0, 0, 0, 0}, tk[6] = {0, 0, 0, 0, 0, 0};
        bool is_nei = false;
        uint32_t cm = 0;                    // children c = 1..4 that survive the sentinel test
        uint64_t e0x0 = 0;                  // x[0] of the `$...$` interval when this lane is a neighbour
        if (live) {
            const int t = fmd_chunk_xor(lane);
            uint64_t tl[6];
            fmd_block_rank6<false>(lds + fmd_lds_base(lane, 0), t, ((uint32_t)ke & 255) + 1, tk);
            fmd_block_rank6<false>(lds + fmd_lds_base(lane, e_sep ? 1 : 0), t, ((uint32_t)le & 255) + 1, tl);
#pragma unroll
            for (int c = 0; c < 6; ++c) s[c] = tl[c] - tk[c];
            // '$' ranks at the child boundaries of the x[0] range, order $,T,G,C,A (exact.c:81-86)
            uint32_t wk[8], wl[8];
            uint64_t ak, al;
            dollar_words(lds + fmd_lds_base(lane, 2), t, wk, ak);
            if (b_sep) dollar_words(lds + fmd_lds_base(lane, 3), t, wl, al);
            else {
#pragma unroll
                for (int c = 0; c < 8; ++c) wl[c] = wk[c];
                al = ak;
            }
            uint64_t b = kb, R[6];
            const uint64_t step[5] = {s[0], s[4], s[3], s[2], s[1]};
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                const bool in_k = (uint32_t)(b >> FMD_BLK_SHIFT) == bkb;
                uint32_t w[8];
#pragma unroll
                for (int c = 0; c < 8; ++c) w[c] = in_k ? wk[c] : wl[c];
                R[i] = rank0_words(w, in_k ? ak : al, ((uint32_t)b & 255) + 1);
                if (i < 5) b += step[i];
            }
            const uint64_t e0sz = R[1] - R[0];
            // unitig.c:111-122: a read ends here, bounded by sentinels on both sides, not contained
            is_nei = round > 0 && s[0] && e0sz && s[0] == sz && sz == e0sz;
            e0x0 = R[0];
            if (s[4] && R[2] != R[1]) cm |= 1u << 4;
            if (s[3] && R[3] != R[2]) cm |= 1u << 3;
            if (s[2] && R[4] != R[3]) cm |= 1u << 2;
            if (s[1] && R[5] != R[4]) cm |= 1u << 1;
        }

        // ---- the reference's sequential loop over the list, as prefix logic on group ballots
        const uint32_t alive_g = (uint32_t)(__ballot(live) >> gbase) & GM;
        const int ncur = __popc(alive_g);
        const uint32_t nei_g = (uint32_t)(__ballot(is_nei) >> gbase) & GM;
        const uint32_t head_g = (uint32_t)(__ballot(live && cat == j) >> gbase) & GM;   // first lane of each category
        const uint32_t in_cat_upto_j = (uint32_t)(bits_below(j + 1) & ~bits_below(cat));
        const bool new_nei = is_nei && (nei_g & in_cat_upto_j & (uint32_t)bits_below(j)) == 0; // first neighbour of its category
        const bool keep = live && (nei_g & in_cat_upto_j) == 0;                                 // not masked, not a neighbour
        const uint32_t newnei_g = (uint32_t)(__ballot(new_nei) >> gbase) & GM;
        if (!keep) cm = 0;
        // neighbours, in list order (unitig.c:119-121)
        if (new_nei) {
            const uint32_t k = n_nei + __popc(newnei_g & (uint32_t)bits_below(j));
            if (k < max_nei) store_entry(nei_out + sid * (size_t)max_nei + k, e0x0, ix.cnt[0] + tk[0], sz, (uint64_t)ori_l - pos);
        }
        if (active && n_nei == 0 && newnei_g) { // info of nei[0] decides rbeg (unitig.c:157)
            const int src = gbase + __ffs((int)newnei_g) - 1;
            nei0_info = (uint64_t)ori_l - (uint32_t)__shfl((int)pos, src);
        }
        if (active) n_nei += __popc(newnei_g);

        // children: destination = #children of earlier categories + same category & smaller base
        //           + same category, same base, earlier lane  (= ks_introsort by info, unitig.c:140)
        const uint32_t c1 = (uint32_t)(__ballot((cm >> 1) & 1) >> gbase) & GM, c2 = (uint32_t)(__ballot((cm >> 2) & 1) >> gbase) & GM;
        const uint32_t c3 = (uint32_t)(__ballot((cm >> 3) & 1) >> gbase) & GM, c4 = (uint32_t)(__ballot((cm >> 4) & 1) >> gbase) & GM;
        const int n_new = __popc(c1) + __popc(c2) + __popc(c3) + __popc(c4);
        const uint32_t later_heads = (cat + 1 >= 32) ? 0u : (head_g >> (cat + 1));
        const int cat_end = later_heads ? cat + 1 + (__ffs((int)later_heads) - 1) : ncur;
        const uint32_t before_m = (uint32_t)bits_below(cat), same_m = (uint32_t)(bits_below(cat_end) & ~bits_below(cat));
        const uint32_t lt_m = same_m & (uint32_t)bits_below(j);
        const int before = __popc(c1 & before_m) + __popc(c2 & before_m) + __popc(c3 & before_m) + __popc(c4 & before_m);
        const int p1 = before, p2 = p1 + __popc(c1 & same_m), p3 = p2 + __popc(c2 & same_m), p4 = p3 + __popc(c3 & same_m);
        const bool too_many = n_new > G;
        bool forked_now = false;
        // x[0] of the children: running sum $,T,G,C,A; x[1] = cnt[c] + tk[c]
        const uint64_t cx0_4 = x0 + s[0], cx0_3 = cx0_4 + s[4], cx0_2 = cx0_3 + s[3], cx0_1 = cx0_2 + s[2];
        if (!too_many) {
#define GRP_PUSH(c, pc, cmask, cx0)                                                                     \
            if ((cm >> c) & 1) {                                                                        \
                const int d = pc + __popc(cmask & lt_m);                                                \
                const uint64_t nx1 = ix.cnt[c] + tk[c];                                                 \
                stage[2 * (gbase + d)] = make_uint4((uint32_t)(cx0), (uint32_t)((cx0) >> 32), (uint32_t)nx1, (uint32_t)(nx1 >> 32)); \
                stage[2 * (gbase + d) + 1] = make_uint4((uint32_t)s[c], (uint32_t)(s[c] >> 32), pos, (uint32_t)pc); \
                forked_now |= pc != 0;                                                                  \
            }
            GRP_PUSH(1, p1, c1, cx0_1) GRP_PUSH(2, p2, c2, cx0_2) GRP_PUSH(3, p3, c3, cx0_3) GRP_PUSH(4, p4, c4, cx0_4)
#undef GRP_PUSH
        }
        const uint32_t fork_g = (uint32_t)(__ballot(forked_now) >> gbase) & GM;
        // base appended this round = base of the first child in push order (unitig.c:138-139)
        const uint32_t anyc_g = c1 | c2 | c3 | c4;
        int first_c = 0;
        if (anyc_g) {
            const int src = gbase + __ffs((int)anyc_g) - 1;
            const uint32_t cm_src = (uint32_t)__shfl((int)cm, src);
            first_c = __ffs((int)cm_src) - 1;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // staging writes done before the re-pack reads
        if (active) {
            if (fork_g) flags |= FMD_OVLP_F_FORKED;
            if (too_many || n_nei > max_nei) { // hand the strand to the lane-per-strand kernel
                if (j == 0) { const uint32_t k = atomicAdd(slow_n, 1u); slow_list[k] = sid; }
                active = false; alive = false;
            } else if (n_new > 0) { // next round (unitig.c:137-153)
                if (j == 0 && (uint32_t)(ori_l + round) < seq_stride) seq_out[sid * (size_t)seq_stride + ori_l + round] = (uint8_t)comp6(first_c);
                ++round;
                alive = j < n_new;
                if (alive) {
                    const uint4 a = stage[2 * (gbase + j)], b = stage[2 * (gbase + j) + 1];
                    x0 = (uint64_t)a.y << 32 | a.x; x1 = (uint64_t)a.w << 32 | a.z;
                    sz = (uint64_t)b.y << 32 | b.x; pos = b.z; cat = (int)b.w;
                }
            } else { // every path is closed (unitig.c:154-178)
                if (n_nei == 1 && (flags & FMD_OVLP_F_FORKED)) { // fake fork: the fix-up needs the slow kernel
                    if (j == 0) { const uint32_t k = atomicAdd(slow_n, 1u); slow_list[k] = sid; }
                } else if (j == 0) {
                    fmd_ovlp_rec_t *o = rec + sid;
                    o->rbeg = n_nei ? ori_l - (int)(uint32_t)nei0_info : -1;
                    o->ext_len = n_nei > 1 ? 0 : round;
                    o->n_nei = (int32_t)n_nei;
                    o->flags |= flags;
                }
                active = false; alive = false;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // staging reads done before the next round overwrites it
    }
}

// explicit instantiations + launcher used by fmd_ovlp.hip
void fmd_launch_nei_grp(int G, int grid, hipStream_t st, const FmdIndexView &ix, const uint32_t *list, const uint32_t *list_n, uint32_t cap,
                        const fmd_intv_t *listA, fmd_ovlp_rec_t *rec, fmd_intv_t *nei_out, uint32_t max_nei, uint8_t *seq_out,
                        uint32_t seq_stride, uint32_t *slow_list, uint32_t *slow_n)
{
    if (G == 16) k_ovl_nei_grp<16><<<grid, 64, 0, st>>>(ix, list, list_n, cap, listA, rec, nei_out, max_nei, seq_out, seq_stride, slow_list, slow_n);
    else k_ovl_nei_grp<32><<<grid, 64, 0, st>>>(ix, list, list_n, cap, listA, rec, nei_out, max_nei, seq_out, seq_stride, slow_list, slow_n);
}
void fmd_launch_classify(hipStream_t st, size_t n, const fmd_ovlp_rec_t *rec, const fmd_intv_t *listA, uint32_t cap, FmdOvlClasses cl)
{
    k_ovl_classify<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(n, rec, listA, cap, cl);
}
