// fmd_ovlp_grp.hip -- fm6_get_nei (unitig.c:93-179) with ONE LANE PER CANDIDATE INTERVAL.
//
// fm6_get_nei advances a list of candidate intervals (one per read that might overlap the strand)
// by one base per round.  The lane-per-strand kernel (k_ovl_nei, fmd_ovlp.hip) keeps those lists
// in HBM and pays for it: profiles/r1_ovlp showed 154 GB of traffic per 4 M-strand launch against
// 53 GB of rank blocks.  Here a strand owns a group of G lanes (G = 8, 12, 16, 21 or 32: the smallest
// that holds its candidates, fmd_grp_size) and its candidate list IS the group's registers:
//   * one wave step = one round of every resident strand (64/G of them): each live lane extends
//     its interval forward (rank2a on the x[1] strand) and answers the backward `$` tests of
//     unitig.c:112 and :129 for ok[0] and all four children from what it CARRIES: D, the '$'
//     positions of BWT[x0, x0 + size), and r0, the '$' before x0 (see the kernel: the x[0] side of
//     the index is read once per candidate at most);
//   * the sequential semantics of the reference's loop (first neighbour of a category masks the
//     rest of it; children ordered by old category, base, start) are prefix computations on
//     group ballots; children are re-packed through a 2 KiB LDS staging area (it reuses the spill pool);
//   * the next strand of every group is prefetched (descriptor, then candidates) under the rank
//     gathers of the current one.
// Strands that do not fit the fast shape -- more than 32 candidates, an interval wider than 63,
// more neighbours than max_nei, or the fake-fork fix-up of unitig.c:158-176 -- are handed
// to k_ovl_nei through the `slow` work list; nothing is approximated.
#include "fmd_kernel_common.h"

// LDS per wave: one block slot per lane for the k side of the forward extension (slot 0) and, in the first round of
// a candidate that came without D / r0, of its x[0] range (slot 1); the l sides, needed only when a range leaves its
// block, share a compacted pool of 30 blocks; a wave step that needs more (0.2 % of them) takes the pool
// in several passes.  One 2 KiB region holds the pool (1920 bytes), the block numbers to fetch
// (128 bytes) and -- later in the step, when the windows have been read -- the re-pack staging
// area.  10 KiB per wave = 8 LDS granules -> 16 waves per CU.
#ifndef GRP_POOL
#define GRP_POOL 30                          // spill blocks per pass (-DGRP_POOL=4 stresses the multi-pass path in the tests)
#endif
#define GRP_SLOTS_U4 (2 * FMD_SLOT_U4)
#define GRP_STAGE_U4 128
#define GRP_REGION_U4 (GRP_POOL * FMD_BLK_U4 + 8 > GRP_STAGE_U4 ? GRP_POOL * FMD_BLK_U4 + 8 : GRP_STAGE_U4) // 128 with 64-byte blocks
#define GRP_LDS_U4 (GRP_SLOTS_U4 + GRP_REGION_U4)
static_assert(GRP_POOL <= 32, "the region ends with 32 block numbers");

// ---------------------------------------------------------------------------- classification
// one thread per strand: work lists for the get_nei kernels.  Positions come from a block-wide count
// (ballots per wave, LDS across the 16 waves) and ONE atomic per block and class, each counter on
// its own 128-byte line: per-wave atomics on adjacent words cost 7-15 ms per 2*10^7 strands.
#define CLS_THREADS 1024
#define CLS_LISTS FMD_CLS_LISTS            // general class k = k, slow = FMD_GRP_CLASSES, fast class k = FMD_GRP_CLASSES + 1 + k (+ FMD_GRP_CLASSES: 64-bit masks)
__global__ __launch_bounds__(CLS_THREADS) void k_ovl_classify(size_t n, const fmd_ovlp_rec_t *__restrict__ rec, const fmd_intv_t *__restrict__ listA,
                                                              uint32_t cap, FmdOvlClasses cl, int use_fast, const uint32_t *__restrict__ gidx, int min_cls)
{
    __shared__ uint32_t wcnt[CLS_THREADS / 64][CLS_LISTS], base[CLS_LISTS];
    const size_t i = (size_t)blockIdx.x * CLS_THREADS + threadIdx.x;
    int cls = -1;
    uint32_t m = 0, len = 0;
    if (i < n) {
        const fmd_ovlp_rec_t *o = rec + (gidx ? (size_t)gidx[i] : i);   // (sorted batches: the record's row is not the slot, fmd_ovlp_sorted_dev)
        if (o->status == 0 && o->n_ovlp > 0 && !(o->flags & FMD_OVLP_F_OVERFLOW)) {
            m = (uint32_t)o->n_ovlp; len = (uint32_t)o->len;
            // the widest candidate is the last one (shortest overlap): the group kernels count
            // symbols of BWT[x, x+size) through a 64-position window
            const uint4 *q = (const uint4 *)(listA + i * (size_t)cap + (cap - 1));
            const FmdCand w = cand_decode(q[0], q[1]);
            cls = FMD_GRP_CLASSES;
            if (w.sz <= 63 && len < 65535) {
#pragma unroll
                for (int k = FMD_GRP_CLASSES - 1; k >= 0; --k) if (k >= min_cls && m <= (uint32_t)fmd_grp_size(k)) cls = k;
                // the widest candidate was pushed first; the fast kernel checks the others when it loads them
                if (cls < FMD_GRP_CLASSES && w.narrow && use_fast) cls += FMD_GRP_CLASSES + 1 + (w.sz > 31 ? FMD_GRP_CLASSES : 0);
            }
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t in_wave = 0;                      // lanes of this wave before me in my class
#pragma unroll
    for (int c = 0; c < CLS_LISTS; ++c) {
        const uint64_t mk = __ballot(cls == c);
        if (lane == 0) wcnt[wave][c] = (uint32_t)__popcll(mk);
        if (cls == c) in_wave = (uint32_t)fmd_below(mk);
    }
    __syncthreads();
    if (threadIdx.x < CLS_LISTS) {             // exclusive scan over the waves, then the block's slice of the list
        uint32_t tot = 0;
        for (int w = 0; w < CLS_THREADS / 64; ++w) { const uint32_t v = wcnt[w][threadIdx.x]; wcnt[w][threadIdx.x] = tot; tot += v; }
        base[threadIdx.x] = tot ? atomicAdd(cl.cnt + threadIdx.x * FMD_CLS_CNT_STRIDE, tot) : 0;
    }
    __syncthreads();
    if (cls >= 0) {
        const uint32_t k = base[cls] + wcnt[wave][cls] + in_wave;
        if (cls == FMD_GRP_CLASSES) cl.lslow[k] = (uint32_t)i;
        else { uint32_t *lst = cls < FMD_GRP_CLASSES ? cl.lst[cls] : cl.fast[cls - FMD_GRP_CLASSES - 1]; lst[2 * k] = (uint32_t)i; lst[2 * k + 1] = m | len << 16; }   // fast[5..9]: the 64-bit lists
    }
}

// ------------------------------------------------------------------------------ dealing the work list
// The strands of a work list used to be dealt to the groups round-robin (group q: positions q, q + n_groups, ...): a group's share is then fixed at
// the launch, the rounds a strand takes are not (1 .. 50 on reads with errors), and the kernel ends when the group with the longest share does --
// 11 % of all group rounds of k_ovl_nei_grp and 10-17 % of k_ovl_nei_fast's were groups with nothing left (GRP_STATS, 10^7 reads with 1 % errors).
// DYN: a wave reserves CHUNKS of consecutive positions from one counter (one atomic per chunk: atomics on one address serialise at ~14 ns; guided
// sizes 16 .. FMD_DEAL_CHUNK, the first chunk of every wave without an atomic), a group asks when its prefetch slot is empty.  Consecutive positions still go to groups that run
// at the same time, which is what the order of the list is for.  The counter is word 16 of the list counter's own 128-byte line (zeroed
// with the header of the batch).  Returns the position for every lane of a group that asked (`want`: group-uniform), 0xffffffff otherwise and once
// the list is dealt out (`dry`).
#ifndef FMD_DEAL_CHUNK
#define FMD_DEAL_CHUNK 64
#endif
#define FMD_DEAL_WORD 16
static int nei_dyn(void)   // FMD_NEI_DYN=0: A/B switch, the work lists dealt round-robin as before
{
    const char *e = getenv("FMD_NEI_DYN");
    return !(e && atoi(e) == 0);
}
struct FmdDeal { uint32_t cur, end; };   // wave-uniform: positions [cur, end) of the list are this wave's
// a wave's FIRST chunk costs no atomic: wave w owns [w * c0, w * c0 + c0), the counter starts behind the last wave's.  c0 is one strand per group
// for a list shorter than the grid (every strand starts at once, as in the round-robin deal; a wave without a strand leaves at once)
template <int G>
__device__ __forceinline__ uint32_t fmd_deal_first(uint32_t N)
{
    const uint32_t g = N / (2u * gridDim.x);
    return g < (uint32_t)(64 / G) ? (uint32_t)(64 / G) : (g > (uint32_t)FMD_DEAL_CHUNK ? (uint32_t)FMD_DEAL_CHUNK : g);
}
template <int G>
__device__ __forceinline__ void fmd_deal_init(FmdDeal &q, uint32_t N, bool &dry)
{
    const uint32_t c0 = fmd_deal_first<G>(N);
    q.cur = blockIdx.x * c0; q.end = q.cur + c0;
    dry = q.cur >= N;
}
template <int G>
__device__ __forceinline__ uint32_t fmd_deal_next(FmdDeal &q, uint32_t *deal, uint32_t N, bool want, bool &dry)
{
    const int lane = fmd_lane(), gbase = lane / G * G;
    const uint64_t m = __ballot(want && !dry && lane == gbase);
    if (m == 0) return 0xffffffffu;
    const uint32_t need = (uint32_t)__popcll(m), r = (uint32_t)__popcll(m & ((1ull << gbase) - 1)), avail = q.end - q.cur;   // r: this group's place among those that ask
    uint32_t pos;
    if (avail < need) {   // the rest of this chunk, then a new one: 1 / (2 * waves) of what is left of the list, between 16 (>= the groups of a wave) and FMD_DEAL_CHUNK
        const uint32_t rem = N > q.end ? N - q.end : 0u, gsz = rem / (2u * gridDim.x), sz = gsz < 16u ? 16u : (gsz > (uint32_t)FMD_DEAL_CHUNK ? (uint32_t)FMD_DEAL_CHUNK : gsz);
        uint32_t v = 0;
        if (lane == 0) v = atomicAdd(deal, sz);
        v = (uint32_t)__builtin_amdgcn_readfirstlane((int)v) + gridDim.x * fmd_deal_first<G>(N);
        pos = r < avail ? q.cur + r : v + (r - avail);
        q.cur = v + (need - avail); q.end = v + sz;
    } else { pos = q.cur + r; q.cur += need; }
    if (q.cur >= N) dry = true;          // positions only grow: nothing is left for this wave (a position >= N is no strand: the caller tests)
    return want ? pos : 0xffffffffu;
}

// ------------------------------------------------------------------------------ the kernel
// A lane's candidate is (x[1], size, D, r0, pos): D = the positions of '$' in BWT[x[0], x[0] + size) (the reads that START with the
// candidate string: the sentinel tests of unitig.c:112 / :129), r0 = the number of '$' before x[0] (x[0] of a neighbour interval,
// unitig.c:113).  Both follow an extension without touching memory -- the x[0] range of child c is the sub-range [o_c, o_c + s_c) of
// its parent's, in the order $,T,G,C,A,N (exact.c:81-86): D shifts, r0 counts along -- so the x[0] side of the index is read ONCE per
// candidate at most: by the walk for candidates it pushes in the narrow form (fmd_kernel_common.h), else in the candidate's first
// round here.  x[0] itself is never needed.  (Round 2 fetched the x[0] block of every lane every round: a second dense slot, a second
// window, a second spill list.)  It is also the state k_ovl_nei_fast holds when a strand stops being simple, so that kernel hands
// its strands on where they stand (FMD_LIST_RESUME) instead of at round 0.
template <int G, bool DYN>
__global__ __launch_bounds__(64, 4) void k_ovl_nei_grp(FmdIndexView ix, const uint32_t *__restrict__ list, const uint32_t *__restrict__ list_n,
                                                    uint32_t cap, const fmd_intv_t *__restrict__ listA, fmd_intv_t *listB, FmdOvlClasses cl,
                                                    fmd_ovlp_rec_t *__restrict__ rec,
                                                    fmd_intv_t *__restrict__ nei_out, uint32_t max_nei, uint8_t *__restrict__ seq_out,
                                                    uint32_t seq_stride, uint32_t *__restrict__ slow_list, uint32_t *__restrict__ slow_n,
                                                    const uint32_t *__restrict__ gidx, size_t fix_off, uint32_t *__restrict__ deal, uint32_t down_cap, uint32_t n_max, int quiet_on)
{
    __shared__ uint4 lds[GRP_LDS_U4];
    uint4 *pool = lds + GRP_SLOTS_U4, *stage = pool;
    uint32_t *pool_blk = (uint32_t *)(pool + GRP_REGION_U4 - 8);
    constexpr int S = 64 / G;
    constexpr uint32_t GM = G == 32 ? 0xffffffffu : (1u << G) - 1;
    const int lane = fmd_lane(), g = lane / G, j = lane % G, gbase = g * G;
    // (n_max: a second-pass list -- its counter also counts the chunks that were asked for when it was full; those entries do not exist)
    const uint32_t N = n_max && *list_n > n_max ? n_max : *list_n;
    const uint32_t n_groups = gridDim.x * S;
    uint32_t idx = !DYN && g < S ? blockIdx.x * S + g : 0xffffffffu; // position of this group's next strand in the list (lanes past S * G idle)
    // DYN: the positions are dealt out instead (fmd_deal_next below) -- dealt round-robin, a group's share of the list is fixed at the launch and
    // the kernel ends when the group with the longest share does (rounds per strand: 1 .. 50), with 11 % of all group rounds idle on reads with errors
    FmdDeal tk;
    bool dry = !DYN;
    if (N == 0) return;
    if (DYN) { fmd_deal_init<G>(tk, N, dry); if (dry) return; }

    // group-uniform strand state (identical in all lanes of the group)
    bool active = false;
    uint32_t sid = 0, gs = 0, n_nei = 0, flags = 0;   // sid: the strand's slot in the batch (listA rows, work lists); gs: its row in rec / nei_out / seq_out
    // bits 8..24 of `flags` hold lfork (FMD_LFORK_* of include/fmd_hip.h) while the strand is resident; bit 24 = closed
#define LF_SHIFT 8
#define LF_GET(f) (((f) >> LF_SHIFT) & 0x1ffffu)
#define LF_SET(f, v) ((f) = ((f) & ~(0x1ffffu << LF_SHIFT)) | ((uint32_t)(v) << LF_SHIFT))
    int ori_l = 0, round = 0;
    uint32_t nei0_info = 0;
    // the lane's candidate
    bool alive = false, need_d = false;   // need_d: D and r0 are still to be read from the index (a candidate the walk pushed in the wide form)
    uint64_t x1 = 0, D = 0, r0 = 0;
    uint32_t sz = 0, pos = 0; int cat = 0;
    // prefetch pipeline: 0 empty, 1 descriptor in flight, 2 candidates in flight / ready
    int pf = 0;
    uint32_t d_sid = 0, d_meta = 0, d_gs = 0;
    uint4 pa = make_uint4(0, 0, 0, 0), pb = make_uint4(0, 0, 0, 0);
    // A strand keeps the group it was admitted to while its candidates die (reads with errors: a read that differs from the others is a category
    // of its own from that base to its end, up to 50 rounds on, one lane of the group's G: 39 % live lanes on 30x reads with 1 % errors).  The sum
    // of the live candidates' interval sizes bounds the lanes the strand can EVER need again (every child takes at least one occurrence of its
    // parent's with it), so a strand whose live candidates are all single occurrences and fit a smaller group moves there for good -- as it stands
    // after the round (FMD_LIST_RESUME), to the second-pass list of that class (the storage of its fast list, idle by now; counter in word
    // FMD_DOWN_WORD of that list's line), which the launcher runs after the first pass, largest class first.  Slots are reserved FMD_FAST_CHUNK at a
    // time per wave and target class (one atomic per strand on one address serialises), unused ones stay holes.  down_cap = entries a second-pass
    // list holds (0: strands stay where they are, the A/B switch FMD_GRP_DOWN=0).
    constexpr int MYK = G == 4 ? 0 : G == 8 ? 1 : G == 12 ? 2 : G == 16 ? 3 : G == 21 ? 4 : 5;
    constexpr int NT = !DYN ? 0 : MYK >= 2 ? 2 : MYK;   // target classes: MYK - 1 and MYK - 2 (further down in one more hop from there); not with the round-robin deal (FMD_NEI_DYN=0, an A/B form: its registers are full)
    uint32_t mv = 0;                                    // group-uniform: 1 + t = this group's strand leaves for target t (class MYK - 1 - t)
    bool chk_down = false;                              // group-uniform: a round has just ended, the strand goes on
    uint32_t res_cur[NT > 0 ? NT : 1] = {0}, res_end[NT > 0 ? NT : 1] = {0};   // wave-uniform

    for (;;) {
        if (NT > 0 && down_cap) {   // ---- strands that leave for a smaller group
            if (__ballot(chk_down)) {   // after a round: does the strand fit a smaller group for good?  W = live candidates + sum (size - 1), from ballots (a size of 5 or more: no move)
                const uint32_t ag = (uint32_t)(__ballot(alive) >> gbase) & GM, nl = (uint32_t)__popc(ag);
                const uint32_t b2 = (uint32_t)(__ballot(alive && sz >= 2) >> gbase) & GM, b3 = (uint32_t)(__ballot(alive && sz >= 3) >> gbase) & GM;
                const uint32_t b4 = (uint32_t)(__ballot(alive && sz >= 4) >> gbase) & GM, b5 = (uint32_t)(__ballot(alive && sz >= 5) >> gbase) & GM;
                const uint32_t Wsum = nl + (uint32_t)(__popc(b2) + __popc(b3) + __popc(b4));
                if (chk_down && active && nl > 0 && b5 == 0 && Wsum <= (uint32_t)fmd_grp_size(MYK - 1) && nl <= cap && fmd_resume_fits((uint32_t)round, n_nei, nei0_info, 0u))
                    mv = NT > 1 && Wsum <= (uint32_t)fmd_grp_size(MYK >= 2 ? MYK - 2 : 0) ? 2u : 1u;
                chk_down = false;
            }
            if (__ballot(mv != 0)) {
#pragma unroll
                for (int tt = 0; tt < NT; ++tt) {
                    const uint64_t hm = __ballot(mv == (uint32_t)(tt + 1) && j == 0 && g < S);
                    if (hm == 0) continue;
                    const int k2 = MYK - 1 - tt;
                    uint32_t *lst2 = cl.fast[k2], *cnt2 = cl.cnt + (FMD_GRP_CLASSES + 1 + k2) * FMD_CLS_CNT_STRIDE + FMD_DOWN_WORD;
                    const uint32_t n = (uint32_t)__popcll(hm), room = res_end[tt] - res_cur[tt];   // n <= S <= FMD_FAST_CHUNK
                    uint32_t base = 0;
                    bool got = true;
                    if (room < n) {
                        if (lane == 0) base = atomicAdd(cnt2, (uint32_t)FMD_FAST_CHUNK);
                        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                        got = base + FMD_FAST_CHUNK <= down_cap;                      // (a full list: the strands beyond `room` stay here)
                        if (got) {
                            if (lane < FMD_FAST_CHUNK) { lst2[2 * (size_t)(base + lane)] = FMD_LIST_HOLE; lst2[2 * (size_t)(base + lane) + 1] = 0; }
                            asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the markers land before the entries that replace them
                        }
                    }
                    const uint32_t q = (uint32_t)__popcll(hm & ((1ull << gbase) - 1));   // this group's place among those that leave for k2
                    const bool mine = mv == (uint32_t)(tt + 1) && (q < room || got);
                    if (mine) {
                        const uint32_t k = q < room ? res_cur[tt] + q : base + (q - room);
                        const uint32_t ag = (uint32_t)(__ballot(alive) >> gbase) & GM;    // (lanes j < the number of children: packed already)
                        if (alive) fmd_resume_encode((uint4 *)(listB + sid * (size_t)cap + j), x1, sz, D, r0, pos, (uint32_t)round, n_nei, nei0_info, LF_GET(flags), (uint32_t)cat, flags & 0xffu);
                        if (j == 0) { lst2[2 * (size_t)k] = sid; lst2[2 * (size_t)k + 1] = (uint32_t)__popc(ag) | FMD_LIST_RESUME | (uint32_t)ori_l << 16; }
                        active = false; alive = false;
                    }
                    if (room < n) { if (got) { res_cur[tt] = base + (n - room); res_end[tt] = base + FMD_FAST_CHUNK; } else res_cur[tt] = res_end[tt]; }
                    else res_cur[tt] += n;
                }
                mv = 0;
            }
        }
        // ---- admission
        if (!active && pf == 2) {
            const uint32_t m = d_meta & 0x7fffu;
            sid = d_sid; gs = d_gs; ori_l = (int)(d_meta >> 16); flags = 0; cat = 0;
            alive = (uint32_t)j < m;
            if (d_meta & FMD_LIST_RESUME) { // the strand as k_ovl_nei_fast left it before round `round` (fmd_resume_*: every entry carries the strand's state)
                uint32_t lf_, rd_, nn_, n0_, ct_, fl_;
                fmd_resume_decode(pa, pb, x1, sz, D, r0, pos, rd_, nn_, n0_, lf_, ct_, fl_);
                round = (int)(uint32_t)__shfl((int)rd_, gbase); n_nei = (uint32_t)__shfl((int)nn_, gbase);   // (lanes past m hold no entry)
                nei0_info = (uint32_t)__shfl((int)n0_, gbase); LF_SET(flags, (uint32_t)__shfl((int)lf_, gbase));
                flags |= (uint32_t)__shfl((int)fl_, gbase); cat = (int)ct_;
                need_d = false;
            } else {
                const FmdCand cd = cand_decode(pa, pb);
                x1 = cd.x1; sz = (uint32_t)cd.sz; pos = (uint32_t)ori_l - cd.depth; // stored: suffix depth
                D = cd.D; r0 = cd.r0; need_d = alive && !cd.narrow;
                round = 0; n_nei = 0; nei0_info = 0;
            }
            active = true;
            pf = 0; if (!DYN) idx += n_groups;
        }
        // ---- prefetch pipeline (loads complete under the rank gather below)
        if (pf == 1 && d_sid == FMD_LIST_HOLE) { pf = 0; if (!DYN) idx += n_groups; }   // an unused slot of a chunk the fast kernel reserved
        if (DYN) idx = fmd_deal_next<G>(tk, deal, N, pf == 0 && g < S, dry);
        if (pf == 1) { // descriptor has arrived: fetch this lane's candidate
            const uint32_t m = d_meta & 0x7fffu;
            if ((uint32_t)j < m) {
                const uint4 *q = (const uint4 *)((d_meta & FMD_LIST_RESUME) ? listB + d_sid * (size_t)cap + j : listA + d_sid * (size_t)cap + (cap - m) + j);
                pa = q[0]; pb = q[1];
            }
            d_gs = gidx ? gidx[d_sid] : d_sid;
            pf = 2;
        } else if (pf == 0 && idx < N) {
            d_sid = list[2 * (size_t)idx]; d_meta = list[2 * (size_t)idx + 1];
            pf = 1;
        }
        const uint64_t act_m = __ballot(active);
        if (act_m == 0) {
            if (__ballot(pf != 0 || (!DYN && idx < N)) == 0 && dry) break;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // let the prefetches land
            continue;
        }

        // ---- one round: forward extension (symbols of BWT[x1, x1 + size)) from one gather
        const bool live = active && alive;
#ifdef GRP_STATS
        { const uint32_t nl = (uint32_t)__popcll(__ballot(live)); if (lane == 0) { atomicAdd(slow_n + 8, 1u); atomicAdd(slow_n + 9, nl); atomicAdd(slow_n + 10, (uint32_t)__popcll(act_m)); } }
#endif
        const uint64_t ke = live ? x1 - 1 : NONE64;
        // the l side lies at most 63 positions after the k side: same block or the next one
        uint32_t bke, oke;
        fmd_split(ke, bke, oke);
        const uint32_t ble = bke + (oke + sz >= FMD_BLK_SYMS);
        const bool e_sep = live && ble != bke;
        // the x[0] side: only for candidates still without D / r0, i.e. in the first round of a candidate in the wide form
        const bool nb = live && need_d;
        const bool any_nb = __ballot(nb) != 0;
        uint32_t bkb = 0, okb = 0, blb = 0, x0lo = 0;
        bool b_sep = false;
        if (any_nb) {
            const uint64_t x0 = cand_decode(pa, pb).x0;   // (pa / pb still hold the entry: the prefetch of the next strand has only fetched its descriptor)
            const uint64_t kb = nb ? x0 - 1 : NONE64;     // x0 >= mcnt[1] > 0 for base strings
            fmd_split(kb, bkb, okb);
            blb = bkb + (okb + sz >= FMD_BLK_SYMS);
            b_sep = nb && blb != bkb; x0lo = (uint32_t)x0;
        }
        fmd_fetch_slot<0>(ix, lds, bke, live);
        if (any_nb) fmd_fetch_slot<1>(ix, lds, bkb, nb);
        // straddling ranges: compact the extra blocks into the pool (ballot prefix), 16 per instruction,
        // GRP_POOL per pass.  Every interval here has size <= 63 (k_ovl_classify), so "rank2a" is a count
        // over a 64-position window of the planes read straight from the lane's LDS block images.
        // A lane reads its windows in the pass that brings its spill block(s), or in the first one.
        const uint64_t me = __ballot(e_sep), mb = __ballot(b_sep);
        const int n_e = __popcll(me), n_spill = n_e + __popcll(mb);
        const int pe = fmd_below(me), pb_ = n_e + fmd_below(mb);
        const int t = fmd_chunk_xor(lane);
        const uint4 *img_e = lds + fmd_lds_base(lane, 0), *img_b = lds + fmd_lds_base(lane, 1);
        uint64_t X = 0, Y = 0, Z = 0;
        bool need_e = live, need_b = nb;
        for (int base = 0;; base += GRP_POOL) {
            const int re = pe - base, rb = pb_ - base;
            const bool in_e = e_sep && re >= 0 && re < GRP_POOL, in_b = b_sep && rb >= 0 && rb < GRP_POOL;
            const int n_here = n_spill - base < GRP_POOL ? n_spill - base : GRP_POOL;
            if (n_here > 0) {
                if (in_e) pool_blk[re] = ble;
                if (in_b) pool_blk[rb] = blb;
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                fmd_fetch_pool(ix, pool, pool_blk, n_here);
            }
            fmd_fetch_wait();
            if (need_e && (!e_sep || in_e)) {   // window of BWT[x1 ...]; x1 may sit in the block after ke's (ke = x1-1)
                uint4 a, b, c;
                grp_window(img_e, t, pool + (in_e ? re : 0) * FMD_BLK_U4, fmd_pool_xor(in_e ? re : 0), bke, ble, true, e_sep, bke, oke, a, b, c);
                const uint32_t sh = (uint32_t)x1 & 31;
                X = win64(a.x, b.x, c.x, sh); Y = win64(a.y, b.y, c.y, sh); Z = win64(a.z, b.z, c.z, sh);
                need_e = false;
            }
            if (need_b && (!b_sep || in_b)) {   // '$' positions of BWT[x0 ...] and the '$' before x0
                uint4 a, b, c;
                grp_window(img_b, t, pool + (in_b ? rb : 0) * FMD_BLK_U4, fmd_pool_xor(in_b ? rb : 0), bkb, blb, true, b_sep, bkb, okb, a, b, c);
                D = win64(~(a.x | a.y | a.z), ~(b.x | b.y | b.z), ~(c.x | c.y | c.z), x0lo & 31) & ((1ull << sz) - 1);
                r0 = fmd_block_rank1(img_b, t, okb + 1, 0, bkb);
                need_b = false; need_d = false;
            }
            if (base + GRP_POOL >= n_spill) break;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // the windows are read before the next pass lands in the pool
        }

        // Absolute ranks are needed only for the coordinates that survive: x[1] of the kept child
        // (one rank of one symbol) and x[1] of a neighbour (one of '$').
        uint32_t s[6] = {0, 0, 0, 0, 0, 0};   // child sizes (<= 63)
        bool is_nei = false;
        uint32_t cm = 0;                    // children c = 1..4 that survive the sentinel test
        uint32_t dm = 0;                    // the same before any masking, N included (check_left's view, lfork)
        if (live) {
            const uint64_t m = (1ull << sz) - 1;
            const uint64_t lo = ~Z & m, hi = Z & ~Y & m;
            s[0] = __popcll(lo & ~Y & ~X); s[1] = __popcll(lo & ~Y & X); s[2] = __popcll(lo & Y & ~X); s[3] = __popcll(lo & Y & X);
            s[4] = __popcll(hi & ~X); s[5] = sz - (s[0] + s[1] + s[2] + s[3] + s[4]);
            // children of the x[0] range laid out in the order $,T,G,C,A (exact.c:81-86);
            // sub-range [o_k, o_k+1) of the window = 2^o_k+1 - 2^o_k (all offsets <= 63)
            const uint64_t B1 = 1ull << s[0], B2 = B1 << s[4], B3 = B2 << s[3], B4 = B3 << s[2], B5 = B4 << s[1];
            const uint32_t e0sz = (uint32_t)__popcll(D & (B1 - 1));
            // unitig.c:111-122: a read ends here, bounded by sentinels on both sides, not contained
            is_nei = round > 0 && s[0] && s[0] == sz && e0sz == sz;
            if (s[4] && (D & (B2 - B1))) cm |= 1u << 4;
            if (s[3] && (D & (B3 - B2))) cm |= 1u << 3;
            if (s[2] && (D & (B4 - B3))) cm |= 1u << 2;
            if (s[1] && (D & (B5 - B4))) cm |= 1u << 1;
            dm = cm | ((s[5] && (D & ((B5 << s[5]) - B5))) ? 1u << 5 : 0u); // reads that start here and go on with A/C/G/T or N
        }

        // ---- the reference's sequential loop over the list, as prefix logic on group ballots
        const uint32_t alive_g = (uint32_t)(__ballot(live) >> gbase) & GM;
        const int ncur = __popc(alive_g);
        // ---- the QUIET round (as in k_ovl_nei_lane): every live candidate of every strand of the wave goes on with exactly ONE base, all of its occurrences, and a read
        // starts with that child -- nothing ends, nothing is a neighbour, nothing forks or dies: sizes, D, r0, order and categories stand, x[1] takes one LF step.
        // On reads with errors the long-lived strands are single reads marching to their ends: most of their rounds.  (quiet_on = 0: FMD_GRP_QUIET=0, the A/B switch)
        if (G <= 8 && quiet_on) {   // (the groups of 4 and 8: where the strands of reads with errors are; the larger instantiations have no registers to spare for it)
            const int qc = cm ? __ffs((int)cm) - 1 : 0;
            const uint32_t qs = qc == 1 ? s[1] : qc == 2 ? s[2] : qc == 3 ? s[3] : s[4];
            const bool lane_q = !live || (cm != 0 && (cm & (cm - 1)) == 0 && qs == sz);
            if (__ballot(!lane_q || (active && alive_g == 0)) == 0) {
                if (active && !(LF_GET(flags) & 0x10000u)) { // check_left's rounds (as below): which bases do the reads that start inside X go on with?
                    uint32_t u = 0;
#pragma unroll
                    for (int c = 1; c <= 4; ++c) if ((uint32_t)(__ballot((dm >> c) & 1) >> gbase) & GM) u |= 1u << c;
                    if (__popc(u) >= 2) LF_SET(flags, LF_GET(flags) | 0x18000u);
                    else LF_SET(flags, (uint32_t)(round + 1));   // (u != 0: every live lane has its child)
                }
                const uint64_t r = fmd_block_rank1(img_e, t, oke + 1, qc, bke);
                if (live) x1 = (qc == 1 ? ix.cnt[1] : qc == 2 ? ix.cnt[2] : qc == 3 ? ix.cnt[3] : ix.cnt[4]) + r;
                const uint32_t fk_g = (uint32_t)(__ballot(live && cat != 0) >> gbase) & GM;
                const int qsrc = alive_g ? gbase + __ffs((int)alive_g) - 1 : lane;
                const int first_c = (int)(uint32_t)__shfl((int)(uint32_t)qc, qsrc);   // base appended this round = base of the first child in push order (unitig.c:138-139)
                if (active) {
                    if (fk_g) flags |= FMD_OVLP_F_FORKED;
                    if (j == 0 && (uint32_t)(ori_l + round) < seq_stride) seq_out[gs * (size_t)seq_stride + ori_l + round] = (uint8_t)comp6(first_c);
                    ++round;
                    chk_down = true;
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // the images are read before the next gather lands in the slots
                continue;
            }
        }
        const uint32_t nei_g = (uint32_t)(__ballot(is_nei) >> gbase) & GM;
        const uint32_t head_g = (uint32_t)(__ballot(live && cat == j) >> gbase) & GM;   // first lane of each category
        const uint32_t in_cat_upto_j = (uint32_t)(bits_below(j + 1) & ~bits_below(cat));
        const bool new_nei = is_nei && (nei_g & in_cat_upto_j & (uint32_t)bits_below(j)) == 0; // first neighbour of its category
        const bool keep = live && (nei_g & in_cat_upto_j) == 0;                                 // not masked, not a neighbour
        const uint32_t newnei_g = (uint32_t)(__ballot(new_nei) >> gbase) & GM;
        if (!keep) cm = 0;
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
        // more children than the group has lanes: nothing of this round happens here (no neighbour is stored, no count moves) -- the
        // strand goes, as it stands BEFORE the round, to the list of the smallest group size that holds them (its kernel runs after
        // this one), or to the lane-per-strand kernel when there is none
        const bool too_many = active && n_new > G;
        const bool run = active && !too_many;
        if (run && !(LF_GET(flags) & 0x10000u)) { // check_left's rounds (include/fmd_hip.h, FMD_LFORK_*): which bases do the reads that start inside X go on with?
            uint32_t u = 0;
#pragma unroll
            for (int c = 1; c <= 5; ++c) if ((uint32_t)(__ballot((dm >> c) & 1) >> gbase) & GM) u |= 1u << c;
            if (__popc(u) >= 2) LF_SET(flags, LF_GET(flags) | 0x18000u);             // two bases in round `round`: D, closed
            else if (u == 0) LF_SET(flags, FMD_LFORK_ALL | 0x10000u);                 // every one of them ended
            else LF_SET(flags, (uint32_t)(round + 1) | ((u & 32u) ? 0x10000u : 0u));  // consistent; an N child is not followed any further
        }
        const uint32_t nei_k = n_nei + __popc(newnei_g & (uint32_t)bits_below(j)); // neighbours in list order (unitig.c:119-121)
        if (run && n_nei == 0 && newnei_g) { // info of nei[0] decides rbeg (unitig.c:157)
            const int src = gbase + __ffs((int)newnei_g) - 1;
            nei0_info = (uint32_t)ori_l - (uint32_t)__shfl((int)pos, src);
        }
        if (run) n_nei += __popc(newnei_g);
        bool forked_now = false;
        // x[1] of a child = cnt[c] + rank_c(x1 - 1): one rank of ONE symbol per surviving child (a second child only exists where the
        // read set forks); its D and r0 from the parent's: offsets of the children in the x[0] range, order $,T,G,C,A.  A new neighbour
        // (unitig.c:112, :119-121) needs one rank of '$': x[0] = r0, x[1] = cnt[0] + rank of '$' before x1.  Both kinds share one loop
        // so that the rank code runs once per wave step when nothing forks and no neighbour is found.
        const uint32_t o4 = s[0], o3 = o4 + s[4], o2 = o3 + s[3], o1 = o2 + s[2];
        {
            uint32_t todo = too_many ? 0u : cm;
            bool nei_todo = new_nei && !too_many;
            while (__ballot(todo != 0 || nei_todo)) {
                const int c = todo ? __ffs((int)todo) - 1 : 0;
                const uint64_t r = fmd_block_rank1(img_e, t, oke + 1, c, bke);
                if (todo) {
                    const int pc = c == 1 ? p1 : c == 2 ? p2 : c == 3 ? p3 : p4;
                    const uint32_t cmask = c == 1 ? c1 : c == 2 ? c2 : c == 3 ? c3 : c4;
                    const uint32_t oc = c == 1 ? o1 : c == 2 ? o2 : c == 3 ? o3 : o4;
                    const uint32_t sc = c == 1 ? s[1] : c == 2 ? s[2] : c == 3 ? s[3] : s[4];
                    const uint64_t nx1 = (c == 1 ? ix.cnt[1] : c == 2 ? ix.cnt[2] : c == 3 ? ix.cnt[3] : ix.cnt[4]) + r;
                    const uint64_t Dc = (D >> oc) & ((1ull << sc) - 1), r0c = r0 + (uint32_t)__popcll(D & ((1ull << oc) - 1));
                    const int d = pc + __popc(cmask & lt_m);
                    stage[2 * (gbase + d)] = make_uint4((uint32_t)nx1, (uint32_t)(nx1 >> 32) | sc << 8 | (uint32_t)pc << 16, (uint32_t)Dc, (uint32_t)(Dc >> 32));
                    stage[2 * (gbase + d) + 1] = make_uint4((uint32_t)r0c, (uint32_t)(r0c >> 32), pos, 0u);
                    forked_now |= pc != 0;
                    todo &= todo - 1;
                } else if (nei_todo) {
                    if (nei_k < max_nei) store_entry(nei_out + gs * (size_t)max_nei + nei_k, r0, ix.cnt[0] + r, (uint64_t)sz, (uint64_t)((uint32_t)ori_l - pos));
                    nei_todo = false;
                }
            }
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
            if (n_nei > max_nei) { // more neighbours than the caller has room for: the record says so (what k_ovl_nei would write after redoing
                if (j == 0) rec[gs].flags |= FMD_OVLP_F_OVERFLOW;   // the strand: n_nei, rbeg, ext_len stay as the walk left them), the caller re-runs it larger
                active = false; alive = false;
            } else if (too_many) {
                int k2 = FMD_GRP_CLASSES;
#pragma unroll
                for (int k = FMD_GRP_CLASSES - 1; k >= 0; --k) if (fmd_grp_size(k) > G && n_new <= fmd_grp_size(k)) k2 = k;
                if (k2 < FMD_GRP_CLASSES && fmd_resume_fits((uint32_t)round, n_nei, nei0_info, 0u) && (uint32_t)ncur <= cap) {
                    if (alive) fmd_resume_encode((uint4 *)(listB + sid * (size_t)cap + j), x1, sz, D, r0, pos, (uint32_t)round, n_nei, nei0_info, LF_GET(flags), (uint32_t)cat, flags & 0xffu);
                    if (j == 0) {
                        const uint32_t k = atomicAdd(cl.cnt + k2 * FMD_CLS_CNT_STRIDE, 1u);
                        cl.lst[k2][2 * (size_t)k] = sid; cl.lst[k2][2 * (size_t)k + 1] = (uint32_t)ncur | FMD_LIST_RESUME | (uint32_t)ori_l << 16;
                    }
                } else if (j == 0) { const uint32_t k = atomicAdd(slow_n, 1u); slow_list[k] = sid; }   // the lane-per-strand kernel starts it over
                active = false; alive = false;
            } else if (n_new > 0) { // next round (unitig.c:137-153)
                if (j == 0 && (uint32_t)(ori_l + round) < seq_stride) seq_out[gs * (size_t)seq_stride + ori_l + round] = (uint8_t)comp6(first_c);
                ++round;
                alive = j < n_new;
                if (alive) {
                    const uint4 a = stage[2 * (gbase + j)], b = stage[2 * (gbase + j) + 1];
                    x1 = (uint64_t)(a.y & 0xffu) << 32 | a.x; sz = (a.y >> 8) & 0xffu; cat = (int)(a.y >> 16);
                    D = (uint64_t)a.w << 32 | a.z; r0 = (uint64_t)b.y << 32 | b.x; pos = b.z;
                }
                chk_down = true;
            } else { // every path is closed (unitig.c:154-178)
                if (j == 0) {
                    fmd_ovlp_rec_t *o = rec + gs;
                    o->lfork = (uint16_t)(LF_GET(flags) & 0xffffu);
                    o->rbeg = n_nei ? ori_l - (int)nei0_info : -1;
                    o->ext_len = n_nei > 1 ? 0 : round;
                    o->n_nei = (int32_t)n_nei;
                    o->flags |= flags & 0xffu;
                    // a fake fork (contained reads, unitig.c:158-176): the record stands as fm6_get_nei leaves it BEFORE the fix-up; k_ovl_fix
                    // re-derives the appended bases from it (a lane walking ~60 dependent steps has no place in a group kernel)
                    if (n_nei == 1 && (flags & FMD_OVLP_F_FORKED)) { const uint32_t k = atomicAdd(slow_n + FMD_CLS_FIX_CNT, 1u); slow_list[fix_off + k] = sid; }
                }
                active = false; alive = false;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // staging reads done before the next round overwrites it
    }
}

// ---------------------------------------------------------------------- the unforked fast path
// fm6_get_nei for the strands it is simple on -- on read sets deep enough to assemble, nearly all of them: every candidate is in
// the narrow form (fmd_kernel_common.h: the walk left D and r0 with it), nothing has forked yet (one category) and, this round, the
// reads that contain any candidate go on with ONE base or end.  Same lane-per-candidate groups as k_ovl_nei_grp, a fraction of
// its work per round:
//   * the x[0] side is never fetched: a child's range is a sub-range of its parent's, so D shifts and r0 counts along;
//   * the x[1] ranges of a strand's candidates are NESTED (they are suffixes W_i of one string, longest first, and x[1] is the
//     interval of revcomp(W_i), of which revcomp(W_j), j > i, is a prefix): one window -- the widest live candidate's, one or two
//     blocks per strand and round through a 16-slot pool, ONE wave instruction -- serves every lane of the group, each lane
//     counting its own sub-range [d, d + size) of it; the child's x[1] is rank_c(X1 - 1) + the c's before d, one absolute rank
//     per group instead of per lane, and the neighbour's x[1] the same with '$';
//   * lanes never move: without categories the list order is the lane order, so there is no re-pack through LDS;
//   * the range masks are 32-bit words when the widest candidate has at most 31 occurrences (M = uint32_t), 64-bit otherwise.
// The moment a strand leaves that regime (a second base or an N among the reads of any candidate, even one no read starts with;
// a candidate in the wide form) it is appended to the general list of its class and k_ovl_nei_grp starts it over: nothing the fast
// path wrote for it survives (records are written at the close only; neighbours and appended bases are rewritten).
// uint4 index, in the two block images side by side (4 uint4 each), of the q-th 32-position word counted from the start of the first
__device__ __forceinline__ uint32_t fast_word(uint32_t q)
{
    return q < 3 ? q : q + 2;          // words 3, 4 = chunks 1, 2 of the next block (word 2 is in both)
}
template <typename M> struct FastW;
template <> struct FastW<uint32_t> {
    static constexpr uint32_t MAXW = 31;
    static __device__ __forceinline__ uint32_t below(uint32_t n) { return (1u << n) - 1u; }          // n <= 31
    static __device__ __forceinline__ uint32_t popc(uint32_t v) { return (uint32_t)__popc(v); }
    static __device__ __forceinline__ int top(uint32_t v) { return 31 - __clz((int)v); }
    // bit-planes of BWT[96 b + o1, .. + 32) from the images of blocks b, b + 1 (4 uint4 each, chunk 3 of a block = its counts)
    static __device__ __forceinline__ void window(const uint4 *img, uint32_t o1, uint32_t &X, uint32_t &Y, uint32_t &Z)
    {
        const uint32_t q = o1 >> 5, sh = o1 & 31;
        const uint4 a = img[fast_word(q)], b = img[fast_word(q + 1)];
        X = __builtin_amdgcn_alignbit(b.x, a.x, sh); Y = __builtin_amdgcn_alignbit(b.y, a.y, sh); Z = __builtin_amdgcn_alignbit(b.z, a.z, sh);
    }
};
template <> struct FastW<uint64_t> {
    static constexpr uint32_t MAXW = 63;
    static __device__ __forceinline__ uint64_t below(uint32_t n) { return (1ull << n) - 1ull; }       // n <= 63
    static __device__ __forceinline__ uint32_t popc(uint64_t v) { return (uint32_t)__popcll(v); }
    static __device__ __forceinline__ int top(uint64_t v) { return 63 - __clzll((long long)v); }
    static __device__ __forceinline__ void window(const uint4 *img, uint32_t o1, uint64_t &X, uint64_t &Y, uint64_t &Z)
    {
        const uint32_t q = o1 >> 5, sh = o1 & 31;
        const uint4 a = img[fast_word(q)], b = img[fast_word(q + 1)], c = img[fast_word(q + 2)];
        X = win64(a.x, b.x, c.x, sh); Y = win64(a.y, b.y, c.y, sh); Z = win64(a.z, b.z, c.z, sh);
    }
};

#ifndef FMD_FAST_LB
#define FMD_FAST_LB 5      // waves per SIMD the register budget is cut for.  6 makes some instantiations spill: the build of the round-2 incident (tools/scratch_incident.py)
#endif
template <int G, typename M, bool DYN>
__global__ __launch_bounds__(64, FMD_FAST_LB) void k_ovl_nei_fast(FmdIndexView ix, const uint32_t *__restrict__ list, const uint32_t *__restrict__ list_n,
                                                     uint32_t cap, const fmd_intv_t *__restrict__ listA, fmd_intv_t *__restrict__ listB, fmd_ovlp_rec_t *__restrict__ rec,
                                                     fmd_intv_t *__restrict__ nei_out, uint32_t max_nei, uint8_t *__restrict__ seq_out,
                                                     uint32_t seq_stride, uint32_t *__restrict__ gen_list, uint32_t *__restrict__ gen_n,
                                                     uint32_t *__restrict__ bail_n, uint32_t *__restrict__ slow_list, uint32_t *__restrict__ slow_n,
                                                     const uint32_t *__restrict__ gidx, uint32_t *__restrict__ deal)
{
    using W = FastW<M>;
    constexpr int S = 64 / G;
    // pool slots 2g, 2g+1 = the block(s) under group g's window, images side by side and unswizzled (the lanes of a group read ONE
    // address: a broadcast); + the two slots the idle lanes past S * G would read
    // (G = 4: 16 strands per wave, 32 slots, two gather instructions per step -- the second one carries the groups 8..15)
    constexpr int NI = (2 * S + FMD_BLK_PER_INST - 1) / FMD_BLK_PER_INST;
    __shared__ uint4 pool[(NI * FMD_BLK_PER_INST + 2) * FMD_BLK_U4];
    static_assert(NI <= 2 && S <= FMD_FAST_CHUNK, "a step hands on at most one chunk of strands");
    constexpr uint32_t GM = G == 32 ? 0xffffffffu : (1u << G) - 1;
    const int lane = fmd_lane(), g = lane / G, j = lane % G, gbase = g * G;
    const uint32_t N = *list_n;
    const uint32_t n_groups = gridDim.x * S;
    uint32_t idx = !DYN && g < S ? blockIdx.x * S + g : 0xffffffffu;
    FmdDeal tk;                                                   // DYN: list positions dealt out as the groups ask (see k_ovl_nei_grp)
    bool dry = !DYN;
    if (N == 0) return;
    if (DYN) { fmd_deal_init<G>(tk, N, dry); if (dry) return; }
    // this lane's part in gather instruction r: chunk (lane & 3) of pool slot 16 r + (lane >> 2), which belongs to group slot / 2
    const uint4 *img = pool + 2 * FMD_BLK_U4 * g;

    // group-uniform strand state
    bool active = false;
    uint32_t sid = 0, gs = 0, meta = 0, n_nei = 0, lf = 0, nei0 = 0, szw = 1, round = 0;   // sid: slot in the batch; gs: row in rec / nei_out / seq_out
    uint64_t X1 = 1;                                              // start of the group's window = x[1] of the widest live candidate
    // the lane's candidate: its range is [X1 + d, X1 + d + sz) on the x[1] side
    bool alive = false;
    uint32_t d = 0, sz = 0, pos = 0;
    M D = 0;
    uint64_t r0 = 0;
    int pf = 0;
    uint32_t d_sid = 0, d_meta = 0, d_gs = 0;
    uint4 pa = make_uint4(0, 0, 0, 0), pb = make_uint4(0, 0, 0, 0);
    // Strands handed on to the general kernel go to slots of its list that the wave reserves FMD_FAST_CHUNK at a time (one atomic on
    // the list counter per chunk; one per strand serialises: 5*10^6 atomics on one address cost 45 ms on reads with 1 % errors);
    // a chunk is filled with hole markers when it is reserved, k_ovl_nei_grp skips what stays a hole -- at most FMD_FAST_CHUNK - 1
    // slots per wave and launch, which is what FMD_FAST_RESERVE adds to the capacity of a general list.
    uint32_t res_cur = 0, res_end = 0, n_handed = 0;              // wave-uniform
    bool hand_on = false;                                         // this group's strand leaves for the general kernel (set at j == 0 too)
    bool resume = false;                                          // ... in the middle: with its state (FMD_LIST_RESUME), not from round 0

    for (;;) {
        {   // ---- hand-overs of the previous step / of the admission below
            const uint64_t hm = __ballot(hand_on && j == 0);
            if (hm) {
                // a strand that leaves in the middle: its live candidates, packed in list order, into its row of listB
                uint32_t meta_out = meta;
                if (hand_on && resume) {
                    const uint32_t ag = (uint32_t)(__ballot(alive) >> gbase) & GM, mp = (uint32_t)__popc(ag);
                    if (mp <= cap && fmd_resume_fits(round, n_nei, nei0, 0u)) {   // (pos <= the strand's length < 65535: k_ovl_classify)
                        if (alive) fmd_resume_encode((uint4 *)(listB + sid * (size_t)cap + __popc(ag & ((1u << j) - 1u))), X1 + d, sz, (uint64_t)D, r0, pos, round, n_nei, nei0, lf);
                        meta_out = mp | FMD_LIST_RESUME | (meta & 0xffff0000u);
                    }
                    alive = false; resume = false;
                }
                const uint32_t n = (uint32_t)__popcll(hm), room = res_end - res_cur;   // n <= S <= FMD_FAST_CHUNK
                uint32_t base = 0;
                if (room < n) {   // the first `room` of them finish the old chunk, the rest start a new one: only a wave's LAST chunk keeps holes
                    if (lane == 0) { base = atomicAdd(gen_n, (uint32_t)FMD_FAST_CHUNK); atomicAdd(bail_n, n_handed); }   // (the count: diagnostics, FMD_OVLP_STATS)
                    n_handed = 0;
                    base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                    if (lane < FMD_FAST_CHUNK) { gen_list[2 * (size_t)(base + lane)] = FMD_LIST_HOLE; gen_list[2 * (size_t)(base + lane) + 1] = 0; }
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the markers land before the entries that replace them
                }
                if (hand_on && j == 0) {
                    const uint32_t q = (uint32_t)fmd_below(hm), k = q < room ? res_cur + q : base + (q - room);
                    gen_list[2 * (size_t)k] = sid; gen_list[2 * (size_t)k + 1] = meta_out;
                }
                if (room < n) { res_cur = base + (n - room); res_end = base + FMD_FAST_CHUNK; }
                else res_cur += n;
                n_handed += n;
                hand_on = false;
            }
        }
        // ---- admission
        if (!active && pf == 2) {
            const uint32_t m = d_meta & 0xffff;
            sid = d_sid; gs = d_gs; meta = d_meta; round = 0; n_nei = 0; lf = 0; nei0 = 0;
            alive = (uint32_t)j < m;
            const FmdCand cd = cand_decode(pa, pb);
            const int wl = gbase + (int)m - 1;                   // the widest candidate is the last
            const uint32_t wlo = (uint32_t)__shfl((int)(uint32_t)cd.x1, wl), whi = (uint32_t)__shfl((int)(uint32_t)(cd.x1 >> 32), wl);
            X1 = (uint64_t)whi << 32 | wlo;
            szw = (uint32_t)__shfl((int)(uint32_t)cd.sz, wl);
            sz = (uint32_t)cd.sz; D = (M)cd.D; r0 = cd.r0; pos = (meta >> 16) - cd.depth;
            d = (uint32_t)(cd.x1 - X1);
            // every candidate in the narrow form and inside the widest one's range (nesting: see above)
            const bool bad = alive && (!cd.narrow || cd.x1 < X1 || cd.x1 - X1 + cd.sz > szw || szw > W::MAXW || cd.sz == 0);
            if ((uint32_t)(__ballot(bad) >> gbase) & GM) { hand_on = true; alive = false; }   // (sid and meta stay until the hand-over above)
            else active = true;
            pf = 0; if (!DYN) idx += n_groups;
        }
        // ---- prefetch pipeline (loads complete under the window gather below)
        if (DYN) idx = fmd_deal_next<G>(tk, deal, N, pf == 0 && g < S, dry);
        if (pf == 1) {
            const uint32_t m = d_meta & 0xffff;
            if ((uint32_t)j < m) { const uint4 *q = (const uint4 *)(listA + d_sid * (size_t)cap + (cap - m) + j); pa = q[0]; pb = q[1]; }
            d_gs = gidx ? gidx[d_sid] : d_sid;
            pf = 2;
        } else if (pf == 0 && idx < N) {
            d_sid = list[2 * (size_t)idx]; d_meta = list[2 * (size_t)idx + 1];
            pf = 1;
        }
        const uint64_t act_m = __ballot(active);
        if (act_m == 0) {
            if (__ballot(pf != 0 || (!DYN && idx < N) || hand_on) == 0 && dry) { if (lane == 0 && n_handed) atomicAdd(bail_n, n_handed); break; }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            continue;
        }

#ifdef GRP_STATS
        { const uint32_t nl = (uint32_t)__popcll(__ballot(active && alive)); if (lane == 0) { atomicAdd(bail_n + 2, 1u); atomicAdd(bail_n + 3, nl); atomicAdd(bail_n + 4, (uint32_t)__popcll(act_m)); } }
#endif
        // ---- the window of every resident strand: BWT[X1, X1 + szw), its block(s) into pool slots 2g (2g + 1)
        uint32_t bke, oke;
        fmd_split(X1 - 1, bke, oke);                              // x[1] >= cnt[1] > 0 for base strings
        const uint32_t o1 = oke + 1;                              // offset of X1 in block bke (96: the first position of the next one)
        const uint64_t sep_m = __ballot(active && o1 + szw > FMD_BLK_SYMS);
#if FMD_COUNT_LINES
        { uint64_t heads = 0; for (int q = 0; q < S; ++q) heads |= 1ull << (q * G); fmd_count_lines(ix, __popcll(act_m & heads) + __popcll(sep_m & heads)); }
#endif
#pragma unroll
        for (int r = 0; r < NI; ++r) {
            const int f_slot = r * FMD_BLK_PER_INST + (lane >> FMD_GRP_SHIFT), f_grp = f_slot >> 1, f_src = f_grp < S ? f_grp * G : 0;
            const uint32_t sb = (uint32_t)__shfl((int)bke, f_src);
            if (f_grp < S && ((act_m >> f_src) & 1) && (!(f_slot & 1) || ((sep_m >> f_src) & 1))) {
                const uint4 *from = ix.blocks + (size_t)(sb + (uint32_t)(f_slot & 1)) * FMD_BLK_U4 + (lane & FMD_GRP_MASK);
                __builtin_amdgcn_global_load_lds((fmd_glb_void *)from, (fmd_lds_void *)(pool + r * FMD_BLK_PER_INST * FMD_BLK_U4), 16, 0, FMD_GLDS_AUX);
            }
        }
        fmd_fetch_wait();
        M X, Y, Z;
        W::window(img, o1, X, Y, Z);   // (words of a block that was not fetched lie past the range and are masked out)
        // The base the strand goes on with.  Usually every read of the window that does not end here shows the same one; reads that
        // show another (sequencing errors, repeats) matter only if one of them STARTS inside a candidate (a surviving child,
        // unitig.c:126-134): the wave takes the longer way below only in steps where some strand sees a second base at all.
        const bool live = active && alive;
        const M mw = W::below(szw), mine = W::below(sz) << d;
        const M any = (X | Y | Z) & mw, xa = X & any, ya = Y & any, za = Z & any;
        int cs = (xa ? 1 : 0) | (ya ? 2 : 0) | (za ? 4 : 0);
        M Cw = any;                                               // the positions of cs in the window
        uint32_t nc = W::popc(any & mine), coff = sz - nc;        // reads of my range that go on with cs; children before it in x[0] order
        const uint32_t nS = W::popc(~(X | Y | Z) & mw & mine);   // reads of my range that end here
        bool bail = false;
        if (__ballot(active && ((xa && xa != any) || (ya && ya != any) || (za && za != any) || cs > 4))) {
#ifdef GRP_STATS
            if (lane == 0) atomicAdd(bail_n + 5, 1u);
#endif
            // per base: how many reads of my range show it (order of the x[0] side: $ T G C A N), and does one start with it?
            const M m1 = X & ~Y & ~Z & mw, m2 = ~X & Y & ~Z & mw, m3 = X & Y & ~Z & mw, m4 = ~X & ~Y & Z & mw, m5 = X & ~Y & Z & mw;
            const uint32_t s1 = W::popc(m1 & mine), s2 = W::popc(m2 & mine), s3 = W::popc(m3 & mine), s4 = W::popc(m4 & mine), s5 = W::popc(m5 & mine);
            const uint32_t o4 = nS, o3 = o4 + s4, o2 = o3 + s3, o1_ = o2 + s2, o5 = o1_ + s1;
            uint32_t cm = 0;
            if (live) {
                if ((D >> o1_) & W::below(s1)) cm |= 2u;
                if ((D >> o2) & W::below(s2)) cm |= 4u;
                if ((D >> o3) & W::below(s3)) cm |= 8u;
                if ((D >> o4) & W::below(s4)) cm |= 16u;
                if ((D >> o5) & W::below(s5)) cm |= 32u;
            }
            uint32_t u = 0;
#pragma unroll
            for (int c = 1; c <= 5; ++c) if ((uint32_t)(__ballot((cm >> c) & 1) >> gbase) & GM) u |= 1u << c;
            bail = active && (__popc(u) >= 2 || (u & 32u));     // a fork (or an N to follow): the general kernel's business
            cs = u ? __ffs((int)u) - 1 : 0;
            if (cs > 4) cs = 0;
            Cw = cs == 1 ? m1 : cs == 2 ? m2 : cs == 3 ? m3 : cs == 4 ? m4 : (M)0;
            nc = cs == 1 ? s1 : cs == 2 ? s2 : cs == 3 ? s3 : cs == 4 ? s4 : 0u;
            coff = cs == 1 ? o1_ : cs == 2 ? o2 : cs == 3 ? o3 : o4;
        }
        // absolute ranks at X1 - 1: of that base (children) and of '$' (x[1] of a neighbour)
        uint64_t Rz;
        const uint64_t Rc = fmd_block_rank1z(img, 0, o1, cs, bke, Rz);

        // ---- this lane's candidate
        const bool is_nei = live && round > 0 && nS == sz && D == W::below(sz);   // unitig.c:111-122
        const M Dc = (D >> coff) & W::below(nc);                                   // reads that start with the child string (unitig.c:129)
        const bool has_child = live && Dc != 0;
        const uint32_t nei_g = (uint32_t)(__ballot(is_nei) >> gbase) & GM, dm_g = (uint32_t)(__ballot(has_child) >> gbase) & GM;
        const int f = nei_g ? __ffs((int)nei_g) - 1 : 64;       // the first neighbour masks the rest of the (only) category
        const bool alive2 = has_child && j < f;
        const uint32_t child_g = (uint32_t)(__ballot(alive2) >> gbase) & GM;

        if (active) {
            if (bail) { hand_on = true; resume = true; active = false; }   // (the candidates stay as they are until the hand-over at the top of the next step)
            else {
                const uint32_t ori_l = meta >> 16;
                if (!(lf & 0x10000u)) lf = dm_g ? round + 1 : (FMD_LFORK_ALL | 0x10000u);   // check_left's rounds (FMD_LFORK_*)
                if (nei_g) {
                    if (n_nei == 0) nei0 = ori_l - (uint32_t)__shfl((int)pos, gbase + f);   // info of nei[0] decides rbeg (unitig.c:157)
                    if (is_nei && j == f && n_nei < max_nei)
                        store_entry(nei_out + gs * (size_t)max_nei + n_nei, r0, ix.cnt[0] + Rz + W::popc(~(X | Y | Z) & W::below(d)), sz, (uint64_t)(ori_l - pos));
                    ++n_nei;
                }
                if (n_nei > max_nei) { // more neighbours than the caller has room for: flagged (as k_ovl_nei would after redoing the strand), re-run larger by the caller
                    if (j == 0) rec[gs].flags |= FMD_OVLP_F_OVERFLOW;
                    active = false; alive = false;
                } else if (child_g) { // next round (unitig.c:137-153)
                    if (j == 0 && ori_l + round < seq_stride) seq_out[gs * (size_t)seq_stride + ori_l + round] = (uint8_t)(5 - cs);   // comp6, cs in 1..4
                    ++round;
                    const int wl = gbase + 31 - __clz((int)child_g);                 // the widest child
                    const uint32_t before = W::popc(Cw & W::below(d));                // cs's of the window before my range
                    const uint32_t wb = (uint32_t)__shfl((int)before, wl);
                    szw = (uint32_t)__shfl((int)nc, wl);
                    X1 = (cs == 1 ? ix.cnt[1] : cs == 2 ? ix.cnt[2] : cs == 3 ? ix.cnt[3] : ix.cnt[4]) + Rc + wb;
                    r0 += W::popc(D & W::below(coff));
                    D = Dc; sz = nc; d = alive2 ? before - wb : 0u; alive = alive2;
                } else { // every path is closed (unitig.c:154-178); nothing forked, so no fix-up
                    if (j == 0) {
                        fmd_ovlp_rec_t *o = rec + gs;
                        o->lfork = (uint16_t)(lf & 0xffffu);
                        o->rbeg = n_nei ? (int)(ori_l - nei0) : -1;
                        o->ext_len = n_nei > 1 ? 0 : (int)round;
                        o->n_nei = (int32_t)n_nei;
                    }
                    active = false; alive = false;
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // the images are read before the next gather lands in the pool
    }
}

static inline int grp_cap(int resident, int cap) { return cap > 0 && cap < resident ? cap : resident; }
template <int G, typename M, bool DYN>
static int fast_blocks_per_cu(void)
{
    static int cached = 0;
    if (!cached) cached = fmd_resident_per_cu(k_ovl_nei_fast<G, M, DYN>, sizeof(uint4) * ((2 * (64 / G) + FMD_BLK_PER_INST - 1) / FMD_BLK_PER_INST * FMD_BLK_PER_INST + 2) * FMD_BLK_U4, 32, "k_ovl_nei_fast");
    return cached;
}
int fmd_nei_fast_available(void) { return 1; }
static inline int fast_grid(int waves) { return waves < FMD_FAST_MAX_WAVES ? waves : FMD_FAST_MAX_WAVES; }   // (the hand-over lists have room for this many waves' holes)
void fmd_launch_nei_fast(int cls, int wide, int n_cu, int per_cu_cap, hipStream_t st, const FmdIndexView &ix, const uint32_t *list, const uint32_t *list_n, uint32_t cap,
                         const fmd_intv_t *listA, fmd_intv_t *listB, fmd_ovlp_rec_t *rec, fmd_intv_t *nei_out, uint32_t max_nei, uint8_t *seq_out,
                         uint32_t seq_stride, uint32_t *gen_list, uint32_t *gen_n, uint32_t *bail_n, uint32_t *slow_list, uint32_t *slow_n, const uint32_t *gidx)
{
    const char *e = getenv("FMD_FAST_WAVES"); // A/B knob: resident waves per CU
    if (e && atoi(e) > 0 && (per_cu_cap <= 0 || atoi(e) < per_cu_cap)) per_cu_cap = atoi(e);
    const bool dyn = nei_dyn() != 0;
    uint32_t *deal = (uint32_t *)list_n + FMD_DEAL_WORD;
#define FAST_LAUNCH_(K, M, DY) k_ovl_nei_fast<fmd_grp_size(K), M, DY><<<fast_grid(n_cu * grp_cap(fast_blocks_per_cu<fmd_grp_size(K), M, DY>(), per_cu_cap)), 64, 0, st>>>(ix, list, list_n, cap, listA, listB, rec, nei_out, max_nei, seq_out, seq_stride, gen_list, gen_n, bail_n, slow_list, slow_n, gidx, deal)
#define FAST_LAUNCH(K, M) do { if (dyn) FAST_LAUNCH_(K, M, true); else FAST_LAUNCH_(K, M, false); } while (0)
#define FAST_LAUNCH2(K) do { if (wide) FAST_LAUNCH(K, uint64_t); else FAST_LAUNCH(K, uint32_t); } while (0)
    switch (cls) {
    case 0: FAST_LAUNCH2(0); break;
    case 1: FAST_LAUNCH2(1); break;
    case 2: FAST_LAUNCH2(2); break;
    case 3: FAST_LAUNCH2(3); break;
    case 4: FAST_LAUNCH2(4); break;
    default: FAST_LAUNCH2(5); break;
    }
#undef FAST_LAUNCH2
#undef FAST_LAUNCH
#undef FAST_LAUNCH_
}

// launcher used by fmd_ovlp.hip.  The strands of a work list are dealt to the groups round-robin, so
// the grid must be exactly the resident set: a block that has to wait for a slot starts when the others
// are done and then works alone through a full share (11 blocks per CU computed from 160 KiB / 14.25 KiB
// ran 17 % slower than 10: LDS is handed out in 1280-byte granules).  Ask the runtime.
template <int G, bool DYN>
static int grp_blocks_per_cu(void)
{
    static int cached = 0;
    if (!cached) {
        int nb = fmd_resident_per_cu(k_ovl_nei_grp<G, DYN>, GRP_LDS_U4 * 16, 16, "k_ovl_nei_grp");
        const char *e = getenv("FMD_GRP_WAVES"); // A/B knob: fewer resident waves per CU
        if (e && atoi(e) > 0 && atoi(e) < nb) nb = atoi(e);
        cached = nb;
    }
    return cached;
}
void fmd_launch_nei_grp(int cls, int n_cu, int per_cu_cap, hipStream_t st, const FmdIndexView &ix, const uint32_t *list, const uint32_t *list_n, uint32_t cap,
                        const fmd_intv_t *listA, fmd_intv_t *listB, const FmdOvlClasses &cl, fmd_ovlp_rec_t *rec, fmd_intv_t *nei_out, uint32_t max_nei, uint8_t *seq_out,
                        uint32_t seq_stride, uint32_t *slow_list, uint32_t *slow_n, const uint32_t *gidx, size_t fix_off, uint32_t down_cap, int second_pass)
{
    const uint32_t n_max = second_pass ? down_cap / FMD_FAST_CHUNK * FMD_FAST_CHUNK : 0u;
    const char *eq = getenv("FMD_GRP_QUIET");
    const int quiet_on = !(eq && atoi(eq) == 0);
    const bool dyn = nei_dyn() != 0;
    uint32_t *deal = (uint32_t *)list_n + FMD_DEAL_WORD;
#define GRP_LAUNCH_(K, DY) k_ovl_nei_grp<fmd_grp_size(K), DY><<<n_cu * grp_cap(grp_blocks_per_cu<fmd_grp_size(K), DY>(), per_cu_cap), 64, 0, st>>>(ix, list, list_n, cap, listA, listB, cl, rec, nei_out, max_nei, seq_out, seq_stride, slow_list, slow_n, gidx, fix_off, deal, down_cap, n_max, quiet_on)
#define GRP_LAUNCH(K) do { if (dyn) GRP_LAUNCH_(K, true); else GRP_LAUNCH_(K, false); } while (0)
    switch (cls) {
    case 0: GRP_LAUNCH(0); break;
    case 1: GRP_LAUNCH(1); break;
    case 2: GRP_LAUNCH(2); break;
    case 3: GRP_LAUNCH(3); break;
    case 4: GRP_LAUNCH(4); break;
    default: GRP_LAUNCH(5); break;
    }
#undef GRP_LAUNCH
#undef GRP_LAUNCH_
}
void fmd_launch_classify(hipStream_t st, size_t n, const fmd_ovlp_rec_t *rec, const fmd_intv_t *listA, uint32_t cap, FmdOvlClasses cl, int use_fast, const uint32_t *gidx)
{
    const char *e = getenv("FMD_GRP4");   // A/B knob: FMD_GRP4=0 = no groups of 4 (round 3's classes)
    k_ovl_classify<<<(unsigned)((n + CLS_THREADS - 1) / CLS_THREADS), CLS_THREADS, 0, st>>>(n, rec, listA, cap, cl, use_fast, gidx, e && atoi(e) == 0 ? 1 : 0);
}
