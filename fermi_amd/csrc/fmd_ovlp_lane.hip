// fmd_ovlp_lane.hip -- the unforked path of fm6_get_nei (unitig.c:93-179) with ONE LANE PER STRAND.
//
// k_ovl_nei_fast (fmd_ovlp_grp.hip) gives a strand a group of G lanes, one per candidate interval, and what it found out about the simple case
// -- every candidate in the narrow form (D and r0 travel with it: the x[0] side of the index is never read), the x[1] ranges of a strand's
// candidates NESTED in the widest one's (one window of at most 63 positions serves them all), nothing forked (lanes never move) -- leaves each of
// those lanes a handful of popcounts per round.  What a round costs there is not those: it is the round trip of ONE gather per strand and round
// with 64 / G strands per wave and 20 waves per CU in flight (80 strands per CU at G = 16: 4.3 dependent gathers of ~2.3 us per strand,
// 70 of the headline's 254 ms), and the window, the ranks and the ballots done once per STRAND by sixty-four lanes.
//
// Here a lane owns a strand: its candidates are 16-byte entries of the lane's own column of LDS (D, r0, packed offset / size / start; entry j of lane q at
// [j][q]: a wave's ds_read_b128 of one j is conflict-free), the round is a loop over them (as registers -- unrolled over G -- they spilled: 515 VGPRs
// at G = 16), the window comes from the lane's own block image (the wave engine's cooperative gather: 64 strands' blocks per wave instruction), and
// 64 strands per wave x 8-13 waves per CU (LDS: G + 4 KiB per wave) are in flight instead of 80.  Classes of up to 21 candidates: more than 8 are admitted in two parts, the widest half first (64-88 raw
// words in registers beside the round's state instead of 128-168); the class of 32 keeps the group form.  Same results by
// construction: the arithmetic of a round is k_ovl_nei_fast's, statement by statement; a strand that leaves the simple regime is handed to
// k_ovl_nei_grp in the same FMD_LIST_RESUME form.  FMD_NEI_LANE=0 runs the group form instead (the A/B switch, and the tests' second opinion).
#include "fmd_kernel_common.h"

#define LANE_CHUNK FMD_LANE_CHUNK          // list slots a wave reserves at a time for the strands it hands on (64: every lane may hand on in one step)

template <typename M> struct LaneW;
template <> struct LaneW<uint32_t> {
    static constexpr uint32_t MAXW = 31;
    static __device__ __forceinline__ uint32_t below(uint32_t n) { return (1u << n) - 1u; }           // n <= 31
    static __device__ __forceinline__ uint32_t popc(uint32_t v) { return (uint32_t)__popc(v); }
    static __device__ __forceinline__ int top(uint32_t v) { return 31 - __clz((int)v); }
};
template <> struct LaneW<uint64_t> {
    static constexpr uint32_t MAXW = 63;
    static __device__ __forceinline__ uint64_t below(uint32_t n) { return (1ull << n) - 1ull; }        // n <= 63
    static __device__ __forceinline__ uint32_t popc(uint64_t v) { return (uint32_t)__popcll(v); }
    static __device__ __forceinline__ int top(uint64_t v) { return 63 - __clzll((long long)v); }
};

// a candidate in three (four: 64-bit masks) registers: D, r0 lo, r0 bits 32..39 | d << 8 | size << 14 | start << 20
#define LC_PACK(r0, d, sz, pos) ((uint32_t)((r0) >> 32) & 0xffu) | (uint32_t)(d) << 8 | (uint32_t)(sz) << 14 | (uint32_t)(pos) << 20
#define LC_D(p) (((p) >> 8) & 63u)
#define LC_SZ(p) (((p) >> 14) & 63u)
#define LC_POS(p) ((p) >> 20)
#define LC_R0(p, lo) ((uint64_t)((p) & 0xffu) << 32 | (lo))
#define LANE_MAX_LEN 4095u                  // the start of a candidate takes 12 bits (longer sequences: the group kernels)


// the appended bases of rounds [from, to) of a strand (2 bits each in `eb`, (nt6 code - 1)) to its sequence row: whole aligned words where there are any, bytes at the ends
__device__ __forceinline__ void lane_flush_bases(uint8_t *seq_out, uint32_t seq_stride, uint32_t gs, uint32_t ori_l, uint32_t from, uint32_t to, uint64_t eb)
{
    uint8_t *row = seq_out + gs * (size_t)seq_stride;
    uint32_t p = ori_l + from, e = ori_l + to;
    if (e > seq_stride) e = seq_stride;
    while (p < e && (p & 3u)) { row[p] = (uint8_t)((eb & 3u) + 1u); eb >>= 2; ++p; }
    while (p + 4 <= e) {
        const uint32_t b = (uint32_t)eb & 0xffu;
        *(uint32_t *)(row + p) = ((b & 3u) | (b & 0xcu) << 6 | (b & 0x30u) << 12 | (b & 0xc0u) << 18) + 0x01010101u;
        eb >>= 8; p += 4;
    }
    while (p < e) { row[p] = (uint8_t)((eb & 3u) + 1u); eb >>= 2; ++p; }
}

#ifndef FMD_LANE_LB
#define FMD_LANE_LB 2                       // (LDS decides the residency: G + 4 KiB per wave)
#endif

template <int G, typename M>
__global__ __launch_bounds__(64, FMD_LANE_LB) void k_ovl_nei_lane(FmdIndexView ix, const uint32_t *__restrict__ list, const uint32_t *__restrict__ list_n,
                                                                uint32_t cap, const fmd_intv_t *__restrict__ listA, fmd_intv_t *__restrict__ listB, fmd_ovlp_rec_t *__restrict__ rec,
                                                                fmd_intv_t *__restrict__ nei_out, uint32_t max_nei, uint8_t *__restrict__ seq_out, uint32_t seq_stride,
                                                                uint32_t *__restrict__ gen_list, uint32_t *__restrict__ gen_n, uint32_t *__restrict__ bail_n,
                                                                const uint32_t *__restrict__ gidx, uint32_t *__restrict__ queue, uint32_t tk_chunk, uint32_t slow_min, uint32_t adm_min)
{
    using W = LaneW<M>;
    constexpr bool WIDE = sizeof(M) == 8;
    __shared__ uint4 fmd_lds[WIDE ? FMD_WAVE_LDS_U4 : FMD_SLOT_U4];   // (32-bit masks: the window never leaves the block of X1 - 1, one dense slot is all)
    uint4 *lds = fmd_lds;
    const int lane = fmd_lane();
    const uint32_t N = *list_n;
    if (N == 0) return;
    FmdTickets tk;
    fmd_tickets_init(tk, queue, tk_chunk, N);

    // the lane's strand
    bool active = false;
    uint32_t sid = 0, gs = 0, meta = 0, n_nei = 0, lf = 0, nei0 = 0, szw = 1, round = 0, am = 0;   // am: bit j = candidate j is alive
    uint64_t X1 = 1;
    uint64_t eb = 0;                                       // the bases appended since round eb0, (5 - cs - 1) in 2 bits each, oldest lowest: they leave in words, not one byte a round
    uint32_t eb0 = 0;
    __shared__ uint4 cand[G * 64];                         // candidate j of this lane: cand[j * 64 + lane] = { D lo, D hi, r0 lo, r0 bits 32..39 | d << 8 | size << 14 | start << 20 }
    uint4 *mc = cand + lane;
#define CD(e) (WIDE ? (M)((uint64_t)(e).y << 32 | (e).x) : (M)(e).x)
    uint32_t m_lane = 0;                                   // candidates the strand came with (alive ones: am)
    uint32_t res_cur = 0, res_end = 0, n_handed = 0;       // wave-uniform: list slots reserved for hand-overs
    int hand = 0;                                          // 1 = hand this strand on from round 0, 2 = with its state (FMD_LIST_RESUME)
    bool slow = false;                                     // the strand's round waits for the wave's next pass through the full round code, with what it saw of its window:
    M sX = 0, sY = 0, sZ = 0;
    uint64_t sR0 = 0, sR1 = 0, sR2 = 0, sR3 = 0, sR4 = 0;    // ranks at X1 - 1 of '$', A, C, G, T (one base in the window: that base's in all four)
    int scs = 0;                                           // the base | 8 = more than one base among the reads of the window
    int st = 0;                                            // admission: 0 none in progress, 2 = (sid, meta) are known, the candidates [m / 2, m) are fetched next, 3 = then the others
    bool adm_bad = false;
    constexpr int HALVES = G > 8 ? 2 : 1, RAWN = (G + HALVES - 1) / HALVES;
    bool drained = false;

    for (;;) {
        {   // ---- hand-overs of the previous step / of the admission below
            const uint64_t hm = __ballot(hand != 0);
            if (hm) {
                uint32_t meta_out = meta;
                if (hand == 2) {
                    const uint32_t mp = (uint32_t)__popc(am);
                    if (mp <= cap && fmd_resume_fits(round, n_nei, nei0, 0u)) {
                        for (uint32_t j = 0; j < m_lane; ++j)
                            if ((am >> j) & 1u) {
                                const uint4 e = mc[j * 64];
                                fmd_resume_encode((uint4 *)(listB + sid * (size_t)cap + __popc(am & ((1u << j) - 1u))), X1 + LC_D(e.w), LC_SZ(e.w), (uint64_t)CD(e),
                                                  LC_R0(e.w, e.z), LC_POS(e.w), round, n_nei, nei0, lf);
                            }
                        meta_out = mp | FMD_LIST_RESUME | (meta & 0xffff0000u);
                    }
                }
                const uint32_t n = (uint32_t)__popcll(hm), room = res_end - res_cur;   // n <= 64 = LANE_CHUNK
                uint32_t base = 0;
                if (room < n) {   // the first `room` of them finish the old chunk, the rest start a new one: only a wave's LAST chunk keeps holes
                    if (lane == 0) { base = atomicAdd(gen_n, (uint32_t)LANE_CHUNK); atomicAdd(bail_n, n_handed); }
                    n_handed = 0;
                    base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                    gen_list[2 * (size_t)(base + lane)] = FMD_LIST_HOLE; gen_list[2 * (size_t)(base + lane) + 1] = 0;
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the markers land before the entries that replace them
                }
                if (hand) {
                    const uint32_t q = (uint32_t)fmd_below(hm), k = q < room ? res_cur + q : base + (q - room);
                    gen_list[2 * (size_t)k] = sid; gen_list[2 * (size_t)k + 1] = meta_out;
                    am = 0;
                }
                if (room < n) { res_cur = base + (n - room); res_end = base + LANE_CHUNK; }
                else res_cur += n;
                n_handed += n;
                hand = 0;
            }
        }
        // ---- admission, in two steps whose loads ride under the other lanes' gather (one s_waitcnt for all): (1) a ticket and the list entry of that
        // position; (2) the strand's candidates (overlap_intv's list, longest overlap first) -- m entries of 32 bytes from the lane's own row, issued at once
        // (a loop that waits for every entry in turn costs the wave m round trips: the first form of this kernel, 15 ms SLOWER than the group form)
        // (more than 8 candidates come in two parts, the widest half first: 8 raw words per candidate sit in registers from the loads to the s_waitcnt)
        uint4 raw[2 * RAWN];
        uint32_t l0 = 0, l1 = 0, g_in = 0;
        // (the candidates' decode is RAWN x 40 instructions for the wave however many lanes take part: not before `slow_min` lanes wait for it, unless nobody has a round to do)
        // (both ballots are taken by the WHOLE wave before the per-lane test: behind `st >= 2 &&` they would run with EXEC cut down to the waiting lanes,
        // none of which is `active`, and the gate would always be open -- which is how rounds 5's numbers were taken; FMD_LANE_ADM_GATE=0 is that form)
        const uint64_t adm_wait_m = __ballot(st >= 2), adm_act_m = __ballot(active);
        const bool ld2 = st >= 2 && ((uint32_t)__popcll(adm_wait_m) >= adm_min || adm_act_m == 0);
        const uint32_t adm_m = meta & 0xffffu, adm_h0 = HALVES == 2 ? adm_m / 2 : 0u;                 // the parts: [h0, m) first, then [0, h0)
        const uint32_t adm_base = st == 2 ? adm_h0 : 0u, adm_cnt = st == 2 ? adm_m - adm_h0 : adm_h0;
        if (ld2) {
            const uint4 *q0 = (const uint4 *)(listA + sid * (size_t)cap + (cap - adm_m) + adm_base);
#pragma unroll
            for (int t = 0; t < RAWN; ++t) if ((uint32_t)t < adm_cnt) { raw[2 * t] = q0[2 * t]; raw[2 * t + 1] = q0[2 * t + 1]; }
            if (st == 2) g_in = gidx ? gidx[sid] : sid;
        }
        bool ld1 = false;
        {
            const bool want = st == 0 && !active && !drained && hand == 0;
            const size_t p = fmd_tickets_take(tk, queue, want, N);
            if (want && (p == (size_t)-1 || p >= N)) drained = true;
            else if (want) { l0 = list[2 * p]; l1 = list[2 * p + 1]; ld1 = true; }
        }
        // ---- the window of every lane's strand: BWT[X1, X1 + szw) from the lane's own block image(s).  A lane whose round is waiting for the wave's next
        // pass through the full round code below (`slow`) keeps what it saw of its window and does not fetch.
        const bool go = active && !slow;
        if (__ballot(go) == 0) fmd_fetch_wait();
        else {
            uint32_t bke, oke;
            fmd_split(X1 - 1, bke, oke);                              // x[1] >= cnt[1] > 0 for base strings
            const uint4 *img_k, *img_l;
            int t_k, t_l;
            uint32_t ble = bke;
            bool sep = false;
            if (WIDE) {   // a window of up to 63 positions may leave the block of X1 - 1 (which holds 96): the second block through the engine's l side
                const FmdRank2 r = fmd_wave_rank2_fetch(ix, lds, go ? X1 - 1 : NONE64, go ? X1 - 1 + szw : NONE64);
                img_k = r.bk; img_l = r.bl; t_k = r.t; t_l = r.tl; ble = r.blk_l; sep = go && ble != bke;
            } else {      // at most 31 positions from offset <= 64: always inside the 96 of one block
                fmd_fetch_slot<0>(ix, lds, bke, go);
                fmd_fetch_wait();
                img_k = img_l = lds + fmd_lds_base(lane, 0); t_k = t_l = fmd_chunk_xor(lane);
            }
            M X, Y, Z;
            {
                uint4 a, b, c;
                grp_window(img_k, t_k, img_l, t_l, bke, ble, true, sep, bke, oke, a, b, c);
                const uint32_t sh = (uint32_t)X1 & 31;
                if (WIDE) { X = (M)win64(a.x, b.x, c.x, sh); Y = (M)win64(a.y, b.y, c.y, sh); Z = (M)win64(a.z, b.z, c.z, sh); }
                else { X = (M)__builtin_amdgcn_alignbit(b.x, a.x, sh); Y = (M)__builtin_amdgcn_alignbit(b.y, a.y, sh); Z = (M)__builtin_amdgcn_alignbit(b.z, a.z, sh); }
            }
            // The base the strand goes on with (k_ovl_nei_fast: usually every read of the window that does not end here shows the same one)
            const M mw = W::below(szw);
            const M any = (X | Y | Z) & mw, xa = X & any, ya = Y & any, za = Z & any;
            const M ends = ~(X | Y | Z) & mw;                        // reads of the window that end here
            int cs = (xa ? 1 : 0) | (ya ? 2 : 0) | (za ? 4 : 0);
            const bool mixed = go && ((xa && xa != any) || (ya && ya != any) || (za && za != any) || cs > 4);
            // absolute ranks at X1 - 1: of the base the strand goes on with (children) and of '$' (x[1] of a neighbour); with more than one base in the window
            // which one it is comes out of the candidates (the full round below), so all of them are taken while the block is here
            uint64_t R[6] = {0, 0, 0, 0, 0, 0};
            if (__ballot(mixed)) { if (mixed) fmd_block_rank6<false>(img_k, t_k, oke + 1, R, bke); }
            if (!mixed) { uint64_t rz; const uint64_t rc = fmd_block_rank1z(img_k, t_k, oke + 1, cs, bke, rz); R[0] = rz; R[1] = R[2] = R[3] = R[4] = rc; }
            if (go) {
                if (!mixed && ends == 0) {
                    // the QUIET round -- no read of the window ends, all go on with one base: nothing ends, nothing is a neighbour, every candidate keeps
                    // its size, its offset in the window, its D (alive => D != 0: a read starts with it) and r0; the window moves by one LF step.
                    if (!(lf & 0x10000u)) lf = round + 1;
                    eb |= (uint64_t)(uint32_t)(4 - cs) << (2 * (round - eb0));
                    ++round;
                    if (round - eb0 == 32) { lane_flush_bases(seq_out, seq_stride, gs, meta >> 16, eb0, round, eb); eb = 0; eb0 = round; }
                    X1 = (cs == 1 ? ix.cnt[1] : cs == 2 ? ix.cnt[2] : cs == 3 ? ix.cnt[3] : ix.cnt[4]) + R[1];
                } else { slow = true; sX = X; sY = Y; sZ = Z; sR0 = R[0]; sR1 = R[1]; sR2 = R[2]; sR3 = R[3]; sR4 = R[4]; scs = cs | (mixed ? 8 : 0); }
            }
        }
        // ---- the full round, for the lanes that wait for it, when enough of them do (its loops over the candidates are most of this kernel's instructions: a wave
        // pays them per PASS, not per lane) or when nobody has anything else to do
        {
            const uint64_t sm = __ballot(slow);
            if (sm && ((uint32_t)__popcll(sm) >= slow_min || __ballot(active && !slow) == 0)) {
                if (slow) {
                    const M X = sX, Y = sY, Z = sZ;
                    int cs = scs & 7;
                    const bool mixed = (scs & 8) != 0;
                    const M mw = W::below(szw), any = (X | Y | Z) & mw, ends = ~(X | Y | Z) & mw;
                    const M m1 = X & ~Y & ~Z & mw, m2 = ~X & Y & ~Z & mw, m3 = X & Y & ~Z & mw, m4 = ~X & ~Y & Z & mw;
                    bool bail = false;
                    if (mixed) {   // which bases does a read that STARTS inside some candidate go on with (a surviving child, unitig.c:126-134)?
                        const M m5 = X & ~Y & Z & mw;
                        uint32_t u = 0;
                        for (uint32_t j = 0; j < m_lane; ++j) {
                            if ((am >> j) & 1u) {
                                const uint4 e = mc[j * 64];
                                const uint32_t d = LC_D(e.w), sz = LC_SZ(e.w);
                                const M mine = W::below(sz) << d;
                                const uint32_t nS = W::popc(ends & mine);
                                const uint32_t s1 = W::popc(m1 & mine), s2 = W::popc(m2 & mine), s3 = W::popc(m3 & mine), s4 = W::popc(m4 & mine), s5 = W::popc(m5 & mine);
                                const uint32_t o4 = nS, o3 = o4 + s4, o2 = o3 + s3, o1_ = o2 + s2, o5 = o1_ + s1;
                                const M D = CD(e);
                                if ((D >> o1_) & W::below(s1)) u |= 2u;
                                if ((D >> o2) & W::below(s2)) u |= 4u;
                                if ((D >> o3) & W::below(s3)) u |= 8u;
                                if ((D >> o4) & W::below(s4)) u |= 16u;
                                if ((D >> o5) & W::below(s5)) u |= 32u;
                            }
                        }
                        bail = __popc(u) >= 2 || (u & 32u);             // a fork (or an N to follow): the general kernel's business
                        cs = u ? __ffs((int)u) - 1 : 0;
                        if (cs > 4) cs = 0;
                    }
                    const uint64_t Rz = sR0, Rc = cs == 1 ? sR1 : cs == 2 ? sR2 : cs == 3 ? sR3 : cs == 4 ? sR4 : sR0;
                    const M Cw = !mixed ? any : cs == 1 ? m1 : cs == 2 ? m2 : cs == 3 ? m3 : cs == 4 ? m4 : (M)0;   // the positions of cs in the window
                    slow = false;
                    const uint32_t ori_l = meta >> 16;
                    if (bail) { hand = 2; active = false; lane_flush_bases(seq_out, seq_stride, gs, ori_l, eb0, round, eb); }   // (the candidates stay as they are until the hand-over at the top of the next step)
                    else {
                    // ---- every candidate of the strand: is it a neighbour (unitig.c:111-122), does a read that starts with it go on (unitig.c:129)?
                    uint32_t nei_m = 0, child_m = 0;
                    for (uint32_t j = 0; j < m_lane; ++j) {
                        if ((am >> j) & 1u) {
                            const uint4 e = mc[j * 64];
                            const uint32_t d = LC_D(e.w), sz = LC_SZ(e.w);
                            const M mine = W::below(sz) << d, D = CD(e);
                            const uint32_t nS = W::popc(ends & mine);
                            uint32_t nc, coff;
                            if (mixed) {
                                const uint32_t s1 = W::popc(m1 & mine), s2 = W::popc(m2 & mine), s3 = W::popc(m3 & mine), s4 = W::popc(m4 & mine);
                                const uint32_t o4 = nS, o3 = o4 + s4, o2 = o3 + s3, o1_ = o2 + s2;
                                nc = cs == 1 ? s1 : cs == 2 ? s2 : cs == 3 ? s3 : cs == 4 ? s4 : 0u;
                                coff = cs == 1 ? o1_ : cs == 2 ? o2 : cs == 3 ? o3 : o4;
                            } else { nc = W::popc(any & mine); coff = sz - nc; }
                            if (round > 0 && nS == sz && D == W::below(sz)) nei_m |= 1u << j;
                            if ((D >> coff) & W::below(nc)) child_m |= 1u << j;
                        }
                    }
                    if (!(lf & 0x10000u)) lf = child_m ? round + 1 : (FMD_LFORK_ALL | 0x10000u);   // check_left's rounds (FMD_LFORK_*): before any masking
                    const int f = nei_m ? __ffs((int)nei_m) - 1 : 32;           // the first neighbour masks the rest of the (only) category
                    const uint32_t keep_m = f >= 32 ? child_m : child_m & ((1u << f) - 1u);
                    if (nei_m) {
                        const uint4 e = mc[f * 64];
                        const uint32_t d = LC_D(e.w), sz = LC_SZ(e.w), pos = LC_POS(e.w);
                        if (n_nei == 0) nei0 = ori_l - pos;                       // info of nei[0] decides rbeg (unitig.c:157)
                        if (n_nei < max_nei)
                            store_entry(nei_out + gs * (size_t)max_nei + n_nei, LC_R0(e.w, e.z), ix.cnt[0] + Rz + W::popc(ends & W::below(d)), (uint64_t)sz, (uint64_t)(ori_l - pos));
                        ++n_nei;
                    }
                    if (n_nei > max_nei) { // more neighbours than the caller has room for: flagged, re-run larger by the caller
                        rec[gs].flags |= FMD_OVLP_F_OVERFLOW;
                        active = false; am = 0;
                    } else if (keep_m) {   // next round (unitig.c:137-153)
                        eb |= (uint64_t)(uint32_t)(4 - cs) << (2 * (round - eb0));   // comp6(cs) - 1, cs in 1..4
                        ++round;
                        if (round - eb0 == 32) { lane_flush_bases(seq_out, seq_stride, gs, ori_l, eb0, round, eb); eb = 0; eb0 = round; }
                        const int wl = 31 - __clz((int)keep_m);                 // the widest child is the last one kept
                        uint32_t wb = 0, new_szw = 0;
                        for (int j = wl; j >= 0; --j) {                         // (downwards: the widest child first, its offset is everybody's origin)
                            if ((keep_m >> j) & 1u) {
                                const uint4 e = mc[j * 64];
                                const uint32_t d = LC_D(e.w), sz = LC_SZ(e.w), pos = LC_POS(e.w);
                                const M mine = W::below(sz) << d, D = CD(e);
                                const uint32_t nS = W::popc(ends & mine);
                                uint32_t nc, coff;
                                if (mixed) {
                                    const uint32_t s1 = W::popc(m1 & mine), s2 = W::popc(m2 & mine), s3 = W::popc(m3 & mine), s4 = W::popc(m4 & mine);
                                    const uint32_t o4 = nS, o3 = o4 + s4, o2 = o3 + s3, o1_ = o2 + s2;
                                    nc = cs == 1 ? s1 : cs == 2 ? s2 : cs == 3 ? s3 : cs == 4 ? s4 : 0u;
                                    coff = cs == 1 ? o1_ : cs == 2 ? o2 : cs == 3 ? o3 : o4;
                                } else { nc = W::popc(any & mine); coff = sz - nc; }
                                const uint32_t before = W::popc(Cw & W::below(d));       // cs's of the window before my range
                                if (j == wl) { wb = before; new_szw = nc; }
                                const uint64_t r0 = LC_R0(e.w, e.z) + W::popc(D & W::below(coff));
                                const M Dn = (D >> coff) & W::below(nc);
                                mc[j * 64] = make_uint4((uint32_t)Dn, WIDE ? (uint32_t)((uint64_t)Dn >> 32) : 0u, (uint32_t)r0, LC_PACK(r0, before - wb, nc, pos));
                            }
                        }
                        X1 = (cs == 1 ? ix.cnt[1] : cs == 2 ? ix.cnt[2] : cs == 3 ? ix.cnt[3] : ix.cnt[4]) + Rc + wb;
                        szw = new_szw;
                        am = keep_m;
                    } else { // every path is closed (unitig.c:154-178); nothing forked, so no fix-up
                        lane_flush_bases(seq_out, seq_stride, gs, ori_l, eb0, round, eb);
                        fmd_ovlp_rec_t *o = rec + gs;
                        o->lfork = (uint16_t)(lf & 0xffffu);
                        o->rbeg = n_nei ? (int)(ori_l - nei0) : -1;
                        o->ext_len = n_nei > 1 ? 0 : (int)round;
                        o->n_nei = (int32_t)n_nei;
                        active = false; am = 0;
                    }
                    }
                }
            }
        }
        // ---- what the admission loads of this step brought
        if (ld2) {
            const uint32_t m = adm_m, ori_l = meta >> 16;
            if (st == 2) {
                gs = g_in;
                round = 0; n_nei = 0; lf = 0; nei0 = 0; am = 0; eb = 0; eb0 = 0;
                adm_bad = m == 0 || m > (uint32_t)G || ori_l > LANE_MAX_LEN;
                uint64_t x1w = 1; uint32_t szw_ = 1;
#pragma unroll
                for (int t = 0; t < RAWN; ++t) if ((uint32_t)t + 1 == adm_cnt) { const FmdCand cw = cand_decode(raw[2 * t], raw[2 * t + 1]); x1w = cw.x1; szw_ = (uint32_t)cw.sz; }   // the widest candidate is the last: its range holds the others'
                adm_bad |= szw_ > W::MAXW || szw_ == 0;
                X1 = x1w; szw = szw_;
                m_lane = adm_bad ? 0u : m;
            }
#pragma unroll
            for (int t = 0; t < RAWN; ++t)
                if ((uint32_t)t < adm_cnt && !adm_bad) {
                    const uint32_t j = adm_base + (uint32_t)t;
                    const FmdCand cd = cand_decode(raw[2 * t], raw[2 * t + 1]);
                    const uint64_t d = cd.x1 - X1;
                    adm_bad |= !cd.narrow || cd.x1 < X1 || d + cd.sz > szw || cd.sz == 0 || cd.depth > ori_l;
                    mc[j * 64] = make_uint4((uint32_t)cd.D, (uint32_t)(cd.D >> 32), (uint32_t)cd.r0, LC_PACK(cd.r0, d & 63u, cd.sz & 63u, (ori_l - cd.depth) & 0xfffu));
                    am |= 1u << j;
                }
            if (st == 2 && adm_h0 != 0 && !adm_bad) st = 3;        // the other half in the next admission pass
            else {
                if (adm_bad) { hand = 1; am = 0; }     // (sid and meta stay until the hand-over at the top of the next step)
                else active = true;
                st = 0;
            }
        }
        if (ld1) { sid = l0; meta = l1; st = 2; }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // the images are read before the next gather lands in the slots
        if (__ballot(active || st != 0 || hand != 0 || !drained) == 0) { if (lane == 0 && n_handed) atomicAdd(bail_n, n_handed); break; }
    }
}

static int lane_blocks_per_cu(const void *fn, size_t lds_bytes)
{
    return fmd_resident_per_cu(fn, lds_bytes, 20, "k_ovl_nei_lane");
}
int fmd_nei_lane_enabled(void)
{
    const char *e = getenv("FMD_NEI_LANE");
    return !(e && atoi(e) == 0);
}
// classes whose candidates fit the registers of a lane: up to 21 (the larger group size is the group kernels' business)
int fmd_nei_lane_class_ok(int cls, int wide) { (void)wide; return cls >= 0 && cls <= 4; }   // up to 21 candidates (32: the group form)

void fmd_launch_nei_lane(int cls, int wide, int n_cu, int per_cu_cap, hipStream_t st, const FmdIndexView &ix, const uint32_t *list, const uint32_t *list_n, uint32_t cap,
                         const fmd_intv_t *listA, fmd_intv_t *listB, fmd_ovlp_rec_t *rec, fmd_intv_t *nei_out, uint32_t max_nei, uint8_t *seq_out,
                         uint32_t seq_stride, uint32_t *gen_list, uint32_t *gen_n, uint32_t *bail_n, const uint32_t *gidx)
{
    uint32_t *queue = (uint32_t *)list_n + FMD_DEAL_WORD_LANE;      // the ticket counter: a word of the list counter's own line (zeroed with the header of the batch)
    const char *e = getenv("FMD_LANE_WAVES");                       // A/B knob: resident waves per CU
    const char *et = getenv("FMD_LANE_TICKETS");                    // largest ticket chunk (guided sizes below it)
    const uint32_t tk_chunk = et && atoi(et) >= 16 ? (uint32_t)atoi(et) : 256u;
    const char *es = getenv("FMD_LANE_BATCH");                      // lanes that must wait for the full round code (and for the candidates' decode) before the wave runs it
    const uint32_t slow_min = es && atoi(es) >= 1 ? (uint32_t)atoi(es) : 32u;
    const char *ea = getenv("FMD_LANE_ADM_GATE");                   // lanes that must wait for the candidates' decode of an admission (1 = never held back: the form every number up to round 5 was taken with)
    const uint32_t adm_min = ea && atoi(ea) >= 1 ? (uint32_t)atoi(ea) : 1u;
#define LANE_LAUNCH_(K, MM) do { \
        static int cached = 0; \
        if (!cached) cached = lane_blocks_per_cu((const void *)k_ovl_nei_lane<fmd_grp_size(K), MM>, sizeof(uint4) * ((sizeof(MM) == 8 ? FMD_WAVE_LDS_U4 : FMD_SLOT_U4) + 64 * fmd_grp_size(K))); \
        int per = cached; \
        if (per_cu_cap > 0 && per_cu_cap < per) per = per_cu_cap; \
        if (e && atoi(e) > 0 && atoi(e) < per) per = atoi(e); \
        int grid = n_cu * per; \
        if (grid > FMD_FAST_MAX_WAVES) grid = FMD_FAST_MAX_WAVES; \
        k_ovl_nei_lane<fmd_grp_size(K), MM><<<grid, 64, 0, st>>>(ix, list, list_n, cap, listA, listB, rec, nei_out, max_nei, seq_out, seq_stride, gen_list, gen_n, bail_n, gidx, queue, tk_chunk, slow_min, adm_min); \
    } while (0)
#define LANE_LAUNCH2(K) do { if (wide) LANE_LAUNCH_(K, uint64_t); else LANE_LAUNCH_(K, uint32_t); } while (0)
    switch (cls) {
    case 0: LANE_LAUNCH2(0); break;
    case 1: LANE_LAUNCH2(1); break;
    case 2: LANE_LAUNCH2(2); break;
    case 3: LANE_LAUNCH2(3); break;
    default: LANE_LAUNCH2(4); break;
    }
#undef LANE_LAUNCH2
#undef LANE_LAUNCH_
}
