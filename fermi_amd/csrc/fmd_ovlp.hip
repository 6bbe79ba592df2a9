// fmd_ovlp.hip -- all-vs-all overlap discovery for unitig construction: the read-only front half
// of unitig1 (unitig.c:274-300) for a batch of sequence ids, one lane per read-strand:
//     fm_retrieve (exact.c:59)  ->  fm6_is_contained / overlap_intv (unitig.c:38-91)
//                               ->  fm6_get_nei (unitig.c:93-179, used = sorted = NULL)
// Phase-uniform persistent kernels on the wave engine (fmd_wave.h): the fused LF-walk + overlap_intv (k_ovl_walk), then fm6_get_nei by group
// kernels (fmd_ovlp_grp.hip) and, for what they set aside, one lane per strand (k_ovl_nei); candidate interval lists travel between the phases
// through an HBM work area.  Every lane free-runs its own search and posts one rank2a request per wave step; finished lanes refill from a queue.
#include <stdlib.h>
#include <string.h>
#include "fmd_kernel_common.h"

// (gidx: slot of a sorted batch -> row of rec / nei_out / seq_out, fmd_ovlp_sorted_dev; nullptr = the slot is the row)
void fmd_launch_nei_grp(int cls, int n_cu, int per_cu_cap, hipStream_t st, const FmdIndexView &ix, const uint32_t *list, const uint32_t *list_n, uint32_t cap,
                        const fmd_intv_t *listA, fmd_intv_t *listB, const FmdOvlClasses &cl, fmd_ovlp_rec_t *rec, fmd_intv_t *nei_out, uint32_t max_nei, uint8_t *seq_out,
                        uint32_t seq_stride, uint32_t *slow_list, uint32_t *slow_n, const uint32_t *gidx, size_t fix_off, uint32_t down_cap, int second_pass);
int fmd_nei_fast_available(void);
int fmd_nei_lane_enabled(void);
int fmd_nei_lane_class_ok(int cls, int wide);
void fmd_launch_nei_lane(int cls, int wide, int n_cu, int per_cu_cap, hipStream_t st, const FmdIndexView &ix, const uint32_t *list, const uint32_t *list_n, uint32_t cap,
                         const fmd_intv_t *listA, fmd_intv_t *listB, fmd_ovlp_rec_t *rec, fmd_intv_t *nei_out, uint32_t max_nei, uint8_t *seq_out,
                         uint32_t seq_stride, uint32_t *gen_list, uint32_t *gen_n, uint32_t *bail_n, const uint32_t *gidx);
void fmd_launch_nei_fast(int cls, int wide, int n_cu, int per_cu_cap, hipStream_t st, const FmdIndexView &ix, const uint32_t *list, const uint32_t *list_n, uint32_t cap,
                         const fmd_intv_t *listA, fmd_intv_t *listB, fmd_ovlp_rec_t *rec, fmd_intv_t *nei_out, uint32_t max_nei, uint8_t *seq_out,
                         uint32_t seq_stride, uint32_t *gen_list, uint32_t *gen_n, uint32_t *bail_n, uint32_t *slow_list, uint32_t *slow_n, const uint32_t *gidx);
void fmd_launch_classify(hipStream_t st, size_t n, const fmd_ovlp_rec_t *rec, const fmd_intv_t *listA, uint32_t cap, FmdOvlClasses cl, int use_fast, const uint32_t *gidx);
// fmd_ovlp_sort.hip: minimizer keys of the parked strands, then their rows sorted by key (-> vals_b)
size_t fmd_park_sort_temp_bytes(size_t n);
int fmd_park_sort(hipStream_t st, size_t n, const FmdWalkPark *park, uint32_t *keys_a, uint32_t *keys_b, uint32_t *vals_a, uint32_t *vals_b, void *tmp, size_t tmp_bytes);

// ------------------------------------------------- phases 0+A fused: LF-walk + overlap_intv in one pass
// fm_retrieve walks the rows k_i of the suffixes "last i bases $" of the sequence; overlap_intv
// extends the interval I_i of "last i bases" backward.  k_i lies INSIDE I_i, so once I_i is
// narrower than a rank block the LF step and the extension read the SAME block: one gather per
// base instead of two (while I_i is still wide -- the first ~log4(n) bases, blocks that live in
// L2 -- the LF step takes a gather of its own).  The base found by the LF step is the base the
// extension needs; when it is '$' the same ranks are fm6_is_contained's left test (unitig.c:83-85).
// Candidates are pushed with info = depth (their start is len - depth, known only at the end).
#ifndef FMD_HEAD_AUX
#define FMD_HEAD_AUX 0
#endif
enum { WK_IDLE = 0, WK_LF, WK_EXT, WK_BOTH, WK_RIGHT, WK_ADM1, WK_ADM2 };
// can the LF step at row k be read from a block the backward extension of [x0, x0 + sz) brings in anyway (the block of x0 - 1, or
// the block of its other end when that one does not reach it)?
__device__ __forceinline__ bool walk_lf_shares(uint64_t k, uint64_t x0, uint64_t sz)
{
    // k lies inside [x0, x0 + sz); a range of at most 64 positions touches two consecutive blocks at most -- those of its two ends
    if (sz <= 63) return true;
    uint32_t o;
    return fmd_in_block(k, fmd_blk_of(x0 - 1), o) || fmd_in_block(k, fmd_blk_of(x0 - 1 + sz), o);
}

// every 4th base: the word moves into its place of the 16-byte group; every 16th: one store
#define WALK_STASH_WORD()                                                                                  \
    do {                                                                                                   \
        const uint32_t wq_ = (depth >> 2) & 3;                                                             \
        if (wq_ == 1) pk0 = pack; else if (wq_ == 2) pk1 = pack; else if (wq_ == 3) pk2 = pack;            \
        else {                                                                                             \
            if (depth <= stride_r) *(uint4 *)(srev + sid * (size_t)stride_r + depth - 16) = make_uint4(pk0, pk1, pk2, pack); \
            pk0 = pk1 = pk2 = 0;                                                                           \
        }                                                                                                  \
        pack = 0;                                                                                          \
    } while (0)

// one more base of the sequence (the one at position `depth` from its end).  WALK_HEAD keeps its 32 bases in the four stash registers,
// 4 bits each (FmdWalkPark::bases), and never stores
#define WALK_PUT_BASE(cc)                                                                                  \
    do {                                                                                                   \
        if (MODE == WALK_TAIL2) {   /* 2 bits per base, a word of 16 into the lane's LDS stash */          \
            pack |= (((uint32_t)(cc) - 1u) & 3u) << (2 * (depth & 15));                                    \
            if ((uint32_t)(cc) > 4u) flags |= WALK_F_HASN;                                                 \
            ++depth;                                                                                       \
            if ((depth & 15) == 0) { if (depth <= WALK_LS_BASES) walk_ls[((depth >> 4) - 1) * 64 + fmd_lane()] = pack; pack = 0; } \
        } else                                                                                             \
        if (MODE == WALK_HEAD) {                                                                           \
            const uint32_t v_ = (uint32_t)(cc) << (4 * (depth & 7)), w_ = depth >> 3;                      \
            pk0 |= w_ == 0 ? v_ : 0u; pk1 |= w_ == 1 ? v_ : 0u; pk2 |= w_ == 2 ? v_ : 0u; pack |= w_ == 3 ? v_ : 0u; \
            ++depth;                                                                                       \
        } else {                                                                                           \
            pack |= (uint32_t)(cc) << (8 * (depth & 3));                                                   \
            ++depth;                                                                                       \
            if ((depth & 3) == 0) WALK_STASH_WORD();                                                       \
        }                                                                                                  \
    } while (0)

// Two-pass form (WALK_HEAD + WALK_TAIL, the locality sort of fmd_ovlp_sorted_dev below).  Strands whose last bases lie next to each
// other on the genome visit the SAME rank blocks (the interval of "g[a, e)" holds the interval of "g[a, e + d)"), d steps apart; in
// id order they are never in flight together and every one of those visits is a DRAM miss.  WALK_HEAD takes every strand of the job
// FMD_WALK_SPLIT bases in and parks it (FmdWalkPark: row, bi-interval, the bases so far); the strands are sorted by the minimizer of
// those bases (k_ovl_park_keys), so that strands of one genomic window sit in neighbouring lanes; WALK_TAIL picks each strand up where
// it was parked, in that order.  Nothing can be pushed before min_match >= FMD_WALK_SPLIT bases, so the two passes together make
// exactly the steps of the one-pass walk and leave the same records, candidates and stash.
enum { WALK_WHOLE = 0, WALK_HEAD = 1, WALK_TAIL = 2, WALK_TAIL2 = 3 };   // (FMD_WALK_SPLIT, FmdWalkPark: fmd_kernel_common.h)

// WALK_TAIL2 = WALK_TAIL for sequences of at most WALK_LS_BASES bases, without the stash in HBM and without k_ovl_seq_out behind it: the bases wait in
// LDS, 2 bits each (code - 1; 7 words per lane: what is left of a CU's 160 KiB beside the gather's 8.25 KiB per wave at 16 waves), and the lane that
// reaches its sequence's '$' writes the caller's row itself, in read order, from those words (walk_emit_row: ~250 instructions once per strand, where
// the separate kernel read 112 + wrote 100 bytes per strand and cost the step 14 ms of its 272).  Reads of one length finish a wave together, so
// nobody waits.  A sequence that holds an N (2 bits do not) is put on a list and k_ovl_seq_redo writes its row afterwards from what WALK_HEAD parked.
#define WALK_LS_WORDS 7
#define WALK_LS_BASES (16 * WALK_LS_WORDS)
#define WALK_F_HASN 0x80000000u       // (in the walk's `flags` register only: never stored)
// the strand's FIRST candidate (the widest: shortest overlap) as k_ovl_classify reads it back from listA -- size <= 63, narrow form, size > 31
#define WALK_F_W63 0x40000000u
#define WALK_F_WNARROW 0x20000000u
#define WALK_F_W32 0x10000000u
#define WALK_F_INTERNAL 0xf0000000u
struct __attribute__((packed, aligned(4))) WalkU4 { uint32_t x, y, z, w; };   // a 16-byte store at a 4-byte aligned address
// four bases, 2 bits each, first found (= LAST in read order) in the low bits -> their nt6 codes as the four bytes of a word in read order
__device__ __forceinline__ uint32_t walk_expand4(uint32_t win8)
{
    uint32_t y = (win8 | win8 << 12) & 0x000f000fu;
    y = (y | y << 6) & 0x03030303u;
    return __builtin_bswap32(y + 0x01010101u);
}
// eight bases as nibbles (nt6 codes 1..5) -> 2 bits each (code - 1) in the low 16 bits; bit 16 set when one of them is not A/C/G/T
__device__ __forceinline__ uint32_t walk_nib_to_2bit(uint32_t v)
{
    const uint32_t t = v - 0x11111111u;
    uint32_t x = t & 0x33333333u;
    x = (x | x >> 2) & 0x0f0f0f0fu; x = (x | x >> 4) & 0x00ff00ffu; x = (x | x >> 8) & 0x0000ffffu;
    return x | ((t & 0xccccccccu) ? 0x10000u : 0u);
}
// st[w * 64]: word w of this lane (bases 16w .. 16w + 15 in the order found, i.e. from the sequence's end); len <= WALK_LS_BASES; dst 4-byte aligned
// with room for len + 3 bytes.  Output word w holds the bases found as [len - 4 - 4w, len - 4w): with r = len & 3 the words line up with the stash
// shifted by r bases, 16 bases = four output words = one 16-byte store.  The word that holds the sequence's last bases (r of them) is padded with
// zeros, bytes beyond it are not written (as k_ovl_seq_out leaves a row).
__device__ __forceinline__ void walk_emit_row(const uint32_t *st, uint32_t len, uint8_t *dst)
{
    const uint32_t r = len & 3u, sh = 2u * r;
    const int q = (int)(len >> 2);
    uint32_t prev = 0;
#pragma unroll
    for (int t = 0; t <= WALK_LS_WORDS; ++t) {
        const uint32_t cur = t < WALK_LS_WORDS ? st[t * 64] : 0u;
        const uint32_t S = __builtin_amdgcn_alignbit(cur, prev, sh);      // bases found as [16 (t - 1) + r, 16 t + r)
        prev = cur;
        const int w0 = q - 4 * t;                                         // S >> 24 -> word w0, ..., S & 0xff -> word w0 + 3
        if (t == 0) { if (r) *(uint32_t *)(dst + 4 * q) = walk_expand4(S >> 24) & ((1u << (8u * r)) - 1u); }
        else if (w0 >= 0) {
            WalkU4 v; v.x = walk_expand4(S >> 24); v.y = walk_expand4((S >> 16) & 0xffu); v.z = walk_expand4((S >> 8) & 0xffu); v.w = walk_expand4(S & 0xffu);
            *(WalkU4 *)(dst + 4 * w0) = v;
        } else if (w0 + 3 >= 0) {                                         // the sequence's first words: fewer than four
            if (w0 + 1 >= 0) *(uint32_t *)(dst + 4 * (w0 + 1)) = walk_expand4((S >> 16) & 0xffu);
            if (w0 + 2 >= 0) *(uint32_t *)(dst + 4 * (w0 + 2)) = walk_expand4((S >> 8) & 0xffu);
            *(uint32_t *)(dst + 4 * (w0 + 3)) = walk_expand4(S & 0xffu);
        }
    }
}

// MODE = WALK_HEAD: item t = admission record t (k_ovl_head_adm: the strand's row in ids[], park[] and rec[] and where its walk stands
// behind the tail table); the first 32 bases stay in registers and leave with the parked state in ONE 64-byte burst.
// MODE = WALK_TAIL: item = slot of the batch (rows of srev, listA), gidx[slot] = its row in park[], rec[] (and, for the kernels
// that follow, nei[] and seq[]).
template <int MODE>
__global__ __launch_bounds__(64, 4) void k_ovl_walk(FmdIndexView ix, size_t n, const uint64_t *__restrict__ ids, int min_match,
                                                 uint8_t *__restrict__ srev, uint32_t stride_r, uint32_t cap,
                                                 fmd_intv_t *__restrict__ listA, fmd_ovlp_rec_t *__restrict__ rec,
                                                 uint8_t *__restrict__ seq_out, uint32_t seq_stride, uint32_t *__restrict__ queue,
                                                 int info_only, FmdWalkPark *__restrict__ park, const uint32_t *__restrict__ gidx,
                                                 const uint4 *__restrict__ adm, uint32_t tchunk, uint32_t *__restrict__ redo,
                                                 uint32_t *__restrict__ cls, int cls_cfg_)
{
    FMD_DECLARE_COMPACT_LDS();
    const int cls_cfg = cls_cfg_ & 0xffff, gate_n = (cls_cfg_ >> 16) & 0xff;   // (bits 16-23: the admission gate, below)
    __shared__ uint32_t walk_ls[MODE == WALK_TAIL2 ? 64 * WALK_LS_WORDS : 1];   // WALK_TAIL2: the lane's bases, word w of lane l at [w * 64 + l]
    constexpr bool TAILM = MODE == WALK_TAIL || MODE == WALK_TAIL2;
    constexpr int WAUX = MODE == WALK_HEAD ? FMD_HEAD_AUX : FMD_GLDS_AUX;   // pass 1 never asks for a line twice (strands in id order: every gather is a DRAM miss)
    size_t sid = 0;
    size_t gs = 0;                        // the strand's row in rec[] (WALK_TAIL: gidx[sid], otherwise sid)
    int st = WK_IDLE, c_pend = 0, ret = 0;
    int fin_cls = -1;                     // WALK_TAIL2 with the work lists of get_nei made here (cls != nullptr): the list of the strand this lane has just finished
    uint32_t depth = 0, npush = 0, pack = 0, flags = 0;
    uint32_t pk0 = 0, pk1 = 0, pk2 = 0;   // the stash is written 16 bases at a time (one 16-byte store per lane instead of four words)
    uint64_t k = 0, x0 = 0, x1 = 0, sz = 0;
    bool exhausted = false;
    // The first ptab_d bases need no interval arithmetic when nothing can be pushed that early
    // (min_match >= ptab_d): LF steps only (one line each instead of three), the bi-interval then comes
    // from the prefix table -- forward string for x[0] and the size, reverse complement for x[1].
    const bool tab_ok = ix.ptab != nullptr && (info_only || min_match >= ix.ptab_d) && ix.ptab_d >= 2;
    bool tab = false;
    uint32_t tfw = 0, trv = 0;
    uint4 adm_a = make_uint4(0, 0, 0, 0), adm_b = make_uint4(0, 0, 0, 0);   // WALK_HEAD: the admission record of a strand on its way in (WK_ADM1)
    FmdTickets tk_;
    fmd_tickets_init(tk_, queue, tchunk & 0xffffffu, (MODE != WALK_WHOLE && (tchunk >> 24)) ? n : 0);   // (bit 24: guided chunks, the two passes of a sorted job)
    for (;;) {
        if (MODE == WALK_TAIL2 && cls != nullptr) {
            // k_ovl_classify's work, by the lanes that finished a strand in the last step (reads of one length: the whole wave): one returning atomic per
            // list that gets entries, all of them issued at once (lane j reserves for the j-th distinct list), then every lane writes its entry.  The lists
            // and counters are where ovl_phase_b expects them (FmdOvlClasses over `cls`); depth, npush and sid are still the finished strand's.
            uint64_t rem = __ballot(fin_cls >= 0);
            if (rem) {
                const int lane = fmd_lane();
                int my_c = 0, nc = 0;
                uint32_t my_cnt = 0, my_slot = 0, my_rank = 0;
                while (rem) {
                    const int c = __builtin_amdgcn_readlane(fin_cls, __ffsll((unsigned long long)rem) - 1);
                    const uint64_t mk = __ballot(fin_cls == c);
                    if (lane == nc) { my_c = c; my_cnt = (uint32_t)__popcll(mk); }
                    if (fin_cls == c) { my_slot = (uint32_t)nc; my_rank = (uint32_t)fmd_below(mk); }
                    rem &= ~mk; ++nc;
                }
                uint32_t base = 0;
                if (lane < nc) base = atomicAdd(cls + my_c * FMD_CLS_CNT_STRIDE, my_cnt);
                base = (uint32_t)__shfl((int)base, (int)my_slot) + my_rank;
                if (fin_cls >= 0) {
                    const size_t gl = 2 * n + 2 * (size_t)FMD_FAST_RESERVE;           // words of a general list
                    uint32_t *lslow = cls + FMD_CLS_HEADER_U32 + gl * FMD_GRP_CLASSES;
                    if (fin_cls == FMD_GRP_CLASSES) lslow[base] = (uint32_t)sid;
                    else {
                        uint32_t *lst = fin_cls < FMD_GRP_CLASSES ? cls + FMD_CLS_HEADER_U32 + gl * fin_cls : lslow + n + 2 * n * (size_t)(fin_cls - FMD_GRP_CLASSES - 1);
                        lst[2 * base] = (uint32_t)sid; lst[2 * base + 1] = npush | depth << 16;
                    }
                    fin_cls = -1;
                }
            }
        }
        // The admission gate (gate_n > 1): idle lanes take their next strand only when gate_n of them are idle (or nobody is walking).  What a lane does once per
        // strand -- the parked state in, the record, the row and the work-list entry out -- costs the wave its whole code in every step in which ANY lane does
        // it; lanes that start together finish within a few steps of each other (a step in which the LF step cannot share the extension's gather sets a lane
        // back by one), lanes that refill one by one drift apart until some lane does it in every step.  MEASURED on reads of one length: no effect at 16 .. 64
        // lanes (214.2 - 214.5 ms per 10^8 strands, profiles/r6_gate) -- such strands keep a wave's lanes in step by themselves; off by default
        // (FMD_TAIL_GATE / FMD_HEAD_GATE), kept for read sets of mixed lengths, where it has not been measured.  (k_ecfix is where this halves the time.)
        const bool gate = gate_n <= 1 || __popcll(__ballot(st == WK_IDLE && !exhausted)) >= gate_n || __ballot(st != WK_IDLE) == 0;
        const size_t my = fmd_tickets_take(tk_, queue, gate && st == WK_IDLE && !exhausted, (MODE != WALK_WHOLE && (tchunk >> 24)) ? n : 0);
        if (gate && st == WK_IDLE && !exhausted) {
            // The two passes of a sorted job take a strand in over one (WALK_HEAD) or two (WALK_TAIL) wave steps: the loads are issued
            // here and complete under the gather of the other lanes (WK_ADM1 / WK_ADM2 below).  A chain of dependent loads in front of
            // the gather -- id, tail-table entry, two prefix-table entries, as the one-pass walk does it -- stalls all 64 lanes of a wave
            // whose strands live 20 steps: k_ovl_head_adm resolves that chain for every strand beforehand, streaming.
            if (TAILM) {
                if (my < n) { sid = my; gs = gidx[my]; st = WK_ADM1; }
                else exhausted = true;
            } else if (MODE == WALK_HEAD) {
                if (my < n) { sid = my; adm_a = adm[2 * my]; adm_b = adm[2 * my + 1]; st = WK_ADM1; }
                else exhausted = true;
            } else
            if (my < n) {
                sid = my; gs = my; k = ids[gs]; depth = 0; npush = 0; pack = 0; pk0 = pk1 = pk2 = 0; flags = 0; ret = 0; st = WK_LF; tab = tab_ok;
                // the first ptab_d LF steps were taken when the index was loaded (FmdIndexView::tail): pick the walk up behind them
                const unsigned long long te = (tab_ok && ix.tail && k < ix.n_seq) ? ix.tail[k] : ~0ull;
                if (te != ~0ull) {
                    tfw = (uint32_t)(te >> (64 - 2 * ix.ptab_d)); k = te & ((1ull << (64 - 2 * ix.ptab_d)) - 1);
                    for (int jb = 0; jb < ix.ptab_d; ++jb) {   // the bases into the stash, as the steps would have put them
                        WALK_PUT_BASE(((tfw >> (2 * jb)) & 3u) + 1u);
                    }
                    // reverse complement of the ptab index: the 2-bit groups in reverse order, complemented
                    { uint32_t r = __brev(~tfw) >> (32 - 2 * ix.ptab_d); trv = ((r >> 1) & 0x55555555u) | ((r & 0x55555555u) << 1); }
                    const uint4 ef = ix.ptab[tfw], er = ix.ptab[trv];
                    fmd_count_lane(ix, 2, 1);
                    x0 = (uint64_t)ef.y << 32 | ef.x;
                    sz = ((uint64_t)ef.w << 32 | ef.z) - x0 + 1;
                    x1 = (uint64_t)er.y << 32 | er.x;
                    tab = false;
                    st = walk_lf_shares(k, x0, sz) ? WK_BOTH : WK_LF;
                }
            }
            else exhausted = true;
        }
        if (__ballot(st != WK_IDLE) == 0) break;

        // ---- requests.  WK_LF: block of k only.  WK_EXT / WK_BOTH: the two ends of I's backward
        //      extension (k sits in one of them for WK_BOTH).  WK_RIGHT: forward '$' extension.
        uint64_t qk = NONE64, ql = NONE64;
        if (st == WK_LF) qk = k;
        else if (st == WK_EXT || st == WK_BOTH) { qk = x0 - 1; ql = x0 - 1 + sz; }
        else if (st == WK_RIGHT) { qk = x1 - 1; ql = x1 - 1 + sz; }
        FmdRank2c r = fmd_wave_rank2_fetch_compact<WAUX>(ix, fmd_lds, qk, ql);
        // two-phase step (more than 32 lanes straddle: wide intervals): the k-side ranks are taken now,
        // the l-side after fmd_wave_l_ready(); a narrow lane whose window straddles sits this step out
        uint64_t tk2[6] = {0, 0, 0, 0, 0, 0};
        const bool wide_ext = st == WK_EXT || (st == WK_BOTH && sz > 63);
        bool skip = false;
        if (r.two_phase) {
            if (wide_ext && r.hk) fmd_block_rank6<false>(r.bk, r.t, r.nk, tk2, r.blk_k);
            if (st == WK_RIGHT && r.hk) tk2[0] = fmd_block_rank1(r.bk, r.t, r.nk, 0, r.blk_k);
            { uint32_t ko_; skip = st == WK_BOTH && (sz <= 63 || !fmd_in_block(k, r.blk_k, ko_)) && r.l_sep; }
            if (st == WK_BOTH && sz > 63 && !skip) skip = true; // wide WK_BOTH never shares a gather in two-phase steps
            if (skip && st == WK_BOTH) st = WK_LF;   // take the LF step on its own next time, then the extension through the
                                                     // general path (a lane that merely waited could wait forever: the same
                                                     // lanes straddle again next step)
        }
        const bool was_two_phase = r.two_phase;
        fmd_wave_l_ready<WAUX>(ix, fmd_lds, r);
        if (st == WK_IDLE || skip) continue;
        if (MODE == WALK_HEAD && st == WK_ADM1) {   // the admission record has arrived (FmdHeadAdm, k_ovl_head_adm)
            gs = adm_a.x;
            depth = 0; npush = 0; pack = 0; pk0 = pk1 = pk2 = 0; flags = 0; ret = 0;
            if (adm_b.w & 1u) {   // no tail-table entry: from the sentinel, on the ordinary path
                k = (uint64_t)(adm_b.y & 0xffu) << 32 | adm_a.y; st = WK_LF; tab = tab_ok;
            } else {
                const uint32_t hi = adm_b.y;
                k = (uint64_t)(hi & 0xffu) << 32 | adm_a.y; x0 = (uint64_t)((hi >> 8) & 0xffu) << 32 | adm_a.z;
                x1 = (uint64_t)((hi >> 16) & 0xffu) << 32 | adm_a.w; sz = (uint64_t)(hi >> 24) << 32 | adm_b.x;
                // the ptab_d bases of the tail as nibbles (2-bit code + 1): 8 per word
                const uint32_t tf = adm_b.z;
                uint32_t lo = tf & 0xffffu, up = tf >> 16;
                lo = (lo | lo << 8) & 0x00ff00ffu; lo = (lo | lo << 4) & 0x0f0f0f0fu; lo = (lo | lo << 2) & 0x33333333u;
                up = (up | up << 8) & 0x00ff00ffu; up = (up | up << 4) & 0x0f0f0f0fu; up = (up | up << 2) & 0x33333333u;
                const int d = ix.ptab_d;
                pk0 = (lo + 0x11111111u) & (d >= 8 ? ~0u : (1u << (4 * d)) - 1u);
                pk1 = d > 8 ? (up + 0x11111111u) & (d >= 16 ? ~0u : (1u << (4 * (d - 8))) - 1u) : 0u;
                depth = (uint32_t)d; tab = false;
                st = walk_lf_shares(k, x0, sz) ? WK_BOTH : WK_LF;
            }
            continue;
        }
        if (TAILM && st == WK_ADM1) {   // the strand's row is known: fetch what WALK_HEAD parked there, straight into the
            const uint4 *pp = (const uint4 *)(park + gs);   // registers the state will live in (the loads land under the next gather)
            const uint4 a = pp[0], b = pp[1], cb = pp[2];
            k = (uint64_t)a.y << 32 | a.x; x0 = (uint64_t)a.w << 32 | a.z; x1 = (uint64_t)b.y << 32 | b.x; sz = (uint64_t)b.w << 32 | b.z;
            pk0 = cb.x; pk1 = cb.y; pk2 = cb.z; pack = cb.w;
            st = WK_ADM2;
            continue;
        }
        if (MODE == WALK_TAIL2 && st == WK_ADM2) {
            st = WK_IDLE;
            if (k != ~0ull) {   // (~0: the sequence ended inside the head)
                const uint32_t c0 = walk_nib_to_2bit(pk0), c1 = walk_nib_to_2bit(pk1), c2 = walk_nib_to_2bit(pk2), c3 = walk_nib_to_2bit(pack);
                walk_ls[fmd_lane()] = (c0 & 0xffffu) | c1 << 16; walk_ls[64 + fmd_lane()] = (c2 & 0xffffu) | c3 << 16;
                depth = FMD_WALK_SPLIT; npush = 0; pack = 0; pk0 = pk1 = pk2 = 0; ret = 0; tab = false;
                flags = ((c0 | c1 | c2 | c3) & 0x10000u) ? WALK_F_HASN : 0u;
                st = walk_lf_shares(k, x0, sz) ? WK_BOTH : WK_LF;
            }
            continue;
        }
        if (MODE == WALK_TAIL && st == WK_ADM2) {
            st = WK_IDLE;
            if (k != ~0ull) {   // (~0: the sequence ended inside the head)
                uint4 *sr = (uint4 *)(srev + sid * (size_t)stride_r);   // the 32 bases into the stash, one per byte
#define WALK_NIB4(v_) (((v_) & 0xfu) | ((v_) & 0xf0u) << 4 | ((v_) & 0xf00u) << 8 | ((v_) & 0xf000u) << 12)
                sr[0] = make_uint4(WALK_NIB4(pk0), WALK_NIB4(pk0 >> 16), WALK_NIB4(pk1), WALK_NIB4(pk1 >> 16));
                sr[1] = make_uint4(WALK_NIB4(pk2), WALK_NIB4(pk2 >> 16), WALK_NIB4(pack), WALK_NIB4(pack >> 16));
#undef WALK_NIB4
                depth = FMD_WALK_SPLIT; npush = 0; pack = 0; pk0 = pk1 = pk2 = 0; flags = 0; ret = 0; tab = false;
                st = walk_lf_shares(k, x0, sz) ? WK_BOTH : WK_LF;
            }
            continue;
        }

        int c = c_pend;
        // Narrow interval (size <= 63, i.e. all but the first ~log4(n) bases): everything comes from ONE
        // 64-position window of BWT[x0, x0+size) read out of the lane's block image -- the six child
        // sizes, the base at row k (k lies inside the window) and rank_c(k) -- plus ONE absolute rank
        // of ONE symbol, rank_c(x0-1).  ~150 VALU instead of two full six-symbol block ranks (~600).
        const bool narrow = st == WK_BOTH && sz <= 63;
        uint64_t ws[6] = {0, 0, 0, 0, 0, 0}, wtk = 0, wD = 0, wr0 = 0;  // wD, wr0: cand_store_narrow (fmd_kernel_common.h)
        if (narrow) {
            const uint32_t sh = (uint32_t)x0 & 31;
            uint4 a, b, cc;
            grp_window(r.bk, r.t, r.bl, r.tl, r.blk_k, r.blk_l, r.hk, r.l_sep, r.blk_k, r.nk - 1, a, b, cc); // window at x0 = (x0 - 1) + 1
            const uint64_t m = bits_below((int)sz);
            const uint64_t X = win64(a.x, b.x, cc.x, sh), Y = win64(a.y, b.y, cc.y, sh), Z = win64(a.z, b.z, cc.z, sh);
            const uint64_t lo = ~Z & m, hi = Z & ~Y & m;
            const uint64_t M0 = lo & ~Y & ~X, M1 = lo & ~Y & X, M2 = lo & Y & ~X, M3 = lo & Y & X, M4 = hi & ~X, M5 = hi & X;
            ws[0] = __popcll(M0); ws[1] = __popcll(M1); ws[2] = __popcll(M2); ws[3] = __popcll(M3); ws[4] = __popcll(M4); ws[5] = __popcll(M5);
            const uint32_t o = (uint32_t)(k - x0);                       // row k inside the window
            c = (int)(((X >> o) & 1) | ((Y >> o) & 1) << 1 | ((Z >> o) & 1) << 2);
            wtk = fmd_block_rank1z(r.bk, r.t, r.nk, c, r.blk_k, wr0);              // rank_c(x0 - 1), rank_$(x0 - 1)
            wD = M0;
            const uint64_t Mc = sel6(c, M0, M1, M2, M3, M4, M5);
            k = ix.cnt[c] + wtk + __popcll(Mc & bits_below((int)o + 1)) - 1;
        } else if (st == WK_LF || st == WK_BOTH) { // LF step at row k: base = BWT[k], k' = cnt[c] + rank_c(k) - 1
            uint32_t kb_ = r.blk_k, off;
            const bool in_k = fmd_in_block(k, r.blk_k, off);     // (WK_LF: the block asked for; WK_BOTH: one of the extension's two)
            if (!in_k) { kb_ = r.blk_l; fmd_in_block(k, r.blk_l, off); }
            const uint4 *img = in_k ? r.bk : r.bl;
            const int tt = in_k ? r.t : r.tl;
            const uint4 v = img[(int)(off >> 5) ^ tt];
            const uint32_t bit = off & 31;
            c = (int)(((v.x >> bit) & 1) | ((v.y >> bit) & 1) << 1 | ((v.z >> bit) & 1) << 2);
            k = ix.cnt[c] + fmd_block_rank1(img, tt, off + 1, c, kb_) - 1;
            if (st == WK_LF && depth > 0 && tab) { // still inside the prefix table: no extension, just collect the base
                if (c < 1 || c > 4) { // the sequence ends, or an ambiguous base: start over on the ordinary path
                    k = ids[gs]; depth = 0; pack = 0; pk0 = pk1 = pk2 = 0; tab = false;
                    continue;
                }
                tfw |= (uint32_t)(c - 1) << (2 * depth); trv = trv << 2 | (uint32_t)(4 - c);
                WALK_PUT_BASE(c);
                if ((int)depth == ix.ptab_d) {
                    const uint4 ef = ix.ptab[tfw], er = ix.ptab[trv];
                    fmd_count_lane(ix, 2, 1);
                    x0 = (uint64_t)ef.y << 32 | ef.x;
                    sz = ((uint64_t)ef.w << 32 | ef.z) - x0 + 1;   // never empty: the sequence is in the index
                    x1 = (uint64_t)er.y << 32 | er.x;
                    tab = false;
                    st = walk_lf_shares(k, x0, sz) ? WK_BOTH : WK_LF;
                }
                continue;
            }
            if (st == WK_LF && depth > 0) { c_pend = c; st = WK_EXT; continue; } // the extension needs its own gather
        }
        if (depth == 0) { // first LF step: the last base of the sequence, or an empty sequence
            if (c == 0) {
                fmd_ovlp_rec_t *o = rec + gs;
                o->rank = k; o->k[0] = o->k[1] = o->k[2] = 0; o->len = 0; o->status = -1; o->n_ovlp = 0; o->rbeg = -1;
                o->ext_len = 0; o->n_nei = 0; o->flags = 0; o->reserved = 2; o->lfork = 0;
                if (MODE == WALK_HEAD) park[gs].k = ~0ull;
                st = WK_IDLE;
                continue;
            }
            x0 = ix.cnt[c]; x1 = ix.cnt[comp6(c)]; sz = ix.cnt[c + 1] - ix.cnt[c];
            WALK_PUT_BASE(c);   // (depth 0 -> 1)
            if (c > 4) tab = false;
            tfw = (uint32_t)(c - 1) & 3; trv = (uint32_t)(4 - c) & 3;
        } else if (st == WK_EXT || st == WK_BOTH) {
            uint64_t tk[6] = {0, 0, 0, 0, 0, 0}, s[6];
            if (narrow) { // only tk[c] is ever read below
#pragma unroll
                for (int a = 0; a < 6; ++a) { s[a] = ws[a]; tk[a] = wtk; }
            } else {
                uint64_t tl[6] = {0, 0, 0, 0, 0, 0};
                if (was_two_phase) {
#pragma unroll
                    for (int a = 0; a < 6; ++a) tk[a] = tk2[a];
                } else if (r.hk) fmd_block_rank6<false>(r.bk, r.t, r.nk, tk, r.blk_k);
                if (r.hl) fmd_block_rank6<false>(r.bl, r.tl, r.nl, tl, r.blk_l);
#pragma unroll
                for (int a = 0; a < 6; ++a) s[a] = tl[a] - tk[a];
            }
            if (c != 0) { // one more base: overlap_intv's loop body (unitig.c:47-59)
                const uint64_t sc = sel6(c, s[0], s[1], s[2], s[3], s[4], s[5]);
                // (sc == 0 cannot happen: the sequence itself is in the index)
                if (MODE != WALK_HEAD && !info_only && (int)depth >= min_match && s[0]) {
                    if (npush < cap) {
                        fmd_intv_t *e = listA + sid * (size_t)cap + (cap - 1 - npush);
                        if (narrow && depth < 65536u) cand_store_narrow(e, x0, x1, (uint32_t)sz, depth, wD, wr0);
                        else store_entry(e, x0, x1, sz, (uint64_t)depth);
                        if (MODE == WALK_TAIL2 && npush == 0)
                            flags |= (sz <= 63 ? WALK_F_W63 : 0u) | (narrow && depth < 65536u ? WALK_F_WNARROW : 0u) | (sz > 31 ? WALK_F_W32 : 0u);
                    } else flags |= FMD_OVLP_F_OVERFLOW;
                    ++npush;
                }
                x0 = sel6(c, ix.cnt[0], ix.cnt[1], ix.cnt[2], ix.cnt[3], ix.cnt[4], ix.cnt[5]) + sel6(c, tk[0], tk[1], tk[2], tk[3], tk[4], tk[5]);
                uint64_t before = 0;             // sizes ordered before c: 0 <4 <3 <2 <1 <5
                if (c != 0) before += s[0];
                if (c == 3 || c == 2 || c == 1 || c == 5) before += s[4];
                if (c == 2 || c == 1 || c == 5) before += s[3];
                if (c == 1 || c == 5) before += s[2];
                if (c == 5) before += s[1];
                x1 += before; sz = sc;
                WALK_PUT_BASE(c);
            } else { // '$': the sequence is complete (len = depth); these ranks are the left test of fm6_is_contained
                if (MODE == WALK_TAIL2) {
                    if ((depth & 15) && depth <= WALK_LS_BASES) walk_ls[(depth >> 4) * 64 + fmd_lane()] = pack;   // the last, partial word
                } else
                if (MODE != WALK_HEAD && (depth & 15) && depth <= stride_r) // the last, partial group of 16 (stride_r is a multiple of 16)
                {   // completed words of the group sit in pk0..2, a partial word in pack; everything past it is zero
                    const uint32_t wq = (depth >> 2) & 3;
                    *(uint4 *)(srev + sid * (size_t)stride_r + (depth & ~15u)) = make_uint4(wq == 0 ? pack : pk0, wq == 1 ? pack : pk1, wq == 2 ? pack : pk2, wq == 3 ? pack : 0u);
                }
                fmd_ovlp_rec_t *o = rec + gs;
                o->rank = k; o->len = (int32_t)depth; o->rbeg = -1; o->ext_len = 0; o->n_nei = 0; o->reserved = 2; o->lfork = 0;
                o->k[0] = o->k[1] = o->k[2] = 0; o->n_ovlp = 0;
                if (MODE == WALK_HEAD) park[gs].k = ~0ull;   // ended inside the head: shorter than min_match, the record below is final
                if (depth > stride_r) { o->status = 0; o->flags = FMD_OVLP_F_OVERFLOW; st = WK_IDLE; continue; } // longer than max_len
                if (!info_only && (int)depth <= min_match) { o->status = -1; o->flags = 0; st = WK_IDLE; continue; } // too short (unitig.c:288)
                // (the caller's copy in read order is made by k_ovl_seq_out: a lane doing it here, from a stash in HBM, holds up the other 63)
                if (MODE == WALK_TAIL2) {          // ... from LDS it does not: every sequence with a complete record gets its row now (k_ovl_seq_out's conditions)
                    if (flags & WALK_F_HASN) redo[1 + atomicAdd(redo, 1u)] = (uint32_t)sid;
                    else walk_emit_row(walk_ls + fmd_lane(), depth, seq_out + gs * (size_t)seq_stride);
                }
                if (sz != s[0]) ret = -1;          // left-contained
                x0 = tk[0]; sz = s[0];             // ok[0]: x[0] = cnt[0] + tk[0], x[1] unchanged
                st = WK_RIGHT;
                continue;
            }
        } else if (st == WK_RIGHT) { // extend by '$' on the right (unitig.c:86-89)
            const uint64_t t0k = was_two_phase ? tk2[0] : (r.hk ? fmd_block_rank1(r.bk, r.t, r.nk, 0, r.blk_k) : 0);
            const uint64_t t0l = r.hl ? fmd_block_rank1(r.bl, r.tl, r.nl, 0, r.blk_l) : 0;
            if (sz != t0l - t0k) ret = -1;
            fmd_ovlp_rec_t *o = rec + gs;
            o->k[0] = x0; o->k[1] = t0k; o->k[2] = t0l - t0k;
            o->status = ret < 0 ? -3 : 0;
            o->n_ovlp = (int32_t)npush;
            o->flags = flags & ~WALK_F_INTERNAL;
            if (MODE == WALK_TAIL2 && cls != nullptr && ret >= 0 && npush > 0 && !(flags & FMD_OVLP_F_OVERFLOW)) {   // k_ovl_classify's rule (fmd_ovlp_grp.hip)
                int c = FMD_GRP_CLASSES;
                if ((flags & WALK_F_W63) && depth < 65535u) {
#pragma unroll
                    for (int kk = FMD_GRP_CLASSES - 1; kk >= 0; --kk) if (kk >= (cls_cfg >> 8) && npush <= (uint32_t)fmd_grp_size(kk)) c = kk;
                    if (c < FMD_GRP_CLASSES && (flags & WALK_F_WNARROW) && (cls_cfg & 1)) c += FMD_GRP_CLASSES + 1 + ((flags & WALK_F_W32) ? FMD_GRP_CLASSES : 0);
                }
                fin_cls = c;
            }
            st = WK_IDLE;
            continue;
        }
        // next base: can the LF step share the extension's gather?
        {
            st = walk_lf_shares(k, x0, sz) ? WK_BOTH : WK_LF;
            if (tab) st = WK_LF;   // inside the prefix table there is no extension to share a gather with
        }
        // park the strand: one 64-byte line, written whole.  cls_cfg (free in this mode): a depth below FMD_WALK_SPLIT at which the head hands the strand to
        // k_ovl_pair (two bases per request from there on; pad.w >> 24 = that depth), 0 = FMD_WALK_SPLIT, the strand parked for good
        // (Handing a strand over at the first even depth at which its interval is narrow -- 63 % of the strands of 30-fold reads at 14, nearly all at 16 -- was
        // measured and not kept: the head's time is its first two, wide, bases, and k_ovl_pair lost more than the head gained: 36.3 -> 39.3 ms, profiles/r6_pair.)
        // A strand whose interval is still wider than 63 at the hand-over is not handed over: it goes on here, one base at a time, and is parked for good.
        if (MODE == WALK_HEAD && !tab && (depth == FMD_WALK_SPLIT || (cls_cfg > 0 && depth == (uint32_t)cls_cfg && sz <= 63))) {
            uint4 *pp = (uint4 *)(park + gs);
            pp[0] = make_uint4((uint32_t)k, (uint32_t)(k >> 32), (uint32_t)x0, (uint32_t)(x0 >> 32));
            pp[1] = make_uint4((uint32_t)x1, (uint32_t)(x1 >> 32), (uint32_t)sz, (uint32_t)(sz >> 32));
            pp[2] = make_uint4(pk0, pk1, pk2, pack); pp[3] = make_uint4(0, 0, 0, depth < FMD_WALK_SPLIT ? depth << 24 : 0u);
            st = WK_IDLE;
        }
    }
}

#undef WALK_STASH_WORD
#undef WALK_PUT_BASE

// ---- two bases per request (round 6; fmd_pair.hip, fmd_wave.h) -----------------------------------------------------------------------
// Pass 1 of a sorted job below FMD_WALK_SPLIT, for an index that has two-base blocks: WALK_HEAD takes every strand to depth `from` (16: by then the
// interval of a strand of 30-fold reads is narrower than 64) and parks it; this kernel takes it on to FMD_WALK_SPLIT two bases per gather and parks it
// for good.  It does NOTHING else -- no single steps, no wide intervals, no sequence ends: a strand it cannot take all the way (an interval still wider than
// 63, an N or the sequence's end within the next two bases) goes on a list, and WALK_HEAD walks those again from their admission records.  That is
// what keeps it lean: one 8 KiB landing slot and ~70 registers per wave, so that a CU holds as many waves as the single-step head -- the first form of
// this (k_ovl_walk<WALK_HEADP>, profiles/r6_pair) carried the whole single-step engine beside the pair step, held 8 waves per CU and lost.
// A pair block starts every 32 positions and describes 96: an interval of up to 64 positions lies inside the block of its first position.
// The step: the positions of the window with first symbol c1 are the interval one base on, in order; those among them with second symbol c2 the interval
// two bases on: sizes, the x[1] sums of both extensions (fm6_extend's order 0 < 4 < 3 < 2 < 1 < 5, exact.c:81-86) and the rank of row k are popcounts,
// the start is one pair count (block + superblock: ix.pair_tab, which also holds K2[c1][c2] = cnt[c2] + #{c2 in BWT[0, cnt[c1])}).
enum { PK_IDLE = 0, PK_LOAD, PK_RUN };
#ifndef FMD_PAIR_AUX
#define FMD_PAIR_AUX FMD_HEAD_AUX      // cache-policy bits of the two-base gather (2 = nt: every line is asked for once)
#endif
#ifndef FMD_PAIR_LB
#define FMD_PAIR_LB 4                  // waves per SIMD the kernel is compiled for
#endif
__global__ __launch_bounds__(64, FMD_PAIR_LB) void k_ovl_pair(FmdIndexView ix, size_t n, FmdWalkPark *__restrict__ park, uint32_t *__restrict__ queue, uint32_t tchunk,
                                                  uint32_t *__restrict__ strag)
{
    __shared__ uint4 pair_lds[FMD_PAIR_SLOT_U4];
    const int q_ = fmd_lane(), px = fmd_pair_xor(q_);
    const uint4 *img = pair_lds + fmd_pair_base(q_);
    const uint32_t *iw = (const uint32_t *)img;
    size_t row = 0;
    int st = PK_IDLE;
    bool exhausted = false;
    uint32_t depth = 0, pk0 = 0, pk1 = 0, pk2 = 0, pk3 = 0;
    uint64_t k = 0, x0 = 0, x1 = 0, sz = 0;
    uint4 la = make_uint4(0, 0, 0, 0), lb = la, lc = la, ld = la;
    FmdTickets tk_;
    fmd_tickets_init(tk_, queue, tchunk & 0xffffffu, (tchunk >> 24) ? n : 0);
    for (;;) {
        const size_t my = fmd_tickets_take(tk_, queue, st == PK_IDLE && !exhausted, (tchunk >> 24) ? n : 0);
        if (st == PK_IDLE && !exhausted) {
            if (my < n) {   // the parked line: its loads land under the gather of the other lanes
                row = my;
                const uint4 *pp = (const uint4 *)(park + row);
                la = pp[0]; lb = pp[1]; lc = pp[2]; ld = pp[3];
                st = PK_LOAD;
            } else exhausted = true;
        }
        if (__ballot(st != PK_IDLE) == 0) break;
        fmd_pair_fetch<FMD_PAIR_AUX>(ix, pair_lds, (uint32_t)(x0 >> 5), st == PK_RUN);
        fmd_fetch_wait();
        if (st == PK_LOAD) {
            k = (uint64_t)la.y << 32 | la.x; x0 = (uint64_t)la.w << 32 | la.z; x1 = (uint64_t)lb.y << 32 | lb.x; sz = (uint64_t)lb.w << 32 | lb.z;
            pk0 = lc.x; pk1 = lc.y; pk2 = lc.z; pk3 = lc.w;
            depth = ld.w >> 24;
            st = PK_RUN;
            if (k == ~0ull || depth == 0) st = PK_IDLE;                        // the sequence ended inside the head / the strand is parked for good already
            else if (sz > 63 || depth + 2 > FMD_WALK_SPLIT) { strag[1 + atomicAdd(strag, 1u)] = (uint32_t)row; st = PK_IDLE; }
            continue;
        }
        if (st != PK_RUN) continue;
        const uint32_t off = (uint32_t)x0 & 31u;
        const uint4 A0 = img[0 ^ px], A1 = img[1 ^ px], A2 = img[2 ^ px], B0 = img[3 ^ px], B1 = img[4 ^ px], B2 = img[5 ^ px];
        const uint64_t X = win64(A0.x, A1.x, A2.x, off), Y = win64(A0.y, A1.y, A2.y, off), Z = win64(A0.z, A1.z, A2.z, off);
        const uint64_t S0 = win64(A0.w, A1.w, A2.w, off), S1 = win64(B0.x, B1.x, B2.x, off), S2 = win64(B0.y, B1.y, B2.y, off);
        const uint64_t m = bits_below((int)sz);
        const uint32_t o = (uint32_t)(k - x0);
        const int c1 = (int)(((X >> o) & 1) | ((Y >> o) & 1) << 1 | ((Z >> o) & 1) << 2);
        const int c2 = (int)(((S0 >> o) & 1) | ((S1 >> o) & 1) << 1 | ((S2 >> o) & 1) << 2);
        if (c1 < 1 || c1 > 4 || c2 < 1 || c2 > 4) { strag[1 + atomicAdd(strag, 1u)] = (uint32_t)row; st = PK_IDLE; continue; }   // the sequence ends within two bases, or an N
        const uint64_t lo = ~Z & m, hi = Z & ~Y & m;
        const uint64_t M0 = lo & ~Y & ~X, M1 = lo & ~Y & X, M2 = lo & Y & ~X, M3 = lo & Y & X, M4 = hi & ~X;
        const uint64_t Mc = c1 == 1 ? M1 : c1 == 2 ? M2 : c1 == 3 ? M3 : M4;
        const uint64_t lo2 = ~S2 & Mc, hi2 = S2 & ~S1 & Mc;
        const uint64_t N0 = lo2 & ~S1 & ~S0, N1 = lo2 & ~S1 & S0, N2 = lo2 & S1 & ~S0, N3 = lo2 & S1 & S0, N4 = hi2 & ~S0;
        const uint64_t Mp = c2 == 1 ? N1 : c2 == 2 ? N2 : c2 == 3 ? N3 : N4;
        uint32_t before = (uint32_t)__popcll(M0) + (uint32_t)__popcll(N0);          // '$' sorts before every base
        if (c1 != 4) before += (uint32_t)__popcll(M4);
        if (c1 == 2 || c1 == 1) before += (uint32_t)__popcll(M3);
        if (c1 == 1) before += (uint32_t)__popcll(M2);
        if (c2 != 4) before += (uint32_t)__popcll(N4);
        if (c2 == 2 || c2 == 1) before += (uint32_t)__popcll(N3);
        if (c2 == 1) before += (uint32_t)__popcll(N2);
        // pairs (c1, c2) before x0: the superblock's (+ K2), the block's 28-bit count, positions [0, off) of the block's own chunk
        const uint32_t e0x = (c1 & 1) ? 0u : ~0u, e0y = (c1 & 2) ? 0u : ~0u, e0z = (c1 & 4) ? 0u : ~0u;
        const uint32_t e1x = (c2 & 1) ? 0u : ~0u, e1y = (c2 & 2) ? 0u : ~0u, e1z = (c2 & 4) ? 0u : ~0u;
        const uint32_t pm0 = (A0.x ^ e0x) & (A0.y ^ e0y) & (A0.z ^ e0z) & (A0.w ^ e1x) & (B0.x ^ e1y) & (B0.y ^ e1z);
        const uint32_t nb_ = (uint32_t)__builtin_popcount(pm0 & fmd_mask32((int)off));
        const int pr = 4 * (c1 - 1) + (c2 - 1), bp = 28 * pr, tw = bp >> 5, tw1 = tw < 13 ? tw + 1 : 13;
#define WP_CW(t) iw[(((t) < 6 ? 3 + ((t) >> 1) : 6 + (((t) - 6) >> 2)) ^ px) * 4 + ((t) < 6 ? 2 + ((t) & 1) : (((t) - 6) & 3))]
        const uint32_t cwl = WP_CW(tw), cwh = WP_CW(tw1);
#undef WP_CW
        const uint32_t rel = __builtin_amdgcn_alignbit(cwh, cwl, (uint32_t)bp & 31u) & 0x0fffffffu;
        const uint64_t base = ix.pair_tab[(x0 >> (5 + FMD_PAIR_SB_SHIFT)) * 16 + (uint64_t)pr];
        const uint64_t nx0 = base + rel + nb_;
        k = nx0 + (uint64_t)__popcll(Mp & bits_below((int)o + 1)) - 1;
        x0 = nx0; sz = (uint64_t)__popcll(Mp); x1 += before;
        {   // the two bases into the parked nibbles (FmdWalkPark::bases: 4 bits each, the sequence's last base first)
            const uint32_t v1 = (uint32_t)c1 << (4 * (depth & 7)), w1 = depth >> 3;
            pk0 |= w1 == 0 ? v1 : 0u; pk1 |= w1 == 1 ? v1 : 0u; pk2 |= w1 == 2 ? v1 : 0u; pk3 |= w1 == 3 ? v1 : 0u;
            const uint32_t d2 = depth + 1, v2 = (uint32_t)c2 << (4 * (d2 & 7)), w2 = d2 >> 3;
            pk0 |= w2 == 0 ? v2 : 0u; pk1 |= w2 == 1 ? v2 : 0u; pk2 |= w2 == 2 ? v2 : 0u; pk3 |= w2 == 3 ? v2 : 0u;
            depth += 2;
        }
        if (depth >= FMD_WALK_SPLIT) {   // parked for good: the line WALK_HEAD would have written
            uint4 *pp = (uint4 *)(park + row);
            pp[0] = make_uint4((uint32_t)k, (uint32_t)(k >> 32), (uint32_t)x0, (uint32_t)(x0 >> 32));
            pp[1] = make_uint4((uint32_t)x1, (uint32_t)(x1 >> 32), (uint32_t)sz, (uint32_t)(sz >> 32));
            pp[2] = make_uint4(pk0, pk1, pk2, pk3); pp[3] = make_uint4(0, 0, 0, 0);
            st = PK_IDLE;
        } else if (sz > 63 || depth + 2 > FMD_WALK_SPLIT) { strag[1 + atomicAdd(strag, 1u)] = (uint32_t)row; st = PK_IDLE; }   // (cannot happen from an even depth with a narrow interval: sizes only shrink)
    }
}
// the admission records of the strands k_ovl_pair could not take all the way, for a second launch of WALK_HEAD
__global__ void k_ovl_strag_adm(const uint32_t *__restrict__ strag, const uint4 *__restrict__ adm, uint4 *__restrict__ adm2)
{
    const uint32_t n = strag[0];
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x) {
        const size_t r = strag[1 + j];
        adm2[2 * (size_t)j] = adm[2 * r]; adm2[2 * (size_t)j + 1] = adm[2 * r + 1];
    }
}

// WALK_TAIL2's strands with an N (redo[0] of them, slots redo[1 ..]): one lane per strand, the row byte by byte -- the 32 bases WALK_HEAD parked, then LF steps from
// the parked row on, read straight from the index (no wave gather: a handful of strands per batch of real reads, none of synthetic ones).
__global__ void k_ovl_seq_redo(FmdIndexView ix, const uint32_t *__restrict__ redo, const uint32_t *__restrict__ gidx, const FmdWalkPark *__restrict__ park,
                               const fmd_ovlp_rec_t *__restrict__ rec, uint8_t *__restrict__ seq_out, uint32_t seq_stride)
{
    const uint32_t n = redo[0];
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x) {
        const size_t g = gidx[redo[1 + j]];
        const int len = rec[g].len;
        uint8_t *dst = seq_out + g * (size_t)seq_stride;
        const uint4 *pp = (const uint4 *)(park + g);
        const uint4 a = pp[0], cb = pp[2];
        const uint32_t nib[4] = {cb.x, cb.y, cb.z, cb.w};
        uint64_t k = (uint64_t)a.y << 32 | a.x;
        for (int f = 0; f < (int)FMD_WALK_SPLIT && f < len; ++f) dst[len - 1 - f] = (uint8_t)((nib[f >> 3] >> (4 * (f & 7))) & 0xfu);
        for (int f = (int)FMD_WALK_SPLIT; f < len; ++f) {
            uint32_t b, o;
            fmd_split(k, b, o);
            const uint4 *img = ix.blocks + (size_t)b * FMD_BLK_U4;
            const uint4 v = img[o >> 5];
            const uint32_t bit = o & 31;
            const int c = (int)(((v.x >> bit) & 1) | ((v.y >> bit) & 1) << 1 | ((v.z >> bit) & 1) << 2);
            dst[len - 1 - f] = (uint8_t)c;
            k = ix.cnt[c] + fmd_block_rank1(img, 0, o + 1, c, b) - 1;
        }
    }
}

// The caller's copy of every sequence in read order: the stash holds it last base first.  One thread per 16 output bytes (four aligned
// dwords of the stash, a fifth when the chunk starts between two, funnel-shifted and byte-swapped; the record's length is read once per
// chunk, not once per word: 2.8 -> 2.3 ms per 2*10^7 strands of 100 bases in rows scattered by the sorted job); the chunk that holds the sequence's last
// bytes, and rows too short for a whole chunk, go word by word as before.  Same conditions under which a record describes a complete
// sequence (k_ovl_walk: not empty, not longer than max_len, longer than min_match unless info_only).  Bytes of a row beyond the sequence
// are written only inside the word that holds its last base (zeros), as before.
__device__ __forceinline__ void seq_out_word(const uint8_t *sr, int L, uint32_t w, uint8_t *dst)
{
    // output bytes 4w..4w+3 = stash bytes a+3..a with a = L - 4 - 4w: an unaligned word, byte-swapped
    const int a = L - 4 - (int)(4 * w);
    uint32_t v;
    if (a >= 0) {
        const uint32_t *q = (const uint32_t *)(sr + (a & ~3));
        const uint64_t two = (a & 3) ? ((uint64_t)q[1] << 32 | q[0]) : q[0];
        v = __builtin_bswap32((uint32_t)(two >> (8 * (a & 3))));
    } else v = __builtin_bswap32(*(const uint32_t *)sr << (8 * -a)); // the first 4 + a bases of the stash, the rest of the word zero
    *(uint32_t *)(dst + 4 * w) = v;
}
__global__ void k_ovl_seq_out_words(size_t n, uint32_t words, const uint8_t *__restrict__ srev, uint32_t stride_r, const fmd_ovlp_rec_t *__restrict__ rec,
                                    int min_match, int info_only, uint8_t *__restrict__ seq_out, uint32_t seq_stride, const uint32_t *__restrict__ gidx)
{
    const size_t total = n * (size_t)words, step = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += step) {
        const size_t sid = i / words;
        const uint32_t w = (uint32_t)(i - sid * words);
        const size_t g = gidx ? (size_t)gidx[sid] : sid;
        const int L = rec[g].len;
        if (L <= 0 || (uint32_t)L > stride_r || (!info_only && L <= min_match) || (int)(4 * w) >= L || 4 * w + 3 >= seq_stride) continue;
        seq_out_word(srev + sid * (size_t)stride_r, L, w, seq_out + g * (size_t)seq_stride);
    }
}
__global__ void k_ovl_seq_out(size_t n, uint32_t chunks, const uint8_t *__restrict__ srev, uint32_t stride_r, const fmd_ovlp_rec_t *__restrict__ rec,
                              int min_match, int info_only, uint8_t *__restrict__ seq_out, uint32_t seq_stride, const uint32_t *__restrict__ gidx)
{
    const size_t total = n * (size_t)chunks, step = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += step) {
        const size_t sid = i / chunks;
        const uint32_t c = (uint32_t)(i - sid * chunks), o = 16 * c;
        const size_t g = gidx ? (size_t)gidx[sid] : sid;   // the strand's row in rec[] and seq_out[] (sorted batches: fmd_ovlp_sorted_dev)
        const int L = rec[g].len;
        if (L <= 0 || (uint32_t)L > stride_r || (!info_only && L <= min_match) || (int)o >= L) continue;
        const uint8_t *sr = srev + sid * (size_t)stride_r;
        uint8_t *dst = seq_out + g * (size_t)seq_stride;
        if ((int)o + 16 <= L && o + 16 <= seq_stride) {   // a whole chunk inside the sequence: stash bytes [a, a + 16), a >= 0
            const int a = L - 16 - (int)o;
            const uint32_t sh = 8 * ((uint32_t)a & 3);
            const uint32_t *q = (const uint32_t *)(sr + (a & ~3));
            const uint32_t q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3], q4 = sh ? q[4] : 0u;   // (q[4] holds stash bytes below a + 16 <= L: inside the row)
            const uint32_t d0 = sh ? (uint32_t)(((uint64_t)q1 << 32 | q0) >> sh) : q0, d1 = sh ? (uint32_t)(((uint64_t)q2 << 32 | q1) >> sh) : q1;
            const uint32_t d2 = sh ? (uint32_t)(((uint64_t)q3 << 32 | q2) >> sh) : q2, d3 = sh ? (uint32_t)(((uint64_t)q4 << 32 | q3) >> sh) : q3;
            uint32_t *out = (uint32_t *)(dst + o);
            out[0] = __builtin_bswap32(d3); out[1] = __builtin_bswap32(d2); out[2] = __builtin_bswap32(d1); out[3] = __builtin_bswap32(d0);
        } else {
#pragma unroll
            for (uint32_t w = 4 * c; w < 4 * c + 4; ++w)
                if ((int)(4 * w) < L && 4 * w + 3 < seq_stride) seq_out_word(sr, L, w, dst);
        }
    }
}

// --------------------------------------------------------------------- phase B: fm6_get_nei
// unitig.c:93-179.  Latency is the enemy here (three dependent rank2a per candidate interval), so
// nothing but the rank fetch is allowed on a wave step's critical path:
//   * the next candidate interval is prefetched into registers while the current one is extended
//     (its load completes under the same s_waitcnt as the rank-block gather);
//   * the first child pushed in a round and the first neighbour stay in registers;
//   * categories (unitig.c:143-151) are assigned while pushing, because children are pushed in
//     sorted order unless a category forks; the stored entry keeps the original sort key in the
//     spare top 16 bits of its size word, and only a round that saw an out-of-order push takes
//     the slow path (sort + recompute), exactly as ks_introsort + the recompute loop would.
enum { ST_IDLE = 0, ST_PICK, ST_EXT, ST_E0, ST_C, ST_FIX1, ST_FIX2 };

struct I3 { uint64_t x0, x1, sz; };
// field-wise select (keeps the candidates in registers; a struct ternary chain goes through scratch)
__device__ __forceinline__ I3 pick5(int c, const I3 &a0, const I3 &a1, const I3 &a2, const I3 &a3, const I3 &a4)
{
    I3 r;
    r.x0 = sel6(c, a0.x0, a1.x0, a2.x0, a3.x0, a4.x0, a0.x0);
    r.x1 = sel6(c, a0.x1, a1.x1, a2.x1, a3.x1, a4.x1, a0.x1);
    r.sz = sel6(c, a0.sz, a1.sz, a2.sz, a3.sz, a4.sz, a0.sz);
    return r;
}


__global__ __launch_bounds__(64) void k_ovl_nei(FmdIndexView ix, size_t n, int min_match, const uint8_t *__restrict__ srev,
                                                uint32_t stride_r, uint32_t cap, fmd_intv_t *__restrict__ listA,
                                                fmd_intv_t *__restrict__ listB, fmd_ovlp_rec_t *__restrict__ rec,
                                                fmd_intv_t *__restrict__ nei_out, uint32_t max_nei,
                                                uint8_t *__restrict__ seq_out, uint32_t seq_stride, uint32_t *__restrict__ queue,
                                                const uint32_t *__restrict__ work_list, const uint32_t *__restrict__ work_n,
                                                const uint32_t *__restrict__ gidx)
{
    FMD_DECLARE_WAVE_LDS();
    if (work_list) n = *work_n;   // only the strands the group kernels could not take
    // per-lane search state
    size_t sid = 0;   // the strand's slot in the batch (rows of srev, listA, listB)
    size_t gs = 0;    // its row in rec[], nei_out[], seq_out[] (gidx[sid] in a sorted batch, sid otherwise)
    int st = ST_IDLE, ori_l = 0, cur_l = 0, cpend = 0, first_c = 0, masked_cat = -2, cat_j = 0, fix_i = 0;
    uint32_t prev_n = 0, curr_n = 0, j = 0, n_nei = 0, flags = 0, cat0 = 0, last_hi = 0;
    bool unsorted = false, exhausted = false, prev_is_a = true, e_valid = false;
    fmd_intv_t *prev = nullptr, *curr = nullptr;
    uint64_t last_key = 0;
    uint4 ea = make_uint4(0, 0, 0, 0), eb = make_uint4(0, 0, 0, 0); // prefetched prev[j], raw (decoded at pick time)
    uint64_t fx0 = 0, fx1 = 0, fsz = 0, finfo = 0;   // first child pushed this round (= next round's prev[0])
    uint64_t px0 = 0, px1 = 0, psz = 0, pinfo = 0;   // interval being extended
    I3 o0 = {0, 0, 0}, oc1 = {0, 0, 0}, oc2 = {0, 0, 0}, oc3 = {0, 0, 0}, oc4 = {0, 0, 0}; // its children
    uint64_t nx0 = 0, nsz = 0, ninfo = 0;             // first neighbour

    FmdTickets tk_;
    fmd_tickets_init(tk_, queue);
    for (;;) {
        // ---- refill
        const size_t my = fmd_tickets_take(tk_, queue, st == ST_IDLE && !exhausted);
        if (st == ST_IDLE && !exhausted) {
            if (my < n) {
                const size_t strand = work_list ? (size_t)work_list[my] : my;
                const size_t grow = gidx ? (size_t)gidx[strand] : strand;
                const fmd_ovlp_rec_t *o = rec + grow;
                if (o->status == 0 && o->n_ovlp > 0 && !(o->flags & FMD_OVLP_F_OVERFLOW)) {
                    sid = strand; gs = grow; ori_l = cur_l = o->len;
                    prev_n = (uint32_t)o->n_ovlp; curr_n = 0; j = 0;
                    prev = listA + sid * (size_t)cap + (cap - prev_n);
                    curr = listB + sid * (size_t)cap; prev_is_a = true;
                    n_nei = 0; flags = 0; masked_cat = -2; unsorted = false; last_key = 0; cat0 = 0;
                    e_valid = false;
                    st = ST_PICK;
                }
            } else exhausted = true;
        }
        // ---- bookkeeping that needs no rank: pick the next interval / finish a round / finish
        while (st == ST_PICK) {
            if (j < prev_n) {
                if (!e_valid) { const uint4 *q = (const uint4 *)(prev + j); ea = q[0]; eb = q[1]; }
                if (cur_l == ori_l) { // round 0: the walk's candidates (either form); it stored the suffix depth, unitig.c:53 wants the start
                    const FmdCand cd = cand_decode(ea, eb);
                    cat_j = 0;
                    px0 = cd.x0; px1 = cd.x1; psz = cd.sz; pinfo = (uint64_t)ori_l - cd.depth;
                } else {
                    cat_j = (int)(eb.w >> 4);                    // info >> 36
                    px0 = (uint64_t)ea.y << 32 | ea.x; px1 = (uint64_t)ea.w << 32 | ea.z;
                    psz = ((uint64_t)eb.y << 32 | eb.x) & FMD_SZ_MASK; pinfo = (uint64_t)eb.w << 32 | eb.z;
                }
                if (cat_j == masked_cat) { ++j; e_valid = false; continue; }
                st = ST_EXT;
                e_valid = j + 1 < prev_n;
                if (e_valid) { const uint4 *q = (const uint4 *)(prev + j + 1); ea = q[0]; eb = q[1]; } // lands under the rank fetch
            } else if (curr_n) { // end of a round (unitig.c:137-153)
                if ((uint32_t)cur_l < seq_stride) seq_out[gs * (size_t)seq_stride + cur_l] = (uint8_t)comp6(first_c);
                ++cur_l;
                if (unsorted) { // slow path: ks_introsort by the original keys, then recompute the categories
                    for (uint32_t a = 1; a < curr_n; ++a) {
                        uint64_t ax0, ax1, asz, ainf;
                        load_entry(curr + a, ax0, ax1, asz, ainf);
                        const uint64_t akey = (asz >> 48) << 32 | (ainf & 0xffffffffull);
                        uint32_t b = a;
                        while (b > 0) {
                            uint64_t bx0, bx1, bsz, binf;
                            load_entry(curr + b - 1, bx0, bx1, bsz, binf);
                            if (((bsz >> 48) << 32 | (binf & 0xffffffffull)) <= akey) break;
                            store_entry(curr + b, bx0, bx1, bsz, binf);
                            --b;
                        }
                        store_entry(curr + b, ax0, ax1, asz, ainf);
                    }
                    uint32_t last = 0; cat0 = 0;
                    for (uint32_t a = 0; a < curr_n; ++a) {
                        uint64_t ax0, ax1, asz, ainf;
                        load_entry(curr + a, ax0, ax1, asz, ainf);
                        const uint32_t hi = (uint32_t)(asz >> 48);
                        if (a == 0) last = hi; else if (hi != last) { last = hi; cat0 = a; }
                        ainf = (ainf & 0xffffffffull) | (uint64_t)cat0 << 36;
                        curr[a].info = ainf;
                        if (a == 0) { fx0 = ax0; fx1 = ax1; fsz = asz & FMD_SZ_MASK; finfo = ainf; }
                    }
                }
                if (cat0 != 0) flags |= FMD_OVLP_F_FORKED;
                prev_is_a = !prev_is_a; // both lists start at index 0 of their areas from now on
                prev = (prev_is_a ? listA : listB) + sid * (size_t)cap;
                curr = (prev_is_a ? listB : listA) + sid * (size_t)cap;
                prev_n = curr_n; curr_n = 0; j = 0; masked_cat = -2; unsorted = false; last_key = 0; cat0 = 0;
                ea = make_uint4((uint32_t)fx0, (uint32_t)(fx0 >> 32), (uint32_t)fx1, (uint32_t)(fx1 >> 32));
                eb = make_uint4((uint32_t)fsz, (uint32_t)(fsz >> 32), (uint32_t)finfo, (uint32_t)(finfo >> 32)); e_valid = true;
            } else { // all paths closed (unitig.c:154-178)
                fmd_ovlp_rec_t *o = rec + gs;
                const int rbeg = ori_l - (int)(uint32_t)ninfo;
                if (n_nei == 1 && (flags & FMD_OVLP_F_FORKED) && !(flags & FMD_OVLP_F_FIXED) && rbeg < ori_l) {
                    // contained reads made a fake fork: re-derive the appended bases (unitig.c:158-176)
                    o0.x0 = 0; o0.x1 = 0; o0.sz = ix.cnt[1]; // fm6_set_intv(e, 0, ok0)
                    fix_i = rbeg;
                    st = ST_FIX1;
                    break;
                }
                if (n_nei > 1) cur_l = ori_l;
                o->rbeg = n_nei ? rbeg : -1;
                o->ext_len = cur_l - ori_l; o->n_nei = (int32_t)n_nei; o->flags |= flags;
                st = ST_IDLE;
            }
        }
        if (__ballot(st != ST_IDLE) == 0) { if (__ballot(!exhausted) == 0) break; else continue; }

        // ---- one rank2a request per lane
        uint64_t qk = NONE64, ql = NONE64;
        if (st == ST_EXT) { qk = px1 - 1; ql = px1 - 1 + psz; }                    // forward: strand x[1]
        else if (st == ST_E0) { qk = o0.x0 - 1; ql = o0.x0 - 1 + o0.sz; }          // backward: strand x[0]
        else if (st == ST_C) {
            const uint64_t a = cpend == 1 ? oc1.x0 : cpend == 2 ? oc2.x0 : cpend == 3 ? oc3.x0 : oc4.x0;
            const uint64_t z = cpend == 1 ? oc1.sz : cpend == 2 ? oc2.sz : cpend == 3 ? oc3.sz : oc4.sz;
            qk = a - 1; ql = a - 1 + z;
        } else if (st == ST_FIX1 || st == ST_FIX2) { qk = o0.x1 - 1; ql = o0.x1 - 1 + o0.sz; }
        const FmdRank2 r = fmd_wave_rank2_fetch(ix, fmd_lds, qk, ql);

        // ---- consume
        if (st == ST_EXT || st == ST_FIX1 || st == ST_FIX2) {
            uint64_t tk[6] = {0, 0, 0, 0, 0, 0}, tl[6] = {0, 0, 0, 0, 0, 0};
            if (r.hk) fmd_block_rank6<false>(r.bk, r.t, r.nk, tk, r.blk_k);
            if (r.hl) fmd_block_rank6<false>(r.bl, r.tl, r.nl, tl, r.blk_l);
            uint64_t s[6];
#pragma unroll
            for (int c = 0; c < 6; ++c) s[c] = tl[c] - tk[c];
            // forward extension (exact.c:72-88, is_back = 0): x[1] from rank, x[0] running sum
            const uint64_t base0 = st == ST_EXT ? px0 : o0.x0;
            I3 k0, k1, k2, k3, k4;
            k0.x0 = base0;            k0.x1 = ix.cnt[0] + tk[0]; k0.sz = s[0];
            k4.x0 = k0.x0 + s[0];     k4.x1 = ix.cnt[4] + tk[4]; k4.sz = s[4];
            k3.x0 = k4.x0 + s[4];     k3.x1 = ix.cnt[3] + tk[3]; k3.sz = s[3];
            k2.x0 = k3.x0 + s[3];     k2.x1 = ix.cnt[2] + tk[2]; k2.sz = s[2];
            k1.x0 = k2.x0 + s[2];     k1.x1 = ix.cnt[1] + tk[1]; k1.sz = s[1];
            if (st == ST_EXT) {
                o0 = k0; oc1 = k1; oc2 = k2; oc3 = k3; oc4 = k4;
                if (o0.sz && cur_l != ori_l) st = ST_E0;   // some reads end here (unitig.c:111)
                else {
                    cpend = oc1.sz ? 1 : oc2.sz ? 2 : oc3.sz ? 3 : oc4.sz ? 4 : 0;
                    if (cpend) st = ST_C; else { ++j; st = ST_PICK; }
                }
            } else if (st == ST_FIX1) { // unitig.c:160-163
                const int b = seq_out[gs * (size_t)seq_stride + fix_i];   // (the caller's row holds the sequence in read order by now: the stash is the walk's own)
                const int c = comp6(b);
                o0 = pick5(c, k0, k1, k2, k3, k4);
                if (c == 5) { o0.x0 = k1.x0 + s[1]; o0.x1 = ix.cnt[5] + tk[5]; o0.sz = s[5]; }
                ++fix_i;
                if (fix_i == ori_l) { st = ori_l < cur_l ? ST_FIX2 : ST_PICK; flags |= FMD_OVLP_F_FIXED; }
            } else { // ST_FIX2: unitig.c:164-175
                int cnt_ok = 0, c0 = -1;
#define FMD_FIX_TRY(c, kc) if (kc.sz && kc.x0 <= nx0 && kc.x0 + kc.sz >= nx0 + nsz) { ++cnt_ok; c0 = c; }
                FMD_FIX_TRY(1, k1) FMD_FIX_TRY(2, k2) FMD_FIX_TRY(3, k3) FMD_FIX_TRY(4, k4)
#undef FMD_FIX_TRY
                bool stop = (cnt_ok == 0 && k0.sz != 0);
                if (!stop && c0 > 0) {
                    if ((uint32_t)fix_i < seq_stride) seq_out[gs * (size_t)seq_stride + fix_i] = (uint8_t)comp6(c0);
                    o0 = pick5(c0, k0, k1, k2, k3, k4);
                    ++fix_i;
                    if (fix_i == cur_l) stop = true;
                } else stop = true;
                if (stop) { cur_l = fix_i; st = ST_PICK; }
            }
        } else if (st == ST_E0 || st == ST_C) {
            // fm6_extend0 (exact.c:90-98), backward: only the '$' child matters
            const uint64_t t0k = r.hk ? fmd_block_rank1(r.bk, r.t, r.nk, 0, r.blk_k) : 0;
            const uint64_t t0l = r.hl ? fmd_block_rank1(r.bl, r.tl, r.nl, 0, r.blk_l) : 0;
            const uint64_t e0sz = t0l - t0k;
            if (st == ST_E0) {
                bool is_nei = false;
                if (e0sz && o0.sz == psz && psz == e0sz) { // bounded by sentinels on both sides and not contained
                    const uint64_t inf = (uint64_t)ori_l - (pinfo & 0xffffffffull);
                    if (n_nei == 0) { nx0 = t0k; nsz = e0sz; ninfo = inf; }
                    if (n_nei < max_nei) store_entry(nei_out + gs * (size_t)max_nei + n_nei, t0k, o0.x1, e0sz, inf);
                    else flags |= FMD_OVLP_F_OVERFLOW;
                    ++n_nei;
                    masked_cat = cat_j; // mask out the other intervals of this category
                    is_nei = true;
                }
                if (is_nei) { ++j; st = ST_PICK; }
                else {
                    cpend = oc1.sz ? 1 : oc2.sz ? 2 : oc3.sz ? 3 : oc4.sz ? 4 : 0;
                    if (cpend) st = ST_C; else { ++j; st = ST_PICK; }
                }
            } else {
                if (e0sz) { // left end bounded by a sentinel: keep the child (unitig.c:128-135)
                    const I3 ch = pick5(cpend, oc1, oc1, oc2, oc3, oc4);
                    const uint64_t key = (pinfo & 0xfffffff0ffffffffull) | (uint64_t)cpend << 32;
                    const uint32_t hi = (uint32_t)(key >> 32);  // old category << 4 | base
                    if (curr_n < cap) {
                        if (curr_n == 0) { first_c = cpend; cat0 = 0; last_hi = hi; }
                        else {
                            if (key < last_key) unsorted = true;
                            if (hi != last_hi) { cat0 = curr_n; last_hi = hi; }
                        }
                        last_key = key;
                        const uint64_t inf = (key & 0xffffffffull) | (uint64_t)cat0 << 36;
                        store_entry(curr + curr_n, ch.x0, ch.x1, ch.sz | (uint64_t)hi << 48, inf);
                        if (curr_n == 0) { fx0 = ch.x0; fx1 = ch.x1; fsz = ch.sz; finfo = inf; }
                        ++curr_n;
                    } else { flags |= FMD_OVLP_F_OVERFLOW; }
                }
                int nc = 0;
                if (cpend < 2 && oc2.sz) nc = 2; else if (cpend < 3 && oc3.sz) nc = 3; else if (cpend < 4 && oc4.sz) nc = 4;
                if (nc) cpend = nc; else { ++j; st = ST_PICK; }
            }
        }
        // an overflowing strand is abandoned; the host re-runs it with larger capacities
        if (st != ST_IDLE && (flags & FMD_OVLP_F_OVERFLOW)) {
            fmd_ovlp_rec_t *o = rec + gs;
            o->flags |= FMD_OVLP_F_OVERFLOW; o->n_nei = 0; o->rbeg = -1; o->ext_len = 0;
            st = ST_IDLE;
        }
    }
}

// ------------------------------------------------------------ the fake-fork fix-up on its own
// unitig.c:158-176 for strands the group kernels closed with ONE neighbour after a fork (contained reads made the fork): the record,
// the neighbour and the appended bases are there as fm6_get_nei's loop leaves them; what remains is to walk the overlap string
// forward from the empty interval (FIX1: ori_l - rbeg dependent steps) and then re-derive the appended bases as long as exactly one
// child still contains the neighbour's interval (FIX2).  k_ovl_nei does the same at the end of its own pass -- after redoing the
// whole of fm6_get_nei with lists in HBM, which is what this kernel spares the strands the group kernels had finished.
__global__ __launch_bounds__(64) void k_ovl_fix(FmdIndexView ix, const uint32_t *__restrict__ list, const uint32_t *__restrict__ list_n,
                                                const uint8_t *__restrict__ srev, uint32_t stride_r, fmd_ovlp_rec_t *__restrict__ rec,
                                                const fmd_intv_t *__restrict__ nei_out, uint32_t max_nei, uint8_t *__restrict__ seq_out, uint32_t seq_stride,
                                                uint32_t *__restrict__ queue, const uint32_t *__restrict__ gidx)
{
    FMD_DECLARE_WAVE_LDS();
    const size_t n = *list_n;
    size_t sid = 0, gs = 0;
    int st = 0, ori_l = 0, cur_l = 0, fix_i = 0;   // st: 0 idle, 1 = FIX1, 2 = FIX2
    uint64_t x0 = 0, x1 = 0, sz = 0, nx0 = 0, nsz = 0;
    bool exhausted = false;
    FmdTickets tk_;
    fmd_tickets_init(tk_, queue);
    for (;;) {
        const size_t my = fmd_tickets_take(tk_, queue, st == 0 && !exhausted);
        if (st == 0 && !exhausted) {
            if (my < n) {
                sid = list[my]; gs = gidx ? (size_t)gidx[sid] : sid;
                const fmd_ovlp_rec_t *o = rec + gs;
                ori_l = o->len; cur_l = ori_l + o->ext_len; fix_i = o->rbeg;
                const uint4 *q = (const uint4 *)(nei_out + gs * (size_t)max_nei);
                const uint4 a = q[0], b = q[1];
                nx0 = (uint64_t)a.y << 32 | a.x; nsz = (uint64_t)b.y << 32 | b.x;
                x0 = 0; x1 = 0; sz = ix.cnt[1];                      // fm6_set_intv(e, 0, ok0)
                if (fix_i >= 0 && fix_i < ori_l) st = 1;
            } else exhausted = true;
        }
        if (__ballot(st != 0) == 0) { if (__ballot(!exhausted) == 0) break; else continue; }
        const FmdRank2 r = fmd_wave_rank2_fetch(ix, fmd_lds, st ? x1 - 1 : NONE64, st ? x1 - 1 + sz : NONE64);   // forward: strand x[1]
        if (st == 0) continue;
        uint64_t tk[6] = {0, 0, 0, 0, 0, 0}, tl[6] = {0, 0, 0, 0, 0, 0}, s[6];
        if (r.hk) fmd_block_rank6<false>(r.bk, r.t, r.nk, tk, r.blk_k);
        if (r.hl) fmd_block_rank6<false>(r.bl, r.tl, r.nl, tl, r.blk_l);
#pragma unroll
        for (int c = 0; c < 6; ++c) s[c] = tl[c] - tk[c];
        // forward extension (exact.c:72-88, is_back = 0): x[1] from rank, x[0] running sum in the order $, T, G, C, A, N
        I3 k0, k1, k2, k3, k4;
        k0.x0 = x0;               k0.x1 = ix.cnt[0] + tk[0]; k0.sz = s[0];
        k4.x0 = k0.x0 + s[0];     k4.x1 = ix.cnt[4] + tk[4]; k4.sz = s[4];
        k3.x0 = k4.x0 + s[4];     k3.x1 = ix.cnt[3] + tk[3]; k3.sz = s[3];
        k2.x0 = k3.x0 + s[3];     k2.x1 = ix.cnt[2] + tk[2]; k2.sz = s[2];
        k1.x0 = k2.x0 + s[2];     k1.x1 = ix.cnt[1] + tk[1]; k1.sz = s[1];
        bool done = false;
        if (st == 1) { // unitig.c:160-163
            const int b = seq_out[gs * (size_t)seq_stride + fix_i];
            const int c = comp6(b);
            I3 n3 = pick5(c, k0, k1, k2, k3, k4);
            if (c == 5) { n3.x0 = k1.x0 + s[1]; n3.x1 = ix.cnt[5] + tk[5]; n3.sz = s[5]; }
            x0 = n3.x0; x1 = n3.x1; sz = n3.sz;
            ++fix_i;
            if (fix_i == ori_l) { if (ori_l < cur_l) st = 2; else done = true; }
        } else {       // unitig.c:164-175
            int cnt_ok = 0, c0 = -1;
#define FMD_FIX_TRY(c, kc) if (kc.sz && kc.x0 <= nx0 && kc.x0 + kc.sz >= nx0 + nsz) { ++cnt_ok; c0 = c; }
            FMD_FIX_TRY(1, k1) FMD_FIX_TRY(2, k2) FMD_FIX_TRY(3, k3) FMD_FIX_TRY(4, k4)
#undef FMD_FIX_TRY
            bool stop = (cnt_ok == 0 && k0.sz != 0);
            if (!stop && c0 > 0) {
                if ((uint32_t)fix_i < seq_stride) seq_out[gs * (size_t)seq_stride + fix_i] = (uint8_t)comp6(c0);
                const I3 n3 = pick5(c0, k0, k1, k2, k3, k4);
                x0 = n3.x0; x1 = n3.x1; sz = n3.sz;
                ++fix_i;
                if (fix_i == cur_l) stop = true;
            } else stop = true;
            if (stop) { cur_l = fix_i; done = true; }
        }
        if (done) {
            fmd_ovlp_rec_t *o = rec + gs;
            o->ext_len = cur_l - ori_l;
            o->flags |= FMD_OVLP_F_FIXED;
            st = 0;
        }
    }
}

// ------------------------------------------------------------ phase C: check_left_simple
// unitig.c:186-204 for the edge (strand -> its unique neighbour): collect, walking the neighbour
// forward from its first base, the reads that END inside it with >= min_match bases (its left
// neighbours), then pull them back over the strand's bases left of the overlap; any of them that
// neither ends nor continues with the strand's base is a backward bifurcation.  A pure function of
// the strand once its neighbour is unique, so the deterministic host walk reads it from the table:
// rec.reserved = 0 (check_left_simple returns 0), 1 (returns -1), 2 (not applicable).
enum { CL_IDLE = 0, CL_FWD, CL_PICK, CL_BWD };

__global__ __launch_bounds__(64) void k_ovl_cls(FmdIndexView ix, size_t n, int min_match, uint32_t cap, fmd_intv_t *__restrict__ listA,
                                                fmd_intv_t *__restrict__ listB, fmd_ovlp_rec_t *__restrict__ rec,
                                                const uint8_t *__restrict__ seq, uint32_t seq_stride, uint32_t *__restrict__ queue)
{
    FMD_DECLARE_WAVE_LDS();
    size_t sid = 0;
    int st = CL_IDLE, rbeg = 0, s_l = 0, depth = 0, i = 0;
    uint32_t prev_n = 0, curr_n = 0, j = 0;
    uint64_t x0 = 0, x1 = 0, sz = 0;
    fmd_intv_t *prev = nullptr, *curr = nullptr;
    const uint8_t *s = nullptr;
    bool exhausted = false;
    FmdTickets tk_;
    fmd_tickets_init(tk_, queue);
    for (;;) {
        const size_t my = fmd_tickets_take(tk_, queue, st == CL_IDLE && !exhausted);
        if (st == CL_IDLE && !exhausted) {
            if (my < n) {
                fmd_ovlp_rec_t *o = rec + my;
                if (o->reserved != 2) {}   // decided already (fmd_ovlp_link_dev): only the rows still open are looked at
                else if (o->status == 0 && o->n_nei == 1 && o->rbeg >= 0 && !(o->flags & FMD_OVLP_F_OVERFLOW) &&
                    (uint32_t)(o->len + o->ext_len) <= seq_stride) {
                    sid = my; rbeg = o->rbeg; s_l = o->len + o->ext_len;
                    s = seq + sid * (size_t)seq_stride;
                    const int c = s[rbeg];
                    x0 = ix.cnt[c]; x1 = ix.cnt[comp6(c)]; sz = ix.cnt[c + 1] - ix.cnt[c];
                    depth = 1; prev = listA + sid * (size_t)cap; curr = listB + sid * (size_t)cap; prev_n = curr_n = 0;
                    if (rbeg + 1 < s_l) st = CL_FWD;
                    else { o->reserved = 0; } // a one-base neighbour cannot collect anything
                } else o->reserved = 2;
            } else exhausted = true;
        }
        while (st == CL_PICK) {
            if (j < prev_n) { uint64_t inf; load_entry(prev + j, x0, x1, sz, inf); st = CL_BWD; }
            else { // next base to the left (unitig.c:194-202)
                fmd_intv_t *t = prev; prev = curr; curr = t;
                prev_n = curr_n; curr_n = 0; j = 0; --i;
                if (i < 0 || prev_n == 0) { rec[sid].reserved = 0; st = CL_IDLE; }
            }
        }
        if (__ballot(st != CL_IDLE) == 0) { if (__ballot(!exhausted) == 0) break; else continue; }
        uint64_t qk = NONE64, ql = NONE64;
        if (st == CL_FWD) { qk = x1 - 1; ql = x1 - 1 + sz; }
        else if (st == CL_BWD) { qk = x0 - 1; ql = x0 - 1 + sz; }
        const FmdRank2 r = fmd_wave_rank2_fetch(ix, fmd_lds, qk, ql);
        if (st == CL_IDLE) continue;
        // the symbol this step extends by: forward along the neighbour, or backward over the strand
        const int c = st == CL_FWD ? comp6(s[rbeg + depth]) : s[i];
        // Child sizes sc[], rank_c(k) and -- where a candidate may be pushed -- rank_$(k).  Narrow interval
        // (all but the first ~log4(n) forward steps): one 64-position window of the lane's block image(s)
        // and one or two single-symbol ranks instead of two six-symbol block ranks.
        uint64_t sc[6], tkc, tk0 = 0;
        if (sz <= 63) {
            const uint64_t a0 = st == CL_FWD ? x1 : x0;
            uint4 wa, wb, wc;
            grp_window(r.bk, r.t, r.bl, r.tl, r.blk_k, r.blk_l, r.hk, r.hl && r.blk_l != r.blk_k, r.blk_k, r.nk - 1, wa, wb, wc); // window at a0 = (a0 - 1) + 1
            const uint32_t sh = (uint32_t)a0 & 31;
            const uint64_t m = (1ull << (int)sz) - 1;
            const uint64_t X = win64(wa.x, wb.x, wc.x, sh), Y = win64(wa.y, wb.y, wc.y, sh), Z = win64(wa.z, wb.z, wc.z, sh);
            const uint64_t lo = ~Z & m, hi = Z & ~Y & m;
            sc[0] = __popcll(lo & ~Y & ~X); sc[1] = __popcll(lo & ~Y & X); sc[2] = __popcll(lo & Y & ~X); sc[3] = __popcll(lo & Y & X);
            sc[4] = __popcll(hi & ~X); sc[5] = __popcll(hi & X);
            tkc = fmd_block_rank1(r.bk, r.t, r.nk, c, r.blk_k);
            if (st == CL_FWD && depth >= min_match && sc[0]) tk0 = fmd_block_rank1(r.bk, r.t, r.nk, 0, r.blk_k);
        } else {
            uint64_t tk[6] = {0, 0, 0, 0, 0, 0}, tl[6] = {0, 0, 0, 0, 0, 0};
            if (r.hk) fmd_block_rank6<false>(r.bk, r.t, r.nk, tk, r.blk_k);
            if (r.hl) fmd_block_rank6<false>(r.bl, r.tl, r.nl, tl, r.blk_l);
#pragma unroll
            for (int a = 0; a < 6; ++a) sc[a] = tl[a] - tk[a];
            tkc = sel6(c, tk[0], tk[1], tk[2], tk[3], tk[4], tk[5]);
            tk0 = tk[0];
        }
        const uint64_t szc = sel6(c, sc[0], sc[1], sc[2], sc[3], sc[4], sc[5]);
        // coordinate of child c on the strand that is not extended: running sum in the order $,T,G,C,A,N (exact.c:81-86)
        uint64_t before = 0;
        if (c != 0) before += sc[0];
        if (c == 3 || c == 2 || c == 1 || c == 5) before += sc[4];
        if (c == 2 || c == 1 || c == 5) before += sc[3];
        if (c == 1 || c == 5) before += sc[2];
        if (c == 5) before += sc[1];
        const uint64_t nxc = sel6(c, ix.cnt[0], ix.cnt[1], ix.cnt[2], ix.cnt[3], ix.cnt[4], ix.cnt[5]) + tkc;
        if (st == CL_FWD) { // overlap_intv(at5 = 1, inc_sentinel = 1), unitig.c:38-64
            bool end_fwd = szc == 0;
            if (!end_fwd) {
                if (depth >= min_match && sc[0]) {
                    if (prev_n < cap) store_entry(prev + prev_n, x0, ix.cnt[0] + tk0, sc[0], 0);
                    ++prev_n;
                }
                x1 = nxc; x0 += before; sz = szc;   // ik = ok[c] (forward)
                ++depth;
                end_fwd = rbeg + depth == s_l;
            }
            if (end_fwd) {
                if (prev_n > cap) { rec[sid].flags |= FMD_OVLP_F_OVERFLOW; rec[sid].reserved = 2; st = CL_IDLE; }
                else if (prev_n == 0 || rbeg == 0) { rec[sid].reserved = 0; st = CL_IDLE; }
                else { i = rbeg - 1; j = 0; curr_n = 0; st = CL_PICK; }
            }
        } else { // CL_BWD: one collected interval against base s[i] (unitig.c:196-200)
            if (sc[0] + szc != sz) { rec[sid].reserved = 1; st = CL_IDLE; } // potential backward bifurcation
            else {
                if (curr_n < cap) store_entry(curr + curr_n, nxc, x1 + before, szc, 0);
                ++curr_n; ++j;
                st = CL_PICK;
            }
        }
    }
}

// ------------------------------------------------------------------------------- host entry
static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
// tickets per atomic of a walk launch (FmdTickets::chunk): `dflt` for large launches, never more than a 64th of a wave's share
// (the last chunks of a launch are worked off by fewer and fewer waves); the environment variable is the A/B knob
static int walk_gate(const char *env, int dflt) { const char *e = getenv(env); int g = e ? atoi(e) : dflt; return g < 0 ? 0 : g > 64 ? 64 : g; }   // lanes that must be idle before any takes a new strand (k_ovl_walk)
static uint32_t walk_ticket_chunk(const char *env, uint32_t dflt, size_t n, int grid)
{
    const char *e = getenv(env);
    uint32_t c = e && atoi(e) > 0 ? (uint32_t)atoi(e) : dflt;
    const size_t share = n / (size_t)(grid > 0 ? grid : 1) / 64;
    if (!e && c > share) c = (uint32_t)share;
    if (c < FMD_TICKET_CHUNK) c = FMD_TICKET_CHUNK;
    if (!getenv("FMD_TICKETS_FIXED")) c |= 1u << 24;   // guided chunks (fmd_wave.h); FMD_TICKETS_FIXED: the A/B switch
    return c;
}
static void launch_seq_out(hipStream_t st, size_t n, uint32_t max_len, const uint8_t *srev, uint32_t stride_r, const fmd_ovlp_rec_t *rec, int min_match,
                           int info_only, uint8_t *seq_out, uint32_t seq_stride, const uint32_t *gidx = nullptr)
{
    if (getenv("FMD_SEQ_OUT_WORDS")) {   // A/B switch: one thread per output word (round 3's kernel)
        const uint32_t words = (max_len + 3) / 4;
        size_t blocks = (n * (size_t)words + 255) / 256;
        if (blocks > (1u << 22)) blocks = 1u << 22;
        k_ovl_seq_out_words<<<(unsigned)blocks, 256, 0, st>>>(n, words, srev, stride_r, rec, min_match, info_only, seq_out, seq_stride, gidx);
        return;
    }
    const uint32_t chunks = (max_len + 15) / 16;
    const size_t total = n * (size_t)chunks;
    size_t blocks = (total + 255) / 256;
    if (blocks > (1u << 22)) blocks = 1u << 22;
    k_ovl_seq_out<<<(unsigned)blocks, 256, 0, st>>>(n, chunks, srev, stride_r, rec, min_match, info_only, seq_out, seq_stride, gidx);
}

extern "C" size_t fmd_ovlp_work_bytes(size_t n, uint32_t max_len, int min_match)
{
    const size_t stride_r = align_up((size_t)max_len, 16);
    const size_t cap = fmd_ovlp_list_cap(max_len, min_match);
    return align_up(n * stride_r, 256) + 2 * align_up(n * cap * sizeof(fmd_intv_t), 256) +
           align_up(n * (4 * FMD_CLS_WORDS_PER_STRAND) + 4 * (size_t)FMD_CLS_PART_U32 * FMD_OVLP_MAX_PARTS, 256) + 256;
}

// The buffers of one fmd_ovlp_dev call; the two phases below work on the strands [b, b + np) of it.
struct OvlBatch {
    fmd_dev *h; FmdIndexView ix;
    const uint64_t *ids; int min_match; uint32_t max_len, max_nei, stride_r, cap, seq_stride;
    uint8_t *srev; fmd_intv_t *listA, *listB; uint32_t *cls;
    fmd_ovlp_rec_t *rec; fmd_intv_t *nei; uint8_t *seq;
    // a batch taken from the sorted order of a larger job (fmd_ovlp_sorted_dev): slot t of the batch is row gidx[t] of rec, nei, seq and of
    // park, where WALK_HEAD left it; nullptr: slot = row, the walk starts at the sentinel of ids[t]
    const uint32_t *gidx; FmdWalkPark *park;
};

// The second pass of a sorted job through k_ovl_walk<WALK_TAIL2> (rows written by the walk, FMD_WALK_TAIL2=0: the A/B switch), and with the work
// lists of get_nei made by the walk as well (FMD_WALK_CLS=0: k_ovl_classify behind it as before)?
static bool ovl_tail2(const OvlBatch &o)
{
    const char *e = getenv("FMD_WALK_TAIL2");
    return o.gidx && o.stride_r <= WALK_LS_BASES && o.seq_stride >= o.stride_r + 4 && (o.seq_stride & 3) == 0 && !(e && atoi(e) == 0);
}
static bool ovl_tail2_cls(const OvlBatch &o)
{
    const char *e = getenv("FMD_WALK_CLS");
    return ovl_tail2(o) && !getenv("FMD_OVLP_SLOW_ONLY") && !(e && atoi(e) == 0);
}
static int ovl_use_fast(void) { const char *ef = getenv("FMD_OVLP_FAST"); return fmd_nei_fast_available() && !(ef && atoi(ef) == 0); }   // FMD_OVLP_FAST=0: A/B switch, every strand through the general group kernels
static int ovl_grp_down(void) { const char *e = getenv("FMD_GRP_DOWN"); return !(e && atoi(e) == 0); }
static int ovl_min_cls(void) { const char *e = getenv("FMD_GRP4"); return e && atoi(e) == 0 ? 1 : 0; }                                  // FMD_GRP4=0: no groups of 4 (as fmd_launch_classify)

// phase A: LF-walk + overlap_intv + fm6_is_contained, then the read-order copy.  per_cu > 0 bounds the
// resident waves per CU (pipelined batches leave room for phase B of the previous part).
static void ovl_phase_a(const OvlBatch &o, hipStream_t st, size_t b, size_t np, int per_cu)
{
    uint8_t *srev = o.srev + b * (size_t)o.stride_r;
    fmd_intv_t *listA = o.listA + b * (size_t)o.cap;
    uint8_t *seq = o.gidx ? o.seq : o.seq + b * (size_t)o.seq_stride;
    uint32_t *q0 = fmd_next_queue(o.h, st);
    if (o.gidx) {   // the second pass of a sorted job: every strand of the batch from where WALK_HEAD parked it
        int grid = fmd_grid_for_lds(o.h, np, FMD_COMPACT_LDS_U4 * 16);
        { const char *e = getenv("FMD_WALK_WAVES"); if (per_cu <= 0 && e && atoi(e) > 0) per_cu = atoi(e); }
        if (per_cu > 0 && grid > o.h->n_cu * per_cu) grid = o.h->n_cu * per_cu;
        // sequences of at most WALK_LS_BASES bases: the rows in read order come from the walk itself (WALK_TAIL2: bases in LDS, no stash in HBM, no k_ovl_seq_out);
        // FMD_WALK_TAIL2=0 is the A/B switch.  The stash area, idle then, holds the list of the strands with an N: [0] their number, [1 ..] their slots.
        if (ovl_tail2(o)) {
            uint32_t *redo = (uint32_t *)srev;
            (void)hipMemsetAsync(redo, 0, 4, st);
            uint32_t *cls = nullptr;      // the work lists of phase B made here (part 0 of the batch: a sorted job's batches are not pipelined)
            if (ovl_tail2_cls(o)) { cls = o.cls + b * FMD_CLS_WORDS_PER_STRAND; (void)hipMemsetAsync(cls, 0, 4 * FMD_CLS_HEADER_U32, st); }
            grid = fmd_grid_for_lds(o.h, np, FMD_COMPACT_LDS_U4 * 16 + 64 * WALK_LS_WORDS * 4);
            if (per_cu > 0 && grid > o.h->n_cu * per_cu) grid = o.h->n_cu * per_cu;
            k_ovl_walk<WALK_TAIL2><<<grid, 64, 0, st>>>(o.ix, np, nullptr, o.min_match, srev, o.stride_r, o.cap, listA, o.rec, seq, o.seq_stride, q0, 0, o.park, o.gidx + b, nullptr, walk_ticket_chunk("FMD_TAIL_TICKETS", 64, np, grid), redo, cls, ovl_use_fast() | ovl_min_cls() << 8 | walk_gate("FMD_TAIL_GATE", 0) << 16);
            k_ovl_seq_redo<<<64, 64, 0, st>>>(o.ix, redo, o.gidx + b, o.park, o.rec, seq, o.seq_stride);
            return;
        }
        k_ovl_walk<WALK_TAIL><<<grid, 64, 0, st>>>(o.ix, np, nullptr, o.min_match, srev, o.stride_r, o.cap, listA, o.rec, seq, o.seq_stride, q0, 0, o.park, o.gidx + b, nullptr, walk_ticket_chunk("FMD_TAIL_TICKETS", 64, np, grid), nullptr, nullptr, 0);
        launch_seq_out(st, np, o.max_len, srev, o.stride_r, o.rec, o.min_match, 0, seq, o.seq_stride, o.gidx + b);
        return;
    }
    int grid = fmd_grid_for_lds(o.h, np, FMD_COMPACT_LDS_U4 * 16);
    { const char *e = getenv("FMD_WALK_WAVES"); if (per_cu <= 0 && e && atoi(e) > 0) per_cu = atoi(e); }   // A/B knob: resident waves per CU
    if (per_cu > 0 && grid > o.h->n_cu * per_cu) grid = o.h->n_cu * per_cu;
    k_ovl_walk<WALK_WHOLE><<<grid, 64, 0, st>>>(o.ix, np, o.ids + b, o.min_match, srev, o.stride_r, o.cap, listA, o.rec + b, seq, o.seq_stride, q0, 0, nullptr, nullptr, nullptr, walk_ticket_chunk("FMD_WALK_TICKETS", 64, np, grid), nullptr, nullptr, 0);
    launch_seq_out(st, np, o.max_len, srev, o.stride_r, o.rec + b, o.min_match, 0, seq, o.seq_stride);
}

static bool ovl_slow_acquire(fmd_dev *h);
// phase B: fm6_get_nei.  `part` selects the counter header of this part's work lists.
static int ovl_phase_b(const OvlBatch &o, hipStream_t st, size_t b, size_t np, int part, int per_cu, int fast_cu)
{
    uint8_t *srev = o.srev + b * (size_t)o.stride_r;
    fmd_intv_t *listA = o.listA + b * (size_t)o.cap, *listB = o.listB + b * (size_t)o.cap;
    const uint32_t *gidx = o.gidx ? o.gidx + b : nullptr;
    fmd_ovlp_rec_t *rec = gidx ? o.rec : o.rec + b;
    fmd_intv_t *nei = gidx ? o.nei : o.nei + b * (size_t)o.max_nei;
    uint8_t *seq = gidx ? o.seq : o.seq + b * (size_t)o.seq_stride;
    const int grid = fmd_grid_for(o.h, np);
    uint32_t *q2 = fmd_next_queue(o.h, st);
    if (getenv("FMD_OVLP_SLOW_ONLY")) { // A/B switch: everything through the lane-per-strand kernel
        k_ovl_nei<<<grid, 64, 0, st>>>(o.ix, np, o.min_match, srev, o.stride_r, o.cap, listA, listB, rec, nei, o.max_nei, seq, o.seq_stride, q2, nullptr, nullptr, gidx);
        return FMD_OK;
    }
    // work lists: the counter header, then one list per group class (2 words per strand), the slow list (1), one list per fast class (2)
    // (a general list has room for every strand of the part + the holes the fast kernels' chunked hand-over may leave)
    uint32_t *cls = o.cls + (size_t)part * FMD_CLS_PART_U32 + b * FMD_CLS_WORDS_PER_STRAND;
    FmdOvlClasses cl;
    cl.cnt = cls;
    for (int k = 0; k < FMD_GRP_CLASSES; ++k) cl.lst[k] = cls + FMD_CLS_HEADER_U32 + (2 * np + 2 * (size_t)FMD_FAST_RESERVE) * k;
    cl.lslow = cls + FMD_CLS_HEADER_U32 + (2 * np + 2 * (size_t)FMD_FAST_RESERVE) * FMD_GRP_CLASSES;
    for (int k = 0; k < 2 * FMD_GRP_CLASSES; ++k) cl.fast[k] = cl.lslow + np + 2 * np * k;
    uint32_t *lslow_late = cl.lslow + np + 2 * np * (size_t)(2 * FMD_GRP_CLASSES);   // strands the fast / group kernels hand back (behind the fast lists)
    const size_t fix_off = np;                                                        // the fix-up list (fake forks the group kernels closed), behind that one
    uint32_t *n_slow = cl.cnt + FMD_GRP_CLASSES * FMD_CLS_CNT_STRIDE, *n_late = n_slow + FMD_CLS_LATE_CNT;
    const int use_fast = ovl_use_fast();
    if (!(part == 0 && ovl_tail2_cls(o))) {   // (else: k_ovl_walk<WALK_TAIL2> has zeroed the header and filled the lists)
        FMD_HIP_TRY(hipMemsetAsync(cls, 0, 4 * FMD_CLS_HEADER_U32, st));
        fmd_launch_classify(st, np, rec, listA, o.cap, cl, use_fast, gidx);
    }
    // what classification sets aside (more than 32 candidates, a candidate wider than 63) goes through the lane-per-strand kernel NOW, on a
    // side stream beside the group kernels: it is a handful of long dependent chains (10 ms per 2*10^7 strands of raw reads for 1 % of
    // them), latency from end to end
    bool side = false;
    if (!getenv("FMD_OVLP_SLOW_SERIAL") && ovl_slow_acquire(o.h)) {
        side = hipEventRecord(o.h->slow_ev[0], st) == hipSuccess && hipStreamWaitEvent(o.h->slow_stream, o.h->slow_ev[0], 0) == hipSuccess;
        if (!side) { (void)hipGetLastError(); __atomic_store_n(&o.h->slow_busy, 0, __ATOMIC_RELEASE); }
    }
    {
        hipStream_t ss = side ? o.h->slow_stream : st;
        k_ovl_nei<<<grid, 64, 0, ss>>>(o.ix, np, o.min_match, srev, o.stride_r, o.cap, listA, listB, rec, nei, o.max_nei, seq, o.seq_stride, fmd_next_queue(o.h, ss), cl.lslow, n_slow, gidx);
    }
    // strands whose candidates the walk left in the narrow form: the unforked path (one lane per candidate, no x[0]-side fetch,
    // one shared window per strand and round); whatever turns out not to be that simple moves on to the general list of its class
    if (use_fast)
        for (int k = 0; k < 2 * FMD_GRP_CLASSES; ++k) {
            uint32_t *nk = cl.cnt + (FMD_GRP_CLASSES + 1 + k) * FMD_CLS_CNT_STRIDE;
            const int kg = k % FMD_GRP_CLASSES;
            // one lane per STRAND where its candidates fit a lane's registers (fmd_ovlp_lane.hip), one lane per candidate otherwise (FMD_NEI_LANE=0: always)
            if (fmd_nei_lane_enabled() && fmd_nei_lane_class_ok(kg, k >= FMD_GRP_CLASSES))
                fmd_launch_nei_lane(kg, k >= FMD_GRP_CLASSES, o.h->n_cu, fast_cu, st, o.ix, cl.fast[k], nk, o.cap, listA, listB, rec, nei, o.max_nei, seq,
                                    o.seq_stride, cl.lst[kg], cl.cnt + kg * FMD_CLS_CNT_STRIDE, nk + 8, gidx);
            else
            fmd_launch_nei_fast(kg, k >= FMD_GRP_CLASSES, o.h->n_cu, fast_cu, st, o.ix, cl.fast[k], nk, o.cap, listA, listB, rec, nei, o.max_nei, seq,
                                o.seq_stride, cl.lst[kg], cl.cnt + kg * FMD_CLS_CNT_STRIDE, nk + 8, lslow_late, n_late, gidx);
        }
    // one lane per candidate interval, 64 / G strands per wave
    // (second pass: the strands a group kernel moved to a smaller group when their candidates had died down to single reads -- reads with errors --, largest class first:
    // a strand may move again; their lists live where the fast lists were.  FMD_GRP_DOWN=0: the A/B switch, every strand stays in the group it was admitted to)
    const uint32_t down_cap = ovl_grp_down() && np > 2 * (size_t)FMD_FAST_CHUNK ? (uint32_t)np : 0u;
    for (int k = 0; k < FMD_GRP_CLASSES; ++k)
        fmd_launch_nei_grp(k, o.h->n_cu, per_cu, st, o.ix, cl.lst[k], cl.cnt + k * FMD_CLS_CNT_STRIDE, o.cap, listA, listB, cl, rec, nei, o.max_nei, seq, o.seq_stride, lslow_late, n_late, gidx, fix_off, down_cap, 0);
    if (down_cap)
        for (int k = FMD_GRP_CLASSES - 2; k >= 0; --k)
            fmd_launch_nei_grp(k, o.h->n_cu, per_cu, st, o.ix, cl.fast[k], cl.cnt + (FMD_GRP_CLASSES + 1 + k) * FMD_CLS_CNT_STRIDE + FMD_DOWN_WORD, o.cap, listA, listB, cl, rec, nei, o.max_nei, seq, o.seq_stride,
                               lslow_late, n_late, gidx, fix_off, down_cap, 1);
    // the rest (too many candidates, wide intervals, fake forks, neighbour overflow): lane per strand
    k_ovl_nei<<<grid, 64, 0, st>>>(o.ix, np, o.min_match, srev, o.stride_r, o.cap, listA, listB, rec, nei, o.max_nei, seq, o.seq_stride, q2, lslow_late, n_late, gidx);
    // fake forks among the strands the group kernels finished: the fix-up alone
    k_ovl_fix<<<grid, 64, 0, st>>>(o.ix, lslow_late + fix_off, n_late + FMD_CLS_FIX_CNT, srev, o.stride_r, rec, nei, o.max_nei, seq, o.seq_stride, fmd_next_queue(o.h, st), gidx);
    if (side) {   // the caller's stream owns every row again (and the work area, which the next batch reuses)
        if (hipEventRecord(o.h->slow_ev[1], o.h->slow_stream) != hipSuccess || hipStreamWaitEvent(st, o.h->slow_ev[1], 0) != hipSuccess) { (void)hipGetLastError(); hipStreamSynchronize(o.h->slow_stream); }
        __atomic_store_n(&o.h->slow_busy, 0, __ATOMIC_RELEASE);
    }
    if (getenv("FMD_OVLP_STATS")) { // where the strands of this part went (synchronises: diagnostics only)
        uint32_t hs[FMD_CLS_HEADER_U32];
        hipStreamSynchronize(st);
        hipMemcpy(hs, cls, sizeof(hs), hipMemcpyDeviceToHost);
        uint32_t nf = 0, nb = 0, ng = 0;
        for (int k = 0; k < 2 * FMD_GRP_CLASSES; ++k) { nf += hs[(FMD_GRP_CLASSES + 1 + k) * FMD_CLS_CNT_STRIDE]; nb += hs[(FMD_GRP_CLASSES + 1 + k) * FMD_CLS_CNT_STRIDE + 8]; }
        for (int k = 0; k < FMD_GRP_CLASSES; ++k) ng += hs[k * FMD_CLS_CNT_STRIDE];
        fprintf(stderr, "[M::fmd_ovlp] part of %zu strands: %u to the unforked path (%u of them handed on), %u slots of the general group kernels' lists (holes of the hand-over included), %u through the lane-per-strand kernel at once + %u handed back to it, %u fake forks fixed up\n",
                np, nf, nb, ng, hs[FMD_GRP_CLASSES * FMD_CLS_CNT_STRIDE], hs[FMD_GRP_CLASSES * FMD_CLS_CNT_STRIDE + FMD_CLS_LATE_CNT], hs[FMD_GRP_CLASSES * FMD_CLS_CNT_STRIDE + FMD_CLS_LATE_CNT + FMD_CLS_FIX_CNT]);
    }
#ifdef GRP_STATS
    {
        uint32_t hs[FMD_CLS_HEADER_U32];
        hipStreamSynchronize(st);
        hipMemcpy(hs, cls, sizeof(hs), hipMemcpyDeviceToHost);
        const uint32_t *g = hs + FMD_GRP_CLASSES * FMD_CLS_CNT_STRIDE + FMD_CLS_LATE_CNT + 8;
        fprintf(stderr, "[grp stats] classes %u %u %u %u %u %u slow %u | wave rounds %u, live lanes %u (%.1f %%), lanes of groups holding a strand %u (%.1f %%)\n",
                hs[0], hs[32], hs[64], hs[96], hs[128], hs[160], hs[192], g[0], g[1], 100.0 * g[1] / (64.0 * g[0]), g[2], 100.0 * g[2] / (64.0 * g[0]));
        for (int k = 0; k < 2 * FMD_GRP_CLASSES; ++k) {
            const uint32_t *f = hs + (FMD_GRP_CLASSES + 1 + k) * FMD_CLS_CNT_STRIDE;
            if (f[0]) fprintf(stderr, "[fast stats] G=%d %s: %u strands, %u handed on (%u) | wave rounds %u (%u with a second base), live lanes %.1f %%, lanes of groups holding a strand %.1f %%\n",
                              fmd_grp_size(k % FMD_GRP_CLASSES), k >= FMD_GRP_CLASSES ? "64-bit" : "32-bit", f[0], f[8], f[9], f[10], f[13], 100.0 * f[11] / (64.0 * f[10]), 100.0 * f[12] / (64.0 * f[10]));
        }
    }
#endif
    return FMD_OK;
}

// Pipelined batches (FMD_OVLP_PIPE="parts,walk_per_cu,grp_per_cu,fast_per_cu"; not the default any more).  A batch can be cut
// into parts with phase B of part p on a second stream beside phase A of part p+1, each with a share of the CU's wave slots.  That
// paid while get_nei was the general group kernel alone (92 ms of serial work in 82 on error-free reads).  With the unforked fast
// path phase B is a quarter of a part and bound by instruction issue, the walk wants every wave slot it can get (it runs at the
// memory system's rate of random lines), and the two side by side finish no sooner than one after the other -- 71 against 75 ms on
// error-free reads with "4,8,8,8", 147 against 117 ms on reads with 1 % errors, whose long get_nei phase then starves the walk
// (profiles/r2_ab/ab_fast_pipe.txt).  Default: the serial order.
static void ovl_pipe_config(size_t n, int &parts, int &walk_cu, int &grp_cu, int &fast_cu)
{
    parts = 1; walk_cu = 8; grp_cu = 8; fast_cu = 8;
    const char *e = getenv("FMD_OVLP_PIPE");
    if (e) {
        int a = 0, b = 0, c = 0, d = 0;
        const int k = sscanf(e, "%d,%d,%d,%d", &a, &b, &c, &d);
        if (k >= 1 && a >= 1) parts = a < FMD_OVLP_MAX_PARTS ? a : FMD_OVLP_MAX_PARTS;
        if (k >= 2 && b >= 1) walk_cu = b;
        if (k >= 3 && c >= 1) grp_cu = c;
        if (k >= 4 && d >= 1) fast_cu = d;
    }
    if (getenv("FMD_OVLP_SLOW_ONLY")) parts = 1;
}
static bool ovl_slow_acquire(fmd_dev *h)
{
    int expect = 0;
    if (!__atomic_compare_exchange_n(&h->slow_busy, &expect, 1, false, __ATOMIC_ACQUIRE, __ATOMIC_RELAXED)) return false;
    if (!h->slow_ready) {
        bool ok = hipStreamCreateWithFlags(&h->slow_stream, hipStreamNonBlocking) == hipSuccess;
        int made = 0;
        for (; ok && made < 2; ++made) ok = hipEventCreateWithFlags(&h->slow_ev[made], hipEventDisableTiming) == hipSuccess;
        if (!ok) {
            for (int i = 0; i < made - 1; ++i) hipEventDestroy(h->slow_ev[i]);
            if (h->slow_stream) { hipStreamDestroy(h->slow_stream); h->slow_stream = nullptr; }
            (void)hipGetLastError();
            __atomic_store_n(&h->slow_busy, 0, __ATOMIC_RELEASE);
            return false;
        }
        h->slow_ready = 1;
    }
    return true;
}
static bool ovl_aux_acquire(fmd_dev *h)
{
    int expect = 0;
    if (!__atomic_compare_exchange_n(&h->aux_busy, &expect, 1, false, __ATOMIC_ACQUIRE, __ATOMIC_RELAXED)) return false;
    if (!h->aux_ready) {
        bool ok = hipStreamCreateWithFlags(&h->aux_stream, hipStreamNonBlocking) == hipSuccess; // must not synchronise with the null stream
        int made = 0;
        for (; ok && made <= FMD_OVLP_MAX_PARTS; ++made) ok = hipEventCreateWithFlags(&h->aux_ev[made], hipEventDisableTiming) == hipSuccess;
        if (!ok) {
            for (int i = 0; i < made - 1; ++i) hipEventDestroy(h->aux_ev[i]);
            if (h->aux_stream) { hipStreamDestroy(h->aux_stream); h->aux_stream = nullptr; }   // or every retry would leak a stream
            (void)hipGetLastError();
            __atomic_store_n(&h->aux_busy, 0, __ATOMIC_RELEASE);
            return false;
        }
        h->aux_ready = 1;
    }
    return true;
}

extern "C" int fmd_ovlp_dev(fmd_dev_t *h, void *stream_, size_t n, const uint64_t *d_ids, int min_match, uint32_t max_len,
                            uint32_t max_nei, fmd_ovlp_rec_t *d_rec, fmd_intv_t *d_nei, uint8_t *d_seq, uint32_t seq_stride,
                            void *d_work, size_t work_bytes)
{
    if (!h || (n && (!d_ids || !d_rec || !d_nei || !d_seq || !d_work)) || max_len == 0 || max_nei == 0 || min_match < 0) return FMD_E_ARG;
    if (n == 0) return FMD_OK;
    if (n >= 0xffffff00ull || work_bytes < fmd_ovlp_work_bytes(n, max_len, min_match)) return FMD_E_ARG;
    if (fmd_ovlp_list_cap(max_len, min_match) >= 4096) return FMD_E_ARG; // category index is packed in 12 bits
    FMD_HIP_TRY(hipSetDevice(h->device));
    hipStream_t st = (hipStream_t)stream_;
    OvlBatch o;
    o.h = h; o.ix = fmd_view(h);
    o.ids = d_ids; o.min_match = min_match; o.max_len = max_len; o.max_nei = max_nei; o.seq_stride = seq_stride;
    o.stride_r = (uint32_t)align_up(max_len, 16);
    o.cap = fmd_ovlp_list_cap(max_len, min_match);
    o.srev = (uint8_t *)d_work;
    o.listA = (fmd_intv_t *)((uint8_t *)d_work + align_up(n * (size_t)o.stride_r, 256));
    o.listB = (fmd_intv_t *)((uint8_t *)o.listA + align_up(n * (size_t)o.cap * sizeof(fmd_intv_t), 256));
    o.cls = (uint32_t *)((uint8_t *)o.listB + align_up(n * (size_t)o.cap * sizeof(fmd_intv_t), 256));
    o.rec = d_rec; o.nei = d_nei; o.seq = d_seq; o.gidx = nullptr; o.park = nullptr;

    int parts, walk_cu, grp_cu, fast_cu;
    ovl_pipe_config(n, parts, walk_cu, grp_cu, fast_cu);
    if (parts > 1 && !ovl_aux_acquire(h)) parts = 1;   // the second stream is in use by another call: serial order
    int rc = FMD_OK;
    if (parts == 1) {
        ovl_phase_a(o, st, 0, n, 0);
        rc = ovl_phase_b(o, st, 0, n, 0, 0, 0);
    } else {
        hipStream_t s2 = h->aux_stream;
        // Equal parts: get_nei is the slower phase while the two run side by side, so the last part --
        // whose get_nei has the GPU to itself -- should not be smaller than the others.
        const size_t per = ((n + parts - 1) / parts + 63) & ~(size_t)63;
        int p = 0;
        bool join_ok = true;
        for (size_t b = 0; b < n && rc == FMD_OK; b += per, ++p) {
            const size_t np = n - b < per ? n - b : per;
            const bool last = b + per >= n;
            ovl_phase_a(o, st, b, np, p == 0 ? 0 : walk_cu);         // the first part has the GPU to itself
            if (hipEventRecord(h->aux_ev[p], st) != hipSuccess || hipStreamWaitEvent(s2, h->aux_ev[p], 0) != hipSuccess) { join_ok = false; break; }
            rc = ovl_phase_b(o, s2, b, np, p, last ? 0 : grp_cu, last ? 0 : fast_cu);      // so has the last phase B
        }
        // the caller's stream owns the results again; if the hand-over itself failed, wait on the host
        if (!join_ok || hipEventRecord(h->aux_ev[FMD_OVLP_MAX_PARTS], s2) != hipSuccess ||
            hipStreamWaitEvent(st, h->aux_ev[FMD_OVLP_MAX_PARTS], 0) != hipSuccess) {
            hipStreamSynchronize(s2);
            if (!join_ok) { fmd_set_hip_error(hipGetLastError(), "overlap batch: event hand-over"); rc = FMD_E_HIP; }
        }
        __atomic_store_n(&h->aux_busy, 0, __ATOMIC_RELEASE);
    }
    if (rc != FMD_OK) return rc;
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { fmd_set_hip_error(e, "overlap kernels"); return FMD_E_HIP; }
    return FMD_OK;
}

// What WALK_HEAD needs to take a strand in, resolved for every strand beforehand by a streaming kernel (one thread per strand) instead
// of a chain of four dependent loads in front of a wave's gather: item t of the head's order -> its row, and where its walk stands
// behind the tail table (FmdIndexView::tail + two prefix-table entries; the items come sorted by tail, so neighbouring threads read
// neighbouring entries).  32 bytes per strand:
//   a = { row, k lo, x0 lo, x1 lo }   b = { size lo, k hi | x0 hi << 8 | x1 hi << 16 | size hi << 24, tail as a prefix-table index, flags }
// flags bit 0: no tail-table entry (shorter than ptab_d bases, or a base that is not A/C/G/T among them): k = the sequence id, and
// the walk starts at its sentinel.
__global__ void k_ovl_head_adm(FmdIndexView ix, size_t n, const uint64_t *__restrict__ ids, const uint32_t *__restrict__ order, int use_tail,
                               uint4 *__restrict__ adm)
{
    const size_t step = (size_t)gridDim.x * blockDim.x;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += step) {
        const uint32_t row = order ? order[t] : (uint32_t)t;
        const uint64_t id = ids[row];
        const unsigned long long te = (use_tail && id < ix.n_seq) ? ix.tail[id] : ~0ull;
        uint4 a, b;
        if (te != ~0ull) {
            const uint32_t tfw = (uint32_t)(te >> (64 - 2 * ix.ptab_d));
            const uint64_t k = te & ((1ull << (64 - 2 * ix.ptab_d)) - 1);
            uint32_t r = __brev(~tfw) >> (32 - 2 * ix.ptab_d);   // reverse complement of the ptab index: the 2-bit groups in reverse order, complemented
            const uint32_t trv = ((r >> 1) & 0x55555555u) | ((r & 0x55555555u) << 1);
            const uint4 ef = ix.ptab[tfw], er = ix.ptab[trv];
            fmd_count_lane(ix, 2, 1);
            const uint64_t x0 = (uint64_t)ef.y << 32 | ef.x, sz = ((uint64_t)ef.w << 32 | ef.z) - x0 + 1, x1 = (uint64_t)er.y << 32 | er.x;
            a = make_uint4(row, (uint32_t)k, (uint32_t)x0, (uint32_t)x1);
            b = make_uint4((uint32_t)sz, (uint32_t)(k >> 32) | (uint32_t)(x0 >> 32) << 8 | (uint32_t)(x1 >> 32) << 16 | (uint32_t)(sz >> 32) << 24, tfw, 0u);
        } else {
            a = make_uint4(row, (uint32_t)id, 0u, 0u);
            b = make_uint4(0u, (uint32_t)(id >> 32) & 0xffu, 0u, 1u);
        }
        adm[2 * t] = a; adm[2 * t + 1] = b;
    }
}

// ---- the whole job in an order that keeps neighbours on the genome in flight together ---------------------------------------------
// Work area of fmd_ovlp_sorted_dev: the parked strands (64 bytes each), two (key, row) arrays for the sort, the sort's own
// temporary storage, and the work area of ONE batch of fmd_ovlp_dev.
struct SortedLayout { size_t park, keys_a, keys_b, vals_a, vals_b, tmp, tmp_bytes, batch_area, total; };
static SortedLayout sorted_layout(size_t n, size_t batch, uint32_t max_len, int min_match)
{
    SortedLayout L;
    size_t o = 0;
    L.park = o; o += align_up(n * sizeof(FmdWalkPark), 256);
    L.keys_a = o; o += align_up(n * 4, 256);
    L.keys_b = o; o += align_up(n * 4, 256);
    L.vals_a = o; o += align_up(n * 4, 256);
    L.vals_b = o; o += align_up(n * 4, 256);
    L.tmp_bytes = fmd_park_sort_temp_bytes(n);
    L.tmp = o; o += align_up(L.tmp_bytes, 256);
    L.batch_area = o;   // (also the head's admission records, 32 bytes per strand of the job, while pass 1 runs)
    { const size_t ba = fmd_ovlp_work_bytes(batch, max_len, min_match), ad = align_up(n * 32, 256); o += ba > ad ? ba : ad; }
    L.total = o;
    return L;
}
extern "C" size_t fmd_ovlp_sorted_work_bytes(size_t n, size_t batch, uint32_t max_len, int min_match)
{
    if (batch == 0 || batch > n) batch = n;
    return sorted_layout(n, batch, max_len, min_match).total;
}
// can the two-pass form be used at all?  (nothing may be pushed inside the head; a parked stash is two 16-byte groups of the row)
static bool sorted_eligible(const fmd_dev *h, size_t n, int min_match, uint32_t max_len)
{
    const char *e = getenv("FMD_OVLP_SORT");   // A/B switch: 0 = batches in id order through the one-pass walk
    if (e && atoi(e) == 0) return false;
    return n < 0xffffff00ull && min_match >= (int)FMD_WALK_SPLIT && max_len >= FMD_WALK_SPLIT && h->ptab_d < (int)FMD_WALK_SPLIT;
}

// ---- the two halves of the sorted job as entry points of their own: a caller may do something between them (hand finished rows on
// while the rest is being computed; exchange the parked strands between GPUs by key: fmd_ovlp_dist.hip) --------------------------------
extern "C" int fmd_ovlp_two_pass_ok(const fmd_dev_t *h, size_t n, int min_match, uint32_t max_len)
{
    return h && sorted_eligible(h, n, min_match, max_len) && fmd_ovlp_list_cap(max_len, min_match) < 4096 ? 1 : 0;
}
// work area of the head: the admission records (32 bytes per strand), the unsorted (key, row) arrays, the sort's temporary storage
struct HeadLayout { size_t adm, keys_a, vals_a, tmp, tmp_bytes, total; };
static HeadLayout head_layout(size_t n)
{
    HeadLayout L;
    size_t o = 0;
    L.keys_a = o; o += align_up(n * 4, 256);
    L.vals_a = o; o += align_up(n * 4, 256);
    L.tmp_bytes = fmd_park_sort_temp_bytes(n);
    L.tmp = o; o += align_up(L.tmp_bytes, 256);
    L.adm = o; o += align_up(n * 32, 256);
    L.total = o;
    return L;
}
extern "C" size_t fmd_ovlp_head_work_bytes(size_t n) { return head_layout(n).total; }

// pass 1 with the arrays where the caller wants them: park[n], the sorted keys and the order (row of the t-th strand in key order)
static int ovl_head(fmd_dev *h, hipStream_t st, size_t n, const uint64_t *d_ids, int min_match, uint32_t seq_stride, fmd_ovlp_rec_t *d_rec, FmdWalkPark *park,
                    uint32_t *keys_a, uint32_t *vals_a, uint32_t *keys_sorted, uint32_t *order, void *tmp, size_t tmp_bytes, uint4 *adm)
{
    (void)fmd_pairs_ensure(h, 0);          // the two-base blocks: built here only where FMD_PAIR asks for it (fmd_pair.hip); a caller that keeps the index for many passes calls fmd_dev_build_pairs
    const FmdIndexView ix = fmd_view(h);
    // pass 1: every strand FMD_WALK_SPLIT bases in, in the caller's order.  (Taking the strands in the order of their last ptab_d bases --
    // the tail table has them, one more radix sort -- makes this pass 7 % faster and costs what it saves: profiles/r3_locality.)
    {
        const uint32_t *order1 = nullptr;
        const int use_tail = ix.tail != nullptr && ix.ptab != nullptr && min_match >= ix.ptab_d && ix.ptab_d >= 2;
        size_t blocks = (n + 255) / 256;
        if (blocks > (1u << 20)) blocks = 1u << 20;
        k_ovl_head_adm<<<(unsigned)blocks, 256, 0, st>>>(ix, n, d_ids, order1, use_tail, adm);
        uint32_t *q = fmd_next_queue(h, st);
        int grid = fmd_grid_for_lds(h, n, FMD_COMPACT_LDS_U4 * 16);
        { const char *e = getenv("FMD_HEAD_WAVES"); if (e && atoi(e) > 0 && grid > h->n_cu * atoi(e)) grid = h->n_cu * atoi(e); }
        bool pairs = ix.pair != nullptr && ix.pair_tab != nullptr && n >= 4096 && n < 0xffffff00ull;
        { const char *e = getenv("FMD_PAIR_USE"); if (e && atoi(e) == 0) pairs = false; }   // A/B switch on a handle that has the blocks
        uint32_t *strag = pairs ? (uint32_t *)fmd_scratch_acquire(h, (n + 1) * 4) : nullptr;
        if (pairs && strag) {
            // single steps up to `from` (by then a strand's interval is narrow), two bases per request from there to FMD_WALK_SPLIT (k_ovl_pair), and
            // the strands that kernel could not take all the way once more from their admission records, single steps all the way
            int from = 16;
            { const char *e = getenv("FMD_PAIR_FROM"); if (e && atoi(e) > ix.ptab_d && atoi(e) < (int)FMD_WALK_SPLIT && !(atoi(e) & 1)) from = atoi(e); }
            if (from <= ix.ptab_d) from = (ix.ptab_d + 2) & ~1;
            k_ovl_walk<WALK_HEAD><<<grid, 64, 0, st>>>(ix, n, d_ids, min_match, nullptr, (uint32_t)sizeof(FmdWalkPark), 0, nullptr, d_rec,
                                                    nullptr, seq_stride, q, 0, park, nullptr, adm, walk_ticket_chunk("FMD_HEAD_TICKETS", 256, n, grid), nullptr, nullptr, from | walk_gate("FMD_HEAD_GATE", 0) << 16);
            (void)hipMemsetAsync(strag, 0, 4, st);
            uint32_t *q2 = fmd_next_queue(h, st);
            int grid2 = h->n_cu * (FMD_PAIR_LB * 4 < 20 ? FMD_PAIR_LB * 4 : 20);      // (8 KiB of LDS per wave: twenty fit a CU)
            if ((size_t)grid2 > (n + 63) / 64) grid2 = (int)((n + 63) / 64);
            { const char *e = getenv("FMD_PAIR_WAVES"); if (e && atoi(e) > 0 && grid2 > h->n_cu * atoi(e)) grid2 = h->n_cu * atoi(e); }
            k_ovl_pair<<<grid2, 64, 0, st>>>(ix, n, park, q2, walk_ticket_chunk("FMD_PAIR_TICKETS", 256, n, grid2), strag);
            uint32_t n_strag = 0;
            hipError_t e1 = hipMemcpyAsync(&n_strag, strag, 4, hipMemcpyDeviceToHost, st);
            if (e1 == hipSuccess) e1 = hipStreamSynchronize(st);
            if (e1 != hipSuccess) { fmd_set_hip_error(e1, "two-base pass"); fmd_scratch_release(h, strag); return FMD_E_HIP; }
            if (n_strag) {
                uint4 *adm2 = (uint4 *)fmd_scratch_acquire(h, (size_t)n_strag * 32);
                if (!adm2) { fmd_scratch_release(h, strag); return FMD_E_NOMEM; }
                k_ovl_strag_adm<<<(n_strag + 255) / 256 < 65536 ? (n_strag + 255) / 256 : 65536, 256, 0, st>>>(strag, adm, adm2);
                uint32_t *q3 = fmd_next_queue(h, st);
                int grid3 = fmd_grid_for_lds(h, n_strag, FMD_COMPACT_LDS_U4 * 16);
                k_ovl_walk<WALK_HEAD><<<grid3, 64, 0, st>>>(ix, n_strag, d_ids, min_match, nullptr, (uint32_t)sizeof(FmdWalkPark), 0, nullptr, d_rec,
                                                         nullptr, seq_stride, q3, 0, park, nullptr, adm2, walk_ticket_chunk("FMD_HEAD_TICKETS", 256, n_strag, grid3), nullptr, nullptr, 0);
                e1 = hipStreamSynchronize(st);      // (adm2 goes back to the handle's cache)
                fmd_scratch_release(h, adm2);
                if (e1 != hipSuccess) { fmd_set_hip_error(e1, "two-base pass"); fmd_scratch_release(h, strag); return FMD_E_HIP; }
            }
            fmd_scratch_release(h, strag);
            if (getenv("FMD_DEBUG_PAIR")) fprintf(stderr, "[M::ovl_head] two-base pass from depth %d: %u of %zu strands walked again one base at a time\n", from, n_strag, n);
        } else {
        if (strag) fmd_scratch_release(h, strag);
        k_ovl_walk<WALK_HEAD><<<grid, 64, 0, st>>>(ix, n, d_ids, min_match, nullptr, (uint32_t)sizeof(FmdWalkPark), 0, nullptr, d_rec,
                                                nullptr, seq_stride, q, 0, park, nullptr, adm, walk_ticket_chunk("FMD_HEAD_TICKETS", 256, n, grid), nullptr, nullptr, walk_gate("FMD_HEAD_GATE", 0) << 16);
        }
    }
    // the order of pass 2: rows sorted by the minimizer of the bases each strand has shown so far
    return fmd_park_sort(st, n, park, keys_a, keys_sorted, vals_a, order, tmp, tmp_bytes);
}

extern "C" int fmd_ovlp_head_dev(fmd_dev_t *h, void *stream_, size_t n, const uint64_t *d_ids, int min_match, uint32_t max_len, fmd_ovlp_rec_t *d_rec,
                                 void *d_park, uint32_t *d_keys, uint32_t *d_order, void *d_work, size_t work_bytes)
{
    if (!h || (n && (!d_ids || !d_rec || !d_park || !d_keys || !d_order || !d_work)) || max_len == 0 || min_match < 0) return FMD_E_ARG;
    if (n == 0) return FMD_OK;
    if (!fmd_ovlp_two_pass_ok(h, n, min_match, max_len)) return FMD_E_ARG;
    const HeadLayout L = head_layout(n);
    if (work_bytes < L.total) return FMD_E_ARG;
    FMD_HIP_TRY(hipSetDevice(h->device));
    uint8_t *w = (uint8_t *)d_work;
    const int rc = ovl_head(h, (hipStream_t)stream_, n, d_ids, min_match, 2 * max_len, d_rec, (FmdWalkPark *)d_park, (uint32_t *)(w + L.keys_a), (uint32_t *)(w + L.vals_a),
                            d_keys, d_order, w + L.tmp, L.tmp_bytes, (uint4 *)(w + L.adm));
    if (rc != FMD_OK) return rc;
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { fmd_set_hip_error(e, "sorted overlap job: head"); return FMD_E_HIP; }
    return FMD_OK;
}

// pass 2 + fm6_get_nei for np parked strands: slot t of the call = row d_rows[t] of d_park, d_rec, d_nei and d_seq
static int ovl_tail(fmd_dev *h, hipStream_t st, size_t np, const uint32_t *rows, FmdWalkPark *park, const uint64_t *d_ids, int min_match, uint32_t max_len, uint32_t max_nei,
                    fmd_ovlp_rec_t *d_rec, fmd_intv_t *d_nei, uint8_t *d_seq, uint32_t seq_stride, uint8_t *area, size_t area_strands)
{
    OvlBatch o;
    o.h = h; o.ix = fmd_view(h); o.ids = d_ids; o.min_match = min_match; o.max_len = max_len; o.max_nei = max_nei; o.seq_stride = seq_stride;
    o.stride_r = (uint32_t)align_up(max_len, 16);
    o.cap = fmd_ovlp_list_cap(max_len, min_match);
    o.srev = area;
    o.listA = (fmd_intv_t *)(o.srev + align_up(area_strands * (size_t)o.stride_r, 256));
    o.listB = (fmd_intv_t *)((uint8_t *)o.listA + align_up(area_strands * (size_t)o.cap * sizeof(fmd_intv_t), 256));
    o.cls = (uint32_t *)((uint8_t *)o.listB + align_up(area_strands * (size_t)o.cap * sizeof(fmd_intv_t), 256));
    o.rec = d_rec; o.nei = d_nei; o.seq = d_seq; o.park = park;
    o.gidx = rows;
    ovl_phase_a(o, st, 0, np, 0);
    return ovl_phase_b(o, st, 0, np, 0, 0, 0);
}

extern "C" int fmd_ovlp_tail_dev(fmd_dev_t *h, void *stream_, size_t np, const uint32_t *d_rows, void *d_park, int min_match, uint32_t max_len, uint32_t max_nei,
                                 fmd_ovlp_rec_t *d_rec, fmd_intv_t *d_nei, uint8_t *d_seq, uint32_t seq_stride, void *d_work, size_t work_bytes)
{
    if (!h || (np && (!d_rows || !d_park || !d_rec || !d_nei || !d_seq || !d_work)) || max_len == 0 || max_nei == 0 || min_match < 0) return FMD_E_ARG;
    if (np == 0) return FMD_OK;
    if (np >= 0xffffff00ull || !fmd_ovlp_two_pass_ok(h, np, min_match, max_len) || work_bytes < fmd_ovlp_work_bytes(np, max_len, min_match)) return FMD_E_ARG;
    FMD_HIP_TRY(hipSetDevice(h->device));
    const int rc = ovl_tail(h, (hipStream_t)stream_, np, d_rows, (FmdWalkPark *)d_park, nullptr, min_match, max_len, max_nei, d_rec, d_nei, d_seq, seq_stride, (uint8_t *)d_work, np);
    if (rc != FMD_OK) return rc;
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { fmd_set_hip_error(e, "sorted overlap job: tail"); return FMD_E_HIP; }
    return FMD_OK;
}

extern "C" int fmd_ovlp_sorted_dev(fmd_dev_t *h, void *stream_, size_t n, const uint64_t *d_ids, int min_match, uint32_t max_len,
                                   uint32_t max_nei, fmd_ovlp_rec_t *d_rec, fmd_intv_t *d_nei, uint8_t *d_seq, uint32_t seq_stride,
                                   void *d_work, size_t work_bytes, size_t batch)
{
    if (!h || (n && (!d_ids || !d_rec || !d_nei || !d_seq || !d_work)) || max_len == 0 || max_nei == 0 || min_match < 0) return FMD_E_ARG;
    if (n == 0) return FMD_OK;
    if (n >= 0xffffff00ull || fmd_ovlp_list_cap(max_len, min_match) >= 4096) return FMD_E_ARG;
    FMD_HIP_TRY(hipSetDevice(h->device));
    hipStream_t st = (hipStream_t)stream_;
    if (batch == 0 || batch > n) batch = n;
    if (!sorted_eligible(h, n, min_match, max_len)) {   // the same strands in id order, batch by batch
        if (work_bytes < fmd_ovlp_work_bytes(batch, max_len, min_match)) return FMD_E_ARG;
        for (size_t b = 0; b < n; b += batch) {
            const size_t np = n - b < batch ? n - b : batch;
            const int rc = fmd_ovlp_dev(h, stream_, np, d_ids + b, min_match, max_len, max_nei, d_rec + b, d_nei + b * (size_t)max_nei,
                                        d_seq + b * (size_t)seq_stride, seq_stride, d_work, work_bytes);
            if (rc != FMD_OK) return rc;
        }
        return FMD_OK;
    }
    const SortedLayout L = sorted_layout(n, batch, max_len, min_match);
    if (work_bytes < L.total) return FMD_E_ARG;
    uint8_t *w = (uint8_t *)d_work;
    FmdWalkPark *park = (FmdWalkPark *)(w + L.park);
    uint32_t *sorted = (uint32_t *)(w + L.vals_b);
    // (the admission records live in the batch area, which is idle until pass 2; 32 bytes per strand of the job)
    {
        const int rc = ovl_head(h, st, n, d_ids, min_match, seq_stride, d_rec, park, (uint32_t *)(w + L.keys_a), (uint32_t *)(w + L.vals_a), (uint32_t *)(w + L.keys_b), sorted,
                                w + L.tmp, L.tmp_bytes, (uint4 *)(w + L.batch_area));
        if (rc != FMD_OK) return rc;
    }
    // pass 2 + fm6_get_nei, batch by batch in that order
    for (size_t b = 0; b < n; b += batch) {
        const size_t np = n - b < batch ? n - b : batch;
        const int rc = ovl_tail(h, st, np, sorted + b, park, d_ids, min_match, max_len, max_nei, d_rec, d_nei, d_seq, seq_stride, w + L.batch_area, batch);
        if (rc != FMD_OK) return rc;
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { fmd_set_hip_error(e, "sorted overlap job"); return FMD_E_HIP; }
    return FMD_OK;
}

// ---- rows that exceeded a capacity: again, alone, larger -------------------------------------------------------------------------
// fm6_get_nei has no capacities (kvec grows, unitig.c:93-179); here a row that needs more neighbours than max_nei, a longer candidate
// list or more bases than max_len is flagged (FMD_OVLP_F_OVERFLOW) instead of answered.  This collects the ids of the flagged rows of
// a finished job on the device and runs them through fmd_ovlp_dev with the capacities the caller names, into a side table; rows of
// the side table that are still flagged are counted so that the caller can go one size up.
__global__ void k_ovl_collect_overflow(size_t n, const fmd_ovlp_rec_t *__restrict__ rec, const uint64_t *__restrict__ ids, uint64_t cap,
                                       uint64_t *__restrict__ out_ids, uint32_t *__restrict__ out_rows, unsigned long long *__restrict__ counter)
{
    const size_t step = (size_t)gridDim.x * blockDim.x;
    for (size_t i0 = (size_t)blockIdx.x * blockDim.x; i0 < n; i0 += step) {
        const size_t i = i0 + threadIdx.x;
        const bool hit = i < n && (rec[i].flags & FMD_OVLP_F_OVERFLOW) != 0;
        const unsigned long long m = __ballot(hit);
        if (m == 0) continue;
        const int lane = threadIdx.x & 63;
        unsigned long long base = 0;
        if (lane == __ffsll((long long)m) - 1) base = atomicAdd(counter, (unsigned long long)__popcll(m));   // one atomic per wave that has any
        base = __shfl(base, __ffsll((long long)m) - 1);
        if (hit) {
            const unsigned long long slot = base + (unsigned long long)__popcll(m & ((1ull << lane) - 1ull));
            if (slot < cap) { out_ids[slot] = ids ? ids[i] : (uint64_t)i; if (out_rows) out_rows[slot] = (uint32_t)i; }
        }
    }
}
__global__ void k_ovl_count_overflow(size_t n, const fmd_ovlp_rec_t *__restrict__ rec, unsigned long long *__restrict__ counter)
{
    const size_t step = (size_t)gridDim.x * blockDim.x;
    unsigned long long c = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += step) c += (rec[i].flags & FMD_OVLP_F_OVERFLOW) != 0;
    if (c) atomicAdd(counter, c);
}
extern "C" size_t fmd_ovlp_side_work_bytes(size_t side_cap, uint32_t max_len, int min_match)
{
    return 256 + fmd_ovlp_work_bytes(side_cap, max_len, min_match);
}
extern "C" int fmd_ovlp_rerun_overflow_dev(fmd_dev_t *h, void *stream_, size_t n, const uint64_t *d_ids, const fmd_ovlp_rec_t *d_rec, int min_match, uint32_t max_len,
                                           uint32_t max_nei, uint64_t side_cap, uint64_t *d_side_ids, uint32_t *d_side_rows, fmd_ovlp_rec_t *d_side_rec, fmd_intv_t *d_side_nei,
                                           uint8_t *d_side_seq, uint32_t side_stride, void *d_work, size_t work_bytes, uint64_t *n_side, uint64_t *n_still)
{
    if (!h || !n_side || !n_still || (n && !d_rec) || (side_cap && (!d_side_ids || !d_side_rec || !d_side_nei || !d_side_seq || !d_work)) || max_len == 0 || max_nei == 0) return FMD_E_ARG;
    *n_side = 0; *n_still = 0;
    if (n == 0) return FMD_OK;
    if (work_bytes < fmd_ovlp_side_work_bytes(side_cap, max_len, min_match)) return FMD_E_ARG;
    FMD_HIP_TRY(hipSetDevice(h->device));
    hipStream_t st = (hipStream_t)stream_;
    unsigned long long *ctr = (unsigned long long *)d_work;   // [0] flagged rows of the job, [1] rows of the side table still flagged
    FMD_HIP_TRY(hipMemsetAsync(ctr, 0, 16, st));
    size_t blocks = (n + 255) / 256;
    if (blocks > (1u << 16)) blocks = 1u << 16;
    k_ovl_collect_overflow<<<(unsigned)blocks, 256, 0, st>>>(n, d_rec, d_ids, side_cap, d_side_ids, d_side_rows, ctr);
    unsigned long long hc[2] = {0, 0};
    FMD_HIP_TRY(hipMemcpyAsync(hc, ctr, 8, hipMemcpyDeviceToHost, st));
    FMD_HIP_TRY(hipStreamSynchronize(st));
    *n_side = hc[0];
    if (hc[0] == 0) return FMD_OK;
    if (hc[0] > side_cap) return FMD_E_OVERFLOW;    // (the caller's side table is too small: *n_side says how many rows there are)
    const size_t k = (size_t)hc[0];
    FMD_HIP_TRY(hipMemsetAsync(d_side_seq, 0, k * (size_t)side_stride, st));
    const int rc = fmd_ovlp_dev(h, stream_, k, d_side_ids, min_match, max_len, max_nei, d_side_rec, d_side_nei, d_side_seq, side_stride, (uint8_t *)d_work + 256, work_bytes - 256);
    if (rc != FMD_OK) return rc;
    k_ovl_count_overflow<<<(unsigned)((k + 255) / 256 > 4096 ? 4096 : (k + 255) / 256), 256, 0, st>>>(k, d_side_rec, ctr + 1);
    FMD_HIP_TRY(hipMemcpyAsync(hc + 1, ctr + 1, 8, hipMemcpyDeviceToHost, st));
    FMD_HIP_TRY(hipStreamSynchronize(st));
    *n_still = hc[1];
    return FMD_OK;
}

// check_left_simple (unitig.c:186-204) for every strand of a finished fmd_ovlp_dev batch that has a
// unique neighbour; writes rec.reserved.  Same buffers and work area as the fmd_ovlp_dev call.
extern "C" int fmd_ovlp_check_left_dev(fmd_dev_t *h, void *stream_, size_t n, int min_match, uint32_t max_len, fmd_ovlp_rec_t *d_rec,
                                       const uint8_t *d_seq, uint32_t seq_stride, void *d_work, size_t work_bytes)
{
    if (!h || (n && (!d_rec || !d_seq || !d_work)) || max_len == 0) return FMD_E_ARG;
    if (n == 0) return FMD_OK;
    if (work_bytes < fmd_ovlp_work_bytes(n, max_len, min_match)) return FMD_E_ARG;
    FMD_HIP_TRY(hipSetDevice(h->device));
    hipStream_t st = (hipStream_t)stream_;
    const uint32_t stride_r = (uint32_t)align_up(max_len, 16);
    const uint32_t cap = fmd_ovlp_list_cap(max_len, min_match);
    fmd_intv_t *listA = (fmd_intv_t *)((uint8_t *)d_work + align_up(n * (size_t)stride_r, 256));
    fmd_intv_t *listB = (fmd_intv_t *)((uint8_t *)listA + align_up(n * (size_t)cap * sizeof(fmd_intv_t), 256));
    uint32_t *q = fmd_next_queue(h, st);
    k_ovl_cls<<<fmd_grid_for(h, n), 64, 0, st>>>(fmd_view(h), n, min_match, cap, listA, listB, d_rec, d_seq, seq_stride, q);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { fmd_set_hip_error(e, "k_ovl_cls"); return FMD_E_HIP; }
    return FMD_OK;
}

// fm6_retrieve (exact.c:100-127) for a batch of sequence ids: rank, `$read$` bi-interval and
// containment of each sequence, i.e. the walk phase alone with no length threshold.  What
// fm6_seqsort (seqsort.c:12-35) needs.  rec.status = -3 when contained on either side.
extern "C" int fmd_seqinfo_dev(fmd_dev_t *h, void *stream_, size_t n, const uint64_t *d_ids, uint32_t max_len, fmd_ovlp_rec_t *d_rec,
                               uint8_t *d_seq, uint32_t seq_stride, void *d_work, size_t work_bytes)
{
    if (!h || (n && (!d_ids || !d_rec || !d_seq || !d_work)) || max_len == 0) return FMD_E_ARG;
    if (n == 0) return FMD_OK;
    if (n >= 0xffffff00ull || work_bytes < fmd_ovlp_work_bytes(n, max_len, (int)max_len - 1)) return FMD_E_ARG;
    FMD_HIP_TRY(hipSetDevice(h->device));
    hipStream_t st = (hipStream_t)stream_;
    const uint32_t stride_r = (uint32_t)align_up(max_len, 16);
    const uint32_t cap = fmd_ovlp_list_cap(max_len, (int)max_len - 1);
    uint8_t *srev = (uint8_t *)d_work;
    fmd_intv_t *listA = (fmd_intv_t *)((uint8_t *)d_work + align_up(n * (size_t)stride_r, 256));
    uint32_t *q0 = fmd_next_queue(h, st);
    k_ovl_walk<WALK_WHOLE><<<fmd_grid_for_lds(h, n, FMD_COMPACT_LDS_U4 * 16), 64, 0, st>>>(fmd_view(h), n, d_ids, 0, srev, stride_r, cap, listA, d_rec,
                                                                                         d_seq, seq_stride, q0, 1, nullptr, nullptr, nullptr, FMD_TICKET_CHUNK, nullptr, nullptr, 0);
    launch_seq_out(st, n, max_len, srev, stride_r, d_rec, 0, 1, d_seq, seq_stride);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { fmd_set_hip_error(e, "k_ovl_walk"); return FMD_E_HIP; }
    return FMD_OK;
}

struct DevBuf2 {
    void *p = nullptr;
    int alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 16) == hipSuccess ? FMD_OK : FMD_E_NOMEM; }
    ~DevBuf2() { if (p) hipFree(p); }
};

// hipHostRegister for the lifetime of a scope; only for large arrays, silently skipped when it fails
struct HostPin {
    void *p;
    HostPin(void *ptr, size_t bytes) : p(nullptr)
    {
        if (bytes >= ((size_t)64 << 20) && !getenv("FMD_NO_PIN") && hipHostRegister(ptr, bytes, hipHostRegisterDefault) == hipSuccess) p = ptr;
        else (void)hipGetLastError();
    }
    ~HostPin() { if (p) hipHostUnregister(p); }
    HostPin(const HostPin &) = delete;
    HostPin &operator=(const HostPin &) = delete;
};

extern "C" int fmd_ovlp_batch(fmd_dev_t *h, size_t n, const uint64_t *ids, int min_match, uint32_t max_len, uint32_t max_nei,
                              fmd_ovlp_rec_t *rec, fmd_intv_t *nei, uint8_t *seq, uint32_t seq_stride, int with_check_left)
{
    if (!h || (n && (!ids || !rec || !nei || !seq))) return FMD_E_ARG;
    if (n == 0) return FMD_OK;
    FMD_HIP_TRY(hipSetDevice(h->device));
    // strands per pass: as many as keep the HBM work area (two candidate lists per strand: 6.4 kB at 100 bp,
    // -l50) around 32 GB, between 2^16 and 2^22
    const size_t per_strand = fmd_ovlp_work_bytes(1u << 16, max_len, min_match) >> 16;
    size_t chunk = ((size_t)32 << 30) / (per_strand ? per_strand : 1);
    if (chunk > (4u << 20)) chunk = 4u << 20;
    if (chunk < (1u << 16)) chunk = 1u << 16;
    const size_t m = n < chunk ? n : chunk;
    const size_t wb = fmd_ovlp_work_bytes(m, max_len, min_match);
    DevBuf2 di, dr, dn, ds, dw;
    if (di.alloc(m * 8) || dr.alloc(m * sizeof(fmd_ovlp_rec_t)) || dn.alloc(m * max_nei * sizeof(fmd_intv_t)) ||
        ds.alloc(m * (size_t)seq_stride) || dw.alloc(wb)) return FMD_E_NOMEM;
    // Gigabytes come back per call: pin the caller's arrays for its duration so the copies run at link speed
    // instead of through the runtime's staging buffers (best effort; pageable copies otherwise).
    HostPin pin_rec(rec, n * sizeof(fmd_ovlp_rec_t)), pin_nei(nei, n * max_nei * sizeof(fmd_intv_t)), pin_seq(seq, n * (size_t)seq_stride);
    for (size_t o = 0; o < n; o += m) {
        const size_t c = n - o < m ? n - o : m;
        FMD_HIP_TRY(hipMemcpy(di.p, ids + o, c * 8, hipMemcpyHostToDevice));
        FMD_HIP_TRY(hipMemset(ds.p, 0, c * (size_t)seq_stride));
        FMD_HIP_TRY(hipMemset(dn.p, 0, c * max_nei * sizeof(fmd_intv_t)));
        int rc = fmd_ovlp_dev(h, nullptr, c, (uint64_t *)di.p, min_match, max_len, max_nei, (fmd_ovlp_rec_t *)dr.p,
                              (fmd_intv_t *)dn.p, (uint8_t *)ds.p, seq_stride, dw.p, wb);
        if (rc) return rc;
        if (with_check_left) {
            rc = fmd_ovlp_check_left_dev(h, nullptr, c, min_match, max_len, (fmd_ovlp_rec_t *)dr.p, (uint8_t *)ds.p, seq_stride, dw.p, wb);
            if (rc) return rc;
        }
        FMD_HIP_TRY(hipMemcpy(rec + o, dr.p, c * sizeof(fmd_ovlp_rec_t), hipMemcpyDeviceToHost));
        FMD_HIP_TRY(hipMemcpy(nei + o * max_nei, dn.p, c * max_nei * sizeof(fmd_intv_t), hipMemcpyDeviceToHost));
        FMD_HIP_TRY(hipMemcpy(seq + o * (size_t)seq_stride, ds.p, c * (size_t)seq_stride, hipMemcpyDeviceToHost));
    }
    return FMD_OK;
}

extern "C" int fmd_seqinfo_batch(fmd_dev_t *h, size_t n, const uint64_t *ids, uint32_t max_len, fmd_ovlp_rec_t *rec)
{
    if (!h || (n && (!ids || !rec))) return FMD_E_ARG;
    if (n == 0) return FMD_OK;
    FMD_HIP_TRY(hipSetDevice(h->device));
    const size_t chunk = 1u << 21;
    const size_t m = n < chunk ? n : chunk;
    const uint32_t stride = (uint32_t)align_up(max_len, 4);
    const size_t wb = fmd_ovlp_work_bytes(m, max_len, (int)max_len - 1);
    DevBuf2 di, dr, ds, dw;
    if (di.alloc(m * 8) || dr.alloc(m * sizeof(fmd_ovlp_rec_t)) || ds.alloc(m * (size_t)stride) || dw.alloc(wb)) return FMD_E_NOMEM;
    for (size_t o = 0; o < n; o += m) {
        const size_t c = n - o < m ? n - o : m;
        FMD_HIP_TRY(hipMemcpy(di.p, ids + o, c * 8, hipMemcpyHostToDevice));
        int rc = fmd_seqinfo_dev(h, nullptr, c, (uint64_t *)di.p, max_len, (fmd_ovlp_rec_t *)dr.p, (uint8_t *)ds.p, stride, dw.p, wb);
        if (rc) return rc;
        FMD_HIP_TRY(hipMemcpy(rec + o, dr.p, c * sizeof(fmd_ovlp_rec_t), hipMemcpyDeviceToHost));
    }
    return FMD_OK;
}
