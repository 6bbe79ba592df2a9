// fmd_kernel_common.h -- small device helpers shared by the search kernels (overlap, SMEM, k-mer harvest)
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

// Candidate intervals of one strand (the list overlap_intv builds, unitig.c:38-64), as the walk leaves them for the get_nei kernels.
// Wide form = store_entry(x0, x1, size, depth).  Narrow form (size <= 63, depth < 65536: every candidate the walk pushes from its
// 64-position window path) carries two more facts about BWT[x0, x0 + size) that the walk has at hand when it pushes and that
// fm6_get_nei would otherwise fetch the block of x0 for again: D = the positions of '$' in that range (bit i = BWT[x0 + i] is '$':
// the reads that START with the candidate string, the sentinel tests of unitig.c:112/:129) and r0 = the number of '$' before x0
// (x[0] of the neighbour interval, unitig.c:113).  Both follow a forward extension without touching memory (the child ranges are
// sub-ranges of [x0, x0 + size)), so the unforked fast path of get_nei (k_ovl_nei_fast) never reads the x[0] side at all.
//   q[0] = { x0 lo, x0 bits 32..39 | r0 bits 32..39 << 8 | depth << 16, x1 lo, x1 hi }     q[1] = { D lo, D hi, r0 lo, NARROW | size }
#define FMD_CAND_NARROW 0x80000000u
struct FmdCand { uint64_t x0, x1, sz, D, r0; uint32_t depth; bool narrow; };
__device__ __forceinline__ FmdCand cand_decode(const uint4 a, const uint4 b)
{
    FmdCand c;
    c.narrow = (b.w & FMD_CAND_NARROW) != 0;
    c.x1 = (uint64_t)a.w << 32 | a.z;
    if (c.narrow) {
        c.x0 = (uint64_t)(a.y & 0xffu) << 32 | a.x; c.r0 = (uint64_t)((a.y >> 8) & 0xffu) << 32 | b.z; c.depth = a.y >> 16;
        c.sz = b.w & 63u; c.D = (uint64_t)b.y << 32 | b.x;
    } else {
        c.x0 = (uint64_t)a.y << 32 | a.x; c.sz = ((uint64_t)b.y << 32 | b.x) & 0xffffffffffffull; c.depth = b.z; c.D = 0; c.r0 = 0;
    }
    return c;
}
__device__ __forceinline__ void cand_store_narrow(fmd_intv_t *e, uint64_t x0, uint64_t x1, uint32_t sz, uint32_t depth, uint64_t D, uint64_t r0)
{
    uint4 *q = (uint4 *)e;
    q[0] = make_uint4((uint32_t)x0, ((uint32_t)(x0 >> 32) & 0xffu) | ((uint32_t)(r0 >> 32) & 0xffu) << 8 | depth << 16, (uint32_t)x1, (uint32_t)(x1 >> 32));
    q[1] = make_uint4((uint32_t)D, (uint32_t)(D >> 32), (uint32_t)r0, FMD_CAND_NARROW | sz);
}

// ---- 64-position window over a lane's block images (used when an SA interval is narrower than 64)
__device__ __forceinline__ uint64_t bits_below(int j) { return j >= 64 ? ~0ull : ((1ull << j) - 1); }

// The three 32-position chunks (bit-plane words) that cover BWT[pos, pos + 64), from the lane's block
// images: img_k holds block blk_k (when has_k), img_l holds blk_l (when has_l).  Words of other blocks
// read as zero (they are masked out by the callers' range masks).
__device__ __forceinline__ uint4 grp_pick(const uint4 *img_k, int t_k, const uint4 *img_l, int t_l, uint32_t blk_k, uint32_t blk_l,
                                          bool has_k, bool has_l, uint32_t blk, uint32_t ch)
{
    const bool in_k = has_k && blk == blk_k, in_l = has_l && blk == blk_l;
    uint4 v = in_k ? img_k[(int)ch ^ t_k] : img_l[(int)ch ^ t_l];
    if (!in_k && !in_l) v = make_uint4(0, 0, 0, 0);
    return v;
}
// The window starts one position after `p`, whose block and offset (blk_p, off_p) the caller already has
// (p = the k side of the rank pair that brought the images in): no second division.
__device__ __forceinline__ void grp_window(const uint4 *img_k, int t_k, const uint4 *img_l, int t_l, uint32_t blk_k, uint32_t blk_l,
                                           bool has_k, bool has_l, uint32_t blk_p, uint32_t off_p, uint4 &a, uint4 &b, uint4 &c)
{
    uint32_t blk = blk_p, ch = (off_p + 1) >> 5;
    // 32-position words counted from the start of block blk_p: words 0..2 are its chunks, word q >= 3 is chunk q - 2 of the next block
    if (ch >= FMD_BLK_CHUNKS) { ch -= FMD_BLK_OWN_CHUNKS; ++blk; }
    a = grp_pick(img_k, t_k, img_l, t_l, blk_k, blk_l, has_k, has_l, blk, ch);
    if (++ch == FMD_BLK_CHUNKS) { ch = FMD_BLK_CHUNKS - FMD_BLK_OWN_CHUNKS; ++blk; }
    b = grp_pick(img_k, t_k, img_l, t_l, blk_k, blk_l, has_k, has_l, blk, ch);
    if (++ch == FMD_BLK_CHUNKS) { ch = FMD_BLK_CHUNKS - FMD_BLK_OWN_CHUNKS; ++blk; }
    c = grp_pick(img_k, t_k, img_l, t_l, blk_k, blk_l, has_k, has_l, blk, ch);
}
__device__ __forceinline__ uint64_t win64(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t sh)
{
    // two funnel shifts (v_alignbit_b32: low word of {hi, lo} >> sh, sh in 0..31)
    const uint32_t lo = __builtin_amdgcn_alignbit(w1, w0, sh), hi = __builtin_amdgcn_alignbit(w2, w1, sh);
    return (uint64_t)hi << 32 | lo;
}
__device__ __forceinline__ uint64_t range64(uint32_t a, uint32_t b) // bits [a, b), b <= 64
{
    return bits_below((int)b) & ~bits_below((int)a);
}


// Work lists of the get_nei kernels.  A strand with m candidate intervals goes to the group kernel with
// the smallest group size G >= m (one lane per candidate, 64 / G strands per wave); the sizes are chosen
// so that 64 / G groups leave at most 4 lanes unused.  G = 4 is there for reads with errors: an error cuts the overlaps a
// strand has short, and 36 % of the strands of 30-fold reads with 1 % substitutions that have any candidate have at most four
// (error-free: 0.1 %) -- in groups of 8 half of their lanes never hold anything.
#define FMD_GRP_CLASSES 6
__device__ __host__ __forceinline__ constexpr int fmd_grp_size(int k) { return k == 0 ? 4 : k == 1 ? 8 : k == 2 ? 12 : k == 3 ? 16 : k == 4 ? 21 : 32; }
#define FMD_CLS_CNT_STRIDE 32              // counters sit on separate 128-byte lines
#define FMD_CLS_HEADER_U32 640             // the counter area in front of the lists (3 * FMD_GRP_CLASSES + 1 lines)
// hand-over from k_ovl_nei_fast to k_ovl_nei_grp: list slots reserved FMD_FAST_CHUNK at a time, unused ones stay holes
#define FMD_FAST_CHUNK 16
#define FMD_FAST_MAX_WAVES 8192
#define FMD_LIST_HOLE 0xffffffffu
#define FMD_DOWN_WORD 4                    // word of a fast list's counter line that counts the SECOND-PASS list of that class (strands k_ovl_nei_grp moved to a smaller group); its deal counter: + FMD_DEAL_WORD
// A strand k_ovl_nei_fast hands on in the middle (a fork, an N: rounds only k_ovl_nei_grp does) travels with its state: bit 15 of the
// list entry's second word, the number of live candidates in the bits below, and in the strand's row of listB (scratch of
// k_ovl_nei, which these strands never reach without being started over from listA) one 32-byte entry per live candidate, in list
// order.  Every entry also carries the strand's own state (round, neighbours so far, lfork, info of the first neighbour): the lanes
// of a group read one entry each and nothing else.
//   a = { x1 lo, x1 bits 32..39 | round << 8 | n_nei << 24, D lo, D hi }     b = { r0 lo, r0 bits 32..39 | nei0 << 8, size | pos << 16, lfork (17 bits) | category << 17 | flags << 22 }
// (category, flags: a strand k_ovl_nei_grp<G> passes on to a larger group size because a round leaves it more than G candidates)
#define FMD_LIST_RESUME 0x8000u
__device__ __forceinline__ void fmd_resume_encode(uint4 *e, uint64_t x1, uint32_t sz, uint64_t D, uint64_t r0, uint32_t pos, uint32_t round, uint32_t n_nei, uint32_t nei0, uint32_t lf,
                                                  uint32_t cat = 0, uint32_t flags8 = 0)
{
    e[0] = make_uint4((uint32_t)x1, ((uint32_t)(x1 >> 32) & 0xffu) | round << 8 | n_nei << 24, (uint32_t)D, (uint32_t)(D >> 32));
    e[1] = make_uint4((uint32_t)r0, ((uint32_t)(r0 >> 32) & 0xffu) | nei0 << 8, sz | pos << 16, (lf & 0x1ffffu) | cat << 17 | (flags8 & 0xffu) << 22);
}
__device__ __forceinline__ bool fmd_resume_fits(uint32_t round, uint32_t n_nei, uint32_t nei0, uint32_t pos) { return round < 65536u && n_nei < 256u && nei0 < 65536u && pos < 65536u; }
__device__ __forceinline__ void fmd_resume_decode(const uint4 a, const uint4 b, uint64_t &x1, uint32_t &sz, uint64_t &D, uint64_t &r0, uint32_t &pos, uint32_t &round, uint32_t &n_nei,
                                                  uint32_t &nei0, uint32_t &lf, uint32_t &cat, uint32_t &flags8)
{
    x1 = (uint64_t)(a.y & 0xffu) << 32 | a.x; round = (a.y >> 8) & 0xffffu; n_nei = a.y >> 24; D = (uint64_t)a.w << 32 | a.z;
    r0 = (uint64_t)(b.y & 0xffu) << 32 | b.x; nei0 = b.y >> 8; sz = b.z & 0xffffu; pos = b.z >> 16; lf = b.w & 0x1ffffu; cat = (b.w >> 17) & 31u; flags8 = (b.w >> 22) & 0xffu;
}
#define FMD_LANE_CHUNK 64                  // the same for k_ovl_nei_lane (fmd_ovlp_lane.hip): every lane of a wave may hand its strand on in one step
#define FMD_DEAL_WORD_LANE 16              // word of a fast list's counter line that is k_ovl_nei_lane's ticket counter (the group form deals from the same word)
#define FMD_FAST_RESERVE (2 * FMD_FAST_MAX_WAVES * FMD_LANE_CHUNK)   // entries a general list may lose to holes (two fast kernels feed it; a wave's last chunk keeps at most FMD_LANE_CHUNK - 1)
#define FMD_CLS_PART_U32 (FMD_CLS_HEADER_U32 + 2 * FMD_GRP_CLASSES * FMD_FAST_RESERVE)   // per part of a pipelined batch: counters + that room
static_assert((3 * FMD_GRP_CLASSES + 1) * FMD_CLS_CNT_STRIDE <= FMD_CLS_HEADER_U32, "one counter line per list");
#define FMD_CLS_LISTS (3 * FMD_GRP_CLASSES + 1)                 // general lists, the slow list, fast lists (32-bit masks, 64-bit masks)
#define FMD_CLS_WORDS_PER_STRAND (6 * FMD_GRP_CLASSES + 3)      // two words per entry of a group list, one each for the slow list, the late slow list, the fix-up list
// Counters on the slow list's 128-byte line: [0] strands k_ovl_classify sets aside (k_ovl_nei takes them at once, beside the group
// kernels), [FMD_CLS_LATE_CNT] strands the fast / group kernels hand back later (a second k_ovl_nei launch behind them; this is the
// counter those kernels get as `slow_n`), and FMD_CLS_FIX_CNT words behind THAT the fix-up list's.
#define FMD_CLS_LATE_CNT 4
#define FMD_CLS_FIX_CNT 16
struct FmdOvlClasses {
    // counters: [k * STRIDE] = strands of general class k, [CLASSES * STRIDE] = the slow list, [(CLASSES + 1 + k) * STRIDE] = fast class k,
    // k >= CLASSES: the 64-bit variant of class k - CLASSES (+8 on that line: strands the fast kernel handed on to the general class)
    uint32_t *cnt;
    uint32_t *lst[FMD_GRP_CLASSES];   // two words per strand: index, candidates | length << 16
    uint32_t *lslow;                  // everything else, plus strands the group kernels hand back
    uint32_t *fast[2 * FMD_GRP_CLASSES];  // strands whose candidates are in the narrow form: k_ovl_nei_fast first (same two words)
};

// Two-pass walk of a sorted job (k_ovl_walk<WALK_HEAD / WALK_TAIL>, fmd_ovlp.hip): where a strand stands FMD_WALK_SPLIT bases in.
#define FMD_WALK_SPLIT 32u            // a multiple of 16: the stash of a parked strand is two whole 16-byte groups
struct FmdWalkPark {                  // k = ~0: the sequence ended inside the head (its record was written there)
    unsigned long long k, x0, x1, sz;  // LF row and bi-interval after FMD_WALK_SPLIT bases
    uint4 bases;                       // those bases, last base of the sequence first, 4 bits each (nt6 codes)
    uint4 pad;                         // (the row is one 64-byte line, written in one burst)
};
static_assert(sizeof(FmdWalkPark) == 64, "one line per parked strand");
