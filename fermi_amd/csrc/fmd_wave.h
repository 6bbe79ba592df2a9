// fmd_wave.h -- HBM layout of the FMD index and the per-wavefront rank engine (gfx950 only).
//
// HBM layout ("rank block"): the BWT string is cut into fixed runs of positions; a block holds the
// nt6 symbols of its run as three bit-planes per 32 positions plus the absolute symbol counts before
// it, so rank(k) -- rld_rank1a in the reference (rld.c:424) -- is ONE aligned fetch of the block k
// falls into plus masked popcounts: no frame lookup, no header walk (rld.c:352), no sequential
// Elias-delta decode (rld.h:77).
//
// A block is 64 bytes (random 64-byte gathers run 30-36 % more lines per second than 128-byte ones on MI355X,
// profiles/r1_bsearch/gather_probe.txt, and a line costs half the LDS landing space) = 4 x uint4:
//     u4[j], j = 0..2 = { p0, p1, p2, meta_j } for 32 positions each: bit i of p0/p1/p2 = bit 0/1/2 of the symbol
//     u4[3]           = { meta_3, meta_4, meta_5, meta_6 }
//     meta_0..4 = low 32 bits of the absolute count of $,A,C,G,T before the block,
//     meta_5 = bits 32..39 of the counts of $,A,C,G, meta_6 & 0xff = those of T; the count of N is what is left.
// Blocks OVERLAP: a block STARTS every 64 positions and holds the symbols of 96 -- its third chunk repeats the first chunk of the
// next block; the counts are those before position 64 b.  A rank pair (k, l) -- the two ends of an SA interval, l - k <= a few
// dozen once a search is past its first bases -- is answered from the ONE block of k whenever l < 64 b + 96, i.e. always for
// intervals up to 32 wide, where disjoint 96-position blocks needed a second line for a fraction size/96 of the steps (12-16 % of
// all lines of backward search and overlap discovery at 30x); the block of a position is a shift.  8 bits per symbol: 141 GB for
// the 1.4e11-symbol human-35x index, which is what 288 GB of HBM3E per GPU is for.  (Rounds 1-4 carried two more geometries behind
// macros -- 128-byte blocks of 256 positions, disjoint 64-byte blocks of 96 -- measured against this one in profiles/r1_blk64,
// r2_ab and r3_locality; they left the tree in round 5.)
//
// Wave engine: a wavefront owns 64 searches, one per lane.  Per step every lane posts up to two
// block numbers (k-side, l-side).  The 64 lanes then fetch those blocks COOPERATIVELY: in round
// r each 4-lane group g streams the block of its lane number r with one 16-byte
// global_load_lds_dwordx4 per lane (16 whole blocks per wave instruction, fully coalesced,
// LDS-DMA, no VGPR round trip).  After one s_waitcnt every lane reads ITS block back from LDS with
// ds_read_b128 and counts the symbols itself.  LDS is the transpose between "coalesced by line"
// and "one search per lane".
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define FMD_BLK_SYMS 96u        // positions whose symbols a block holds
#define FMD_BLK_U4 4            // uint4 per block
#define FMD_BLK_CHUNKS 3        // 32-position plane chunks per block
#define FMD_GRP_SHIFT 2         // 4 lanes x 16 B fetch one block
#define FMD_BLK_STRIDE 64u      // positions between the starts of consecutive blocks
#define FMD_BLK_OWN_CHUNKS 2    // chunks a block counts as its own (the rest is look-ahead)
#define FMD_BLK_BYTES (FMD_BLK_U4 * 16)
#define FMD_GRP_MASK ((1 << FMD_GRP_SHIFT) - 1)
#define FMD_BLK_PER_INST (64 >> FMD_GRP_SHIFT)   // blocks moved by one 64-lane global_load_lds
#define FMD_SLOT_U4 (64 * FMD_BLK_U4)            // dense slot: one block per lane
// blocks of the compacted l-side pool: as many bytes either way; with 64-byte blocks it holds one
// block per lane, so the dense fallback / two-phase step is never needed there
#define FMD_POOL_BLOCKS (32 * 8 / FMD_BLK_U4)
#define FMD_WAVE_LDS_U4 (2 * FMD_SLOT_U4 + FMD_POOL_BLOCKS / 4)   // 2 slots (+ ids): 16 KiB (8 KiB) per wave

// cache policy of the rank-block gather (global_load_lds aux bits: 0 = default, 2 = nt "stream")
#ifndef FMD_GLDS_AUX
#define FMD_GLDS_AUX 0
#endif

typedef __attribute__((address_space(3))) void fmd_lds_void;
typedef const __attribute__((address_space(1))) void fmd_glb_void;

struct FmdIndexView {            // passed by value as a kernel argument (lives in SGPRs)
    const uint4 *blocks;         // n_blocks x 8 uint4
    uint64_t cnt[7];             // C array: cnt[c] = # symbols < c (rld.c:282-284)
    uint64_t n_sym;              // mcnt[0]
    uint64_t n_seq;              // mcnt[1] = number of sentinels
    // SA intervals of all ACGT strings of ptab_d bases: entry = {k lo, k hi, l lo, l hi} (k > l: absent),
    // index of s_0 s_1 .. s_{d-1} = sum (s_j - 1) << 2(d-1-j).  Lets a search start ptab_d bases in.
    const uint4 *ptab;
    int ptab_d;
    // Where the LF-walk of every sequence stands after its last ptab_d bases (tail[id], id = the sequence's sentinel row):
    // row | the bases as a ptab index << (64 - 2 ptab_d) (40 at depth 12); ~0 = shorter than that, or not A/C/G/T.  fm_retrieve (exact.c:59) begins with
    // exactly these ptab_d dependent steps -- one DRAM line each, a ninth of all lines of overlap discovery -- for every
    // sequence, every time; they are taken once, when the index is loaded (8 bytes per sequence; nullptr = not built).
    const unsigned long long *tail;
    // Two-base blocks (fmd_pair.hip; nullptr = not built): 128 bytes per 64 positions, for the steps of a walk below min_match -- nothing is pushed
    // there, so a step may take TWO bases from one line (a 128-byte random line costs this memory system what a 64-byte one costs: profiles/r6_probe).
    // pair_tab[(block >> FMD_PAIR_SB_SHIFT) * 16 + pair]: where the pair's range starts + the pairs before that superblock (see fmd_pair_step).
    const uint4 *pair;
    const unsigned long long *pair_tab;
    // 64 counters on separate 128-byte lines: rank blocks requested from the memory system by the gathers.
    // Only the instrumented build (-DFMD_COUNT_LINES=1, libfmdhip_count.so) adds to them; bench.py runs one step
    // of each leg through that build to price the shipped kernels in DEVICE bytes (64 bytes per block).
    unsigned long long *stat;
};

#ifndef FMD_COUNT_LINES
#define FMD_COUNT_LINES 0
#endif
#define FMD_STAT_SLOTS 64
#define FMD_STAT_STRIDE 16   // u64 per slot = 128 bytes
// kind 0 = 64-byte rank blocks, kind 1 = other random lines (prefix-table look-ups), kind 2 = 128-byte two-base blocks
__device__ __forceinline__ void fmd_count_lines(const FmdIndexView &ix, int n, int kind = 0)
{
#if FMD_COUNT_LINES
    if (n > 0 && (threadIdx.x & 63) == 0) atomicAdd(ix.stat + (blockIdx.x & (FMD_STAT_SLOTS - 1)) * FMD_STAT_STRIDE + kind, (unsigned long long)n);
#else
    (void)ix; (void)n; (void)kind;
#endif
}

__device__ __forceinline__ int fmd_lane() { return (int)(threadIdx.x & 63); }
// number of set bits of a wave-uniform mask below this lane (v_mbcnt_lo/hi: two instructions)
__device__ __forceinline__ int fmd_below(uint64_t mask)
{
    return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// per-LANE form (divergent call sites: the prefix-table look-up of one lane)
__device__ __forceinline__ void fmd_count_lane(const FmdIndexView &ix, int n, int kind)
{
#if FMD_COUNT_LINES
    atomicAdd(ix.stat + (blockIdx.x & (FMD_STAT_SLOTS - 1)) * FMD_STAT_STRIDE + kind, (unsigned long long)n);
#else
    (void)ix; (void)n; (void)kind;
#endif
}

// position -> (block, offset inside the block)
__device__ __forceinline__ void fmd_word_split(uint64_t w, uint32_t &blk, uint32_t &ch) { blk = (uint32_t)(w >> 1); ch = (uint32_t)w & 1; } // the OWN chunk of word w
__device__ __forceinline__ void fmd_split(uint64_t k, uint32_t &blk, uint32_t &off) { blk = (uint32_t)(k >> 6); off = (uint32_t)k & 63; }
__device__ __forceinline__ uint32_t fmd_blk_of(uint64_t k) { uint32_t b, o; fmd_split(k, b, o); return b; }
// Can position p be read from the image of block blk (its own positions, or -- overlapped blocks -- the look-ahead chunk)?  off = its offset there.
__device__ __forceinline__ bool fmd_in_block(uint64_t p, uint32_t blk, uint32_t &off)
{
    const uint64_t o = p - (uint64_t)blk * FMD_BLK_STRIDE;     // wraps to a huge value when p lies before the block
    off = (uint32_t)o;
    return o < FMD_BLK_SYMS;
}
// uint4 index of 32-position word w in the block array (transcode kernels)
__device__ __forceinline__ uint64_t fmd_word_u4(uint64_t w) { uint32_t b, c; fmd_word_split(w, b, c); return (uint64_t)b * FMD_BLK_U4 + c; }

// XOR applied to the chunk index inside a block's LDS image so that the eight ds_read_b128 of a
// lane-owned block are bank-conflict free (the b128 lane groups are {0-3,12-15,20-27}, ... --
// MI355X_MICROARCH.md LDS table; lanes of one group whose blocks start on the same 128-B
// half-row get distinct 16-byte slots).
// 64-byte images: slot of chunk j of lane q = 4*((q>>2)&3) + (j ^ t) mod 16; the four quads of a b128
// lane group have distinct (q>>2)&3, so t = q & 3 separates the lanes of a quad.
__device__ __forceinline__ int fmd_chunk_xor(int q) { return q & 3; }

// One cooperative round: lane group g fetches the block of lane (g << FMD_GRP_SHIFT) + R for slot SLOT
// (8 lanes x 16 B for a 128-byte block, 4 lanes for a 64-byte one).
template <int SLOT, int R, int AUX = FMD_GLDS_AUX>
__device__ __forceinline__ void fmd_fetch_round(const FmdIndexView &ix, uint4 *lds, uint32_t blk, uint64_t need_mask)
{
    if ((need_mask >> R) & 0x1111111111111111ull) {          // wave-uniform: anybody in this round?
        const int lane = fmd_lane();
        const int j = lane & 3;
        const uint32_t sb = (uint32_t)__builtin_amdgcn_ds_swizzle((int)blk, (R << 5) | 0x1C); // blk of lane 4g+R
        if ((need_mask >> ((lane & ~3) | R)) & 1) {
            const uint4 *src = ix.blocks + (size_t)sb * FMD_BLK_U4 + (j ^ R);  // R = fmd_chunk_xor(4g+R)
            __builtin_amdgcn_global_load_lds((fmd_glb_void *)src, (fmd_lds_void *)(lds + SLOT * FMD_SLOT_U4 + R * 64), 16, 0, AUX);
        }
    }
}

// AUX = the cache-policy bits of this gather (a kernel whose lines are never asked for twice -- pass 1 of the sorted job -- may say nt: profiles/r6_probe)
template <int SLOT, int AUX = FMD_GLDS_AUX>
__device__ __forceinline__ void fmd_fetch_slot(const FmdIndexView &ix, uint4 *lds, uint32_t blk, bool need)
{
    const uint64_t m = __ballot(need);
    if (m == 0) return;
    fmd_count_lines(ix, __popcll(m));
    fmd_fetch_round<SLOT, 0, AUX>(ix, lds, blk, m); fmd_fetch_round<SLOT, 1, AUX>(ix, lds, blk, m);
    fmd_fetch_round<SLOT, 2, AUX>(ix, lds, blk, m); fmd_fetch_round<SLOT, 3, AUX>(ix, lds, blk, m);
}

__device__ __forceinline__ void fmd_fetch_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// ---- two-base blocks ------------------------------------------------------------------------------------------------------------------
// A pair block starts every 32 positions and describes 96 (so that ANY interval of up to 64 positions lies inside the block of its first position: a
// walk that has become narrow never needs a second line); per position p it holds BWT[p] (as the rank block does) and BWT[LF(p)] -- the base the
// walk would find one step later -- as three bit planes each, and for its 16 A/C/G/T pairs the number of positions before the block with that
// pair, relative to its superblock of 2^28 positions (28 bits each):
//     u4[j], j = 0..2     = { p0, p1, p2, s0 } of 32 positions: p = planes of BWT[p], s = planes of BWT[LF(p)] (0 where BWT[p] is not A/C/G/T)
//     u4[3 + j], j = 0..2 = { s1, s2, cw[2j], cw[2j + 1] }
//     u4[6], u4[7]        = cw[6 .. 13]
//     cw[0 .. 13] as one string of 448 bits: count of pair i = 4 (c1 - 1) + (c2 - 1) at bits [28 i, 28 i + 28)
// LF(LF(k)) for a row with BWT[k] = c1, BWT[LF(k)] = c2 is K2[c1][c2] + #{q <= k: pair(q) = (c1, c2)} - 1 with the constant
// K2 = cnt[c2] + #{c2 in BWT[0, cnt[c1])}; pair_tab holds K2 + the pairs before the superblock, the block the rest.
#define FMD_PAIR_U4 8
#define FMD_PAIR_BYTES 128
#define FMD_PAIR_STRIDE 32u                     // positions between the starts of consecutive pair blocks
#define FMD_PAIR_SB_SHIFT 23                    // blocks per superblock: 2^23 (2^28 positions)
#define FMD_PAIR_SLOT_U4 (64 * FMD_PAIR_U4)     // one image per lane: 8 KiB per wave
// chunk XOR of lane q's image: the sixteen lanes of a ds_read_b128 service group ({0-3,12-15,20-27}, ...) get the sixteen 16-byte slots of a 256-byte row
__device__ __forceinline__ int fmd_pair_xor(int q) { return (q & 3) | ((q >> 4) & 1) << 2; }
__device__ __forceinline__ int fmd_pair_base(int q) { return (q & 7) * 64 + (q >> 3) * FMD_PAIR_U4; }   // uint4 index of lane q's image in the slot
// round R: the 8-lane group g fetches the pair block of lane 8 g + R, 16 bytes per lane (8 blocks per wave instruction)
template <int R, int AUX>
__device__ __forceinline__ void fmd_pair_round(const FmdIndexView &ix, uint4 *slot, uint32_t blk, uint64_t need_mask)
{
    if ((need_mask >> R) & 0x0101010101010101ull) {
        const int lane = fmd_lane();
        const int j = lane & 7;
        const uint32_t sb = (uint32_t)__builtin_amdgcn_ds_swizzle((int)blk, (R << 5) | 0x18);   // blk of lane 8 g + R (lanes of one half-wave: bits 3, 4 kept)
        if ((need_mask >> ((lane & ~7) | R)) & 1) {
            const int x = (R & 3) | ((lane >> 4) & 1) << 2;                                      // fmd_pair_xor(8 g + R)
            const uint4 *src = ix.pair + (size_t)sb * FMD_PAIR_U4 + (j ^ x);
            __builtin_amdgcn_global_load_lds((fmd_glb_void *)src, (fmd_lds_void *)(slot + R * 64), 16, 0, AUX);
        }
    }
}
// every lane that `need`s one posts its block; no wait (the caller's next fmd_fetch_wait covers these loads too)
template <int AUX = FMD_GLDS_AUX>
__device__ __forceinline__ void fmd_pair_fetch(const FmdIndexView &ix, uint4 *slot, uint32_t blk, bool need)
{
    const uint64_t m = __ballot(need);
    if (m == 0) return;
    fmd_count_lines(ix, __popcll(m), 2);
    fmd_pair_round<0, AUX>(ix, slot, blk, m); fmd_pair_round<1, AUX>(ix, slot, blk, m); fmd_pair_round<2, AUX>(ix, slot, blk, m); fmd_pair_round<3, AUX>(ix, slot, blk, m);
    fmd_pair_round<4, AUX>(ix, slot, blk, m); fmd_pair_round<5, AUX>(ix, slot, blk, m); fmd_pair_round<6, AUX>(ix, slot, blk, m); fmd_pair_round<7, AUX>(ix, slot, blk, m);
}

// uint4 index (inside the wave's LDS area) of chunk 0^t of the block fetched for lane q, slot s
__device__ __forceinline__ int fmd_lds_base(int q, int slot)
{
    return slot * FMD_SLOT_U4 + (q & FMD_GRP_MASK) * 64 + (q >> FMD_GRP_SHIFT) * FMD_BLK_U4;
}

// chunk XOR of pool slot p: 16 consecutive slots use the 16 distinct (quad, chunk) 16-byte slot classes
__device__ __forceinline__ int fmd_pool_xor(int p) { return (p ^ (p >> 2)) & 3; }

// Pool of compacted blocks (ballot-prefix slots): `n` block ids in ids[], FMD_BLK_PER_INST per wave
// instruction, written to pool; the lane that owns pool slot p reads pool + p * FMD_BLK_U4 with XOR fmd_pool_xor(p).
template <int AUX = FMD_GLDS_AUX>
__device__ __forceinline__ void fmd_fetch_pool(const FmdIndexView &ix, uint4 *pool, const uint32_t *ids, int n)
{
    const int q = fmd_lane();
    fmd_count_lines(ix, n);
    for (int rr = 0; rr * FMD_BLK_PER_INST < n; ++rr) {
        const int slot = rr * FMD_BLK_PER_INST + (q >> FMD_GRP_SHIFT);
        if (slot < n) {
            const uint4 *src = ix.blocks + (size_t)ids[slot] * FMD_BLK_U4 + ((q & FMD_GRP_MASK) ^ fmd_pool_xor(slot));
            __builtin_amdgcn_global_load_lds((fmd_glb_void *)src, (fmd_lds_void *)(pool + rr * 64), 16, 0, AUX);
        }
    }
}

__device__ __forceinline__ uint32_t fmd_mask32(int rem) // low `rem` bits set, rem clamped to [0,32]
{
    const int r = rem < 0 ? 0 : (rem > 32 ? 32 : rem);      // v_med3_i32
    return (uint32_t)(1ull << r) - 1u;                      // r = 32: low word of 2^32 is 0 -> all ones
}

// Counts of all six symbols in BWT[0..k] from the lane's block image; npos = offset of k in its block + 1,
// blk = the block's number (the 64-byte layout derives the count of N from it).  WANT_SYM also
// returns BWT[k].
template <bool WANT_SYM>
__device__ __forceinline__ int fmd_block_rank6(const uint4 *blk, int t, uint32_t npos, uint64_t out[6], uint32_t blk_no)
{
    // five popcounts per chunk instead of six symbol masks: with x = |X|, y = |Y|, z = |Z|, xy = |X&Y|,
    // xz = |X&Z| over the counted positions (Y and Z are never set together: codes 6, 7 do not occur)
    //   T = z - xz, N = xz, G = xy, C = y - xy, A = x - xy - xz, $ = npos - (x + y + z - xy - xz)
    uint32_t cx = 0, cy = 0, cz = 0, cxy = 0, cxz = 0, meta[8];
    uint32_t s0 = 0, s1 = 0, s2 = 0;
    const uint32_t off = npos - 1;
#pragma unroll
    for (int c = 0; c < FMD_BLK_CHUNKS; ++c) {
        const uint4 v = blk[c ^ t];
        const uint32_t m = fmd_mask32((int)npos - 32 * c);
        const uint32_t xm = v.x & m;
        cx += __builtin_popcount(xm);
        cy += __builtin_popcount(v.y & m);
        cz += __builtin_popcount(v.z & m);
        cxy += __builtin_popcount(xm & v.y);
        cxz += __builtin_popcount(xm & v.z);
        meta[c] = v.w;
        if (WANT_SYM) {
            const bool here = (off >> 5) == (uint32_t)c;
            s0 = here ? v.x : s0; s1 = here ? v.y : s1; s2 = here ? v.z : s2;
        }
    }
    const uint32_t n5 = cxz, n4 = cz - cxz, n3 = cxy, n2 = cy - cxy, n1 = cx - cxy - cxz;
    const uint32_t n0 = npos - (n1 + n2 + n3 + n4 + n5); // positions past the BWT end are never counted
    {
        const uint4 mv = blk[3 ^ t];
        meta[3] = mv.x; meta[4] = mv.y; meta[5] = mv.z; meta[6] = mv.w;
    }
    const uint64_t b0 = ((uint64_t)(meta[5] & 0xff) << 32 | meta[0]), b1 = ((uint64_t)((meta[5] >> 8) & 0xff) << 32 | meta[1]);
    const uint64_t b2 = ((uint64_t)((meta[5] >> 16) & 0xff) << 32 | meta[2]), b3 = ((uint64_t)(meta[5] >> 24) << 32 | meta[3]);
    const uint64_t b4 = ((uint64_t)(meta[6] & 0xff) << 32 | meta[4]);
    out[0] = b0 + n0; out[1] = b1 + n1; out[2] = b2 + n2; out[3] = b3 + n3; out[4] = b4 + n4;
    out[5] = (uint64_t)blk_no * FMD_BLK_STRIDE - (b0 + b1 + b2 + b3 + b4) + n5;
    if (WANT_SYM) {
        const uint32_t bit = off & 31;
        return (int)(((s0 >> bit) & 1) | ((s1 >> bit) & 1) << 1 | ((s2 >> bit) & 1) << 2);
    }
    return 0;
}

// Count of ONE symbol c (per-lane value 0..5) in BWT[0..k]: what fm_backward_search needs
// (rld_rank11, rld.c:448).
__device__ __forceinline__ uint64_t fmd_block_rank1(const uint4 *blk, int t, uint32_t npos, int c, uint32_t blk_no)
{
    const uint32_t x0 = (c & 1) ? 0u : ~0u, x1 = (c & 2) ? 0u : ~0u, x2 = (c & 4) ? 0u : ~0u;
    uint32_t n = 0, lo = 0, hi = 0;
    uint32_t m0 = 0, m1 = 0, m2 = 0;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const uint4 v = blk[j ^ t];
        const uint32_t m = fmd_mask32((int)npos - 32 * j);
        n += __builtin_popcount((v.x ^ x0) & (v.y ^ x1) & (v.z ^ x2) & m);
        if (j == 0) m0 = v.w; else if (j == 1) m1 = v.w; else m2 = v.w;
    }
    const uint4 mv = blk[3 ^ t];
    if (c < 5) {
        lo = c == 0 ? m0 : c == 1 ? m1 : c == 2 ? m2 : c == 3 ? mv.x : mv.y;
        hi = c < 4 ? (mv.z >> (8 * c)) & 0xff : mv.w & 0xff;
        return ((uint64_t)hi << 32 | lo) + n;
    }
    // N: everything before the block that is none of the other five
    const uint64_t five = ((uint64_t)(mv.z & 0xff) << 32 | m0) + ((uint64_t)((mv.z >> 8) & 0xff) << 32 | m1) +
                          ((uint64_t)((mv.z >> 16) & 0xff) << 32 | m2) + ((uint64_t)(mv.z >> 24) << 32 | mv.x) + ((uint64_t)(mv.w & 0xff) << 32 | mv.y);
    return (uint64_t)blk_no * FMD_BLK_STRIDE - five + n;
}

// fmd_block_rank1 of symbol c AND of '$' from the same three chunk reads (the walk's window path and k_ovl_nei_fast want both).
__device__ __forceinline__ uint64_t fmd_block_rank1z(const uint4 *blk, int t, uint32_t npos, int c, uint32_t blk_no, uint64_t &rz)
{
    const uint32_t x0 = (c & 1) ? 0u : ~0u, x1 = (c & 2) ? 0u : ~0u, x2 = (c & 4) ? 0u : ~0u;
    uint32_t n = 0, nz = 0, lo = 0, hi = 0, m0 = 0, m1 = 0, m2 = 0;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const uint4 v = blk[j ^ t];
        const uint32_t m = fmd_mask32((int)npos - 32 * j);
        n += __builtin_popcount((v.x ^ x0) & (v.y ^ x1) & (v.z ^ x2) & m);
        nz += __builtin_popcount(~(v.x | v.y | v.z) & m);
        if (j == 0) m0 = v.w; else if (j == 1) m1 = v.w; else m2 = v.w;
    }
    const uint4 mv = blk[3 ^ t];
    rz = ((uint64_t)(mv.z & 0xff) << 32 | m0) + nz;
    if (c < 5) {
        lo = c == 0 ? m0 : c == 1 ? m1 : c == 2 ? m2 : c == 3 ? mv.x : mv.y;
        hi = c < 4 ? (mv.z >> (8 * c)) & 0xff : mv.w & 0xff;
        return ((uint64_t)hi << 32 | lo) + n;
    }
    const uint64_t five = ((uint64_t)(mv.z & 0xff) << 32 | m0) + ((uint64_t)((mv.z >> 8) & 0xff) << 32 | m1) +
                          ((uint64_t)((mv.z >> 16) & 0xff) << 32 | m2) + ((uint64_t)(mv.z >> 24) << 32 | mv.x) + ((uint64_t)(mv.w & 0xff) << 32 | mv.y);
    return (uint64_t)blk_no * FMD_BLK_STRIDE - five + n;
}

// ---- work queue of the persistent kernels --------------------------------------------------------
// Items are handed out by ONE device-wide counter.  A wave does not pay an atomic round trip (which
// crosses the fabric: the counter is shared by all XCDs) every time a lane finishes: it holds a
// small pool of tickets and reserves the next chunk as soon as it starts on the current one, so
// the atomic's latency hides behind ~FMD_TICKET_CHUNK finished searches.  Lanes that want an item
// get consecutive tickets by ballot prefix; a ticket >= n means the queue is drained.
#ifndef FMD_TICKET_CHUNK
#define FMD_TICKET_CHUNK 16
#endif
struct FmdTickets {
    uint32_t cur, end;   // wave-uniform: tickets [cur, end) are ours
    uint32_t nxt;        // lane 0: first ticket of the chunk reserved ahead (atomic may still be in flight)
    uint32_t nxt_size;   // wave-uniform: its size
    uint32_t chunk;      // largest chunk.  Atomics on ONE address serialise at ~14 ns each whatever the number of waves: a launch of
                         // 10^8 short items (k_ovl_walk<WALK_HEAD>: 20 steps each) in chunks of 16 is 6*10^6 atomics = 85 ms of
                         // counter time for 50 ms of work.  Kernels with many short items therefore take large chunks -- and, so that
                         // the end of the launch is not worked off by a few waves holding the last large chunks, guided ones: a chunk
                         // is 1/(2 * waves) of what is left of the n items when it is reserved, between FMD_TICKET_CHUNK and `chunk`
};                       // (guided chunks: the caller passes the number of items of the launch to every call; 0 = every chunk is `chunk`)

__device__ __forceinline__ uint32_t fmd_tickets_size(const FmdTickets &t, uint32_t at, size_t n_guided)
{
    if (n_guided == 0) return t.chunk;
    const uint32_t n = (uint32_t)n_guided, rem = n > at ? n - at : 0u, g = rem / (2u * gridDim.x);
    return g < FMD_TICKET_CHUNK ? FMD_TICKET_CHUNK : (g > t.chunk ? t.chunk : g);
}

// n_guided > 0: guided chunks over a launch of that many items (all waves of the grid take part)
__device__ __forceinline__ void fmd_tickets_init(FmdTickets &t, uint32_t *queue, uint32_t chunk = FMD_TICKET_CHUNK, size_t n_guided = 0)
{
    uint32_t v = 0;
    t.chunk = chunk;
    t.nxt_size = fmd_tickets_size(t, 64u * gridDim.x, n_guided);   // (where the queue stands once every wave has taken its first 64)
    if (fmd_lane() == 0) v = atomicAdd(queue, 64u + t.nxt_size); // one item per lane to start with + the chunk ahead
    v = (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
    t.cur = v; t.end = v + 64; t.nxt = v + 64;
}

// All 64 lanes call this together; returns the item of each lane that asked, (size_t)-1 otherwise.
__device__ __forceinline__ size_t fmd_tickets_take(FmdTickets &t, uint32_t *queue, bool want, size_t n_guided = 0)
{
    const uint64_t m = __ballot(want);
    if (m == 0) return (size_t)-1;
    const uint32_t p = (uint32_t)__popcll(m & ((1ull << fmd_lane()) - 1)), need = (uint32_t)__popcll(m);
    uint32_t served = 0;
    size_t res = (size_t)-1;
    for (;;) {
        const uint32_t avail = t.end - t.cur, k = avail < need - served ? avail : need - served;
        if (want && p - served < k) res = (size_t)t.cur + (p - served);
        t.cur += k; served += k;
        if (served == need) break;
        t.cur = (uint32_t)__builtin_amdgcn_readfirstlane((int)t.nxt); t.end = t.cur + t.nxt_size;
        t.nxt_size = fmd_tickets_size(t, t.end, n_guided);
        uint32_t v = 0;
        if (fmd_lane() == 0) v = atomicAdd(queue, t.nxt_size);
        t.nxt = v;
    }
    return res;
}

// The per-wave LDS area.  Kernels are launched with 64-thread workgroups (one wave each), so no
// workgroup barrier is ever needed: the wave is its own synchronisation domain.
#define FMD_DECLARE_WAVE_LDS() __shared__ uint4 fmd_lds[FMD_WAVE_LDS_U4]

// rank2: every lane posts k and l (positions, either may be UINT64_MAX = "none").
// Fetches both sides (the l-side only when it lives in another block), then returns the lane's
// LDS block pointers.  All 64 lanes must call this together.
//
// The k side is a dense slot (one block per lane, 8 cooperative rounds).  The l side is needed by
// few lanes once the SA intervals are narrow (the two ends of [k, l] usually share a block), so
// those lanes are compacted with a ballot prefix into a 32-block pool and fetched 8 per wave
// instruction: at most 4 instructions instead of 8 rounds of swizzle + address arithmetic.  Wide
// intervals (every lane straddles: the first few bases of a search) fall back to a dense l slot.
// overlapped blocks: the l side of a rank pair is read from the block of the k side whenever that block reaches it
__device__ __forceinline__ void fmd_l_from_k(bool both, uint64_t l, uint32_t blk_k, uint32_t &blk_l, uint32_t &off_l)
{
    uint32_t o;
    if (both && blk_l != blk_k && fmd_in_block(l, blk_k, o)) { blk_l = blk_k; off_l = o; }
}

struct FmdRank2 {
    const uint4 *bk, *bl;  // lane-owned block images in LDS
    int t, tl;             // chunk XOR of each image
    uint32_t nk, nl;       // positions to count in each
    uint32_t blk_k, blk_l; // block numbers
    bool hk, hl;           // side present
};

__device__ __forceinline__ FmdRank2 fmd_wave_rank2_fetch(const FmdIndexView &ix, uint4 *lds, uint64_t k, uint64_t l)
{
    const int q = fmd_lane();
    FmdRank2 r;
    r.hk = k != ~0ull; r.hl = l != ~0ull;
    uint32_t ok_, ol_;
    fmd_split(k, r.blk_k, ok_); fmd_split(l, r.blk_l, ol_);
    fmd_l_from_k(r.hk && r.hl, l, r.blk_k, r.blk_l, ol_);
    const bool l_sep = r.hl && !(r.hk && r.blk_k == r.blk_l);
    fmd_fetch_slot<0>(ix, lds, r.blk_k, r.hk);
    r.t = fmd_chunk_xor(q);
    r.bk = lds + fmd_lds_base(q, 0);
    r.bl = r.bk; r.tl = r.t;
    const uint64_t m = __ballot(l_sep);
    if (m) {
        const int n_sep = __popcll(m);
        if (n_sep <= FMD_POOL_BLOCKS) { // compact pool in slot 1; block ids after the two slots
            uint4 *pool = lds + FMD_SLOT_U4;
            uint32_t *ids = (uint32_t *)(lds + 2 * FMD_SLOT_U4);
            const int p = fmd_below(m);
            if (l_sep) ids[p] = r.blk_l;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            fmd_fetch_pool(ix, pool, ids, n_sep);
            if (l_sep) { r.bl = pool + p * FMD_BLK_U4; r.tl = fmd_pool_xor(p); }
        } else {
            fmd_fetch_slot<1>(ix, lds, r.blk_l, l_sep);
            if (l_sep) r.bl = lds + fmd_lds_base(q, 1);
        }
    }
    r.nk = ok_ + 1; r.nl = ol_ + 1;
    fmd_fetch_wait();
    return r;
}


// ---- compact engine: dense slot + pool (12.4 KiB of LDS per wave with 128-byte blocks: 13 waves per
// CU instead of 10; 6.3 KiB with 64-byte blocks) ----------------------------------------------------
// More waves = more lines in flight, which is what these latency-bound searches need (DESIGN.md
// section 4).  Dense slot for the k side + the 32-block pool for the l side.  When more than 32
// lanes straddle (only while the SA intervals are still wide: the first ~log4(n) bases, whose
// blocks sit in L2) the step becomes two-phase: the caller consumes the k side, calls
// fmd_wave_l_ready(), which re-uses the dense slot for the l blocks, then consumes the l side.
#define FMD_COMPACT_LDS_U4 (FMD_SLOT_U4 + FMD_POOL_BLOCKS * FMD_BLK_U4 + FMD_POOL_BLOCKS / 4)
#define FMD_DECLARE_COMPACT_LDS() __shared__ uint4 fmd_lds[FMD_COMPACT_LDS_U4]

struct FmdRank2c {
    const uint4 *bk, *bl;
    int t, tl;
    uint32_t nk, nl, blk_k, blk_l;
    bool hk, hl, l_sep;
    bool two_phase;        // wave-uniform: bl is not valid until fmd_wave_l_ready()
};

template <int AUX = FMD_GLDS_AUX>
__device__ __forceinline__ FmdRank2c fmd_wave_rank2_fetch_compact(const FmdIndexView &ix, uint4 *lds, uint64_t k, uint64_t l)
{
    const int q = fmd_lane();
    FmdRank2c r;
    r.hk = k != ~0ull; r.hl = l != ~0ull;
    uint32_t ok_, ol_;
    fmd_split(k, r.blk_k, ok_);
    fmd_split(l, r.blk_l, ol_);
    fmd_l_from_k(r.hk && r.hl, l, r.blk_k, r.blk_l, ol_);
    r.l_sep = r.hl && !(r.hk && r.blk_k == r.blk_l);
    fmd_fetch_slot<0, AUX>(ix, lds, r.blk_k, r.hk);
    r.t = fmd_chunk_xor(q);
    r.bk = lds + fmd_lds_base(q, 0);
    r.bl = r.bk; r.tl = r.t;
    r.two_phase = false;
    const uint64_t m = __ballot(r.l_sep);
    if (m) {
        const int n_sep = __popcll(m);
        if (n_sep <= FMD_POOL_BLOCKS) {
            uint4 *pool = lds + FMD_SLOT_U4;
            uint32_t *ids = (uint32_t *)(pool + FMD_POOL_BLOCKS * FMD_BLK_U4);
            const int p = fmd_below(m);
            if (r.l_sep) ids[p] = r.blk_l;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            fmd_fetch_pool<AUX>(ix, pool, ids, n_sep);
            if (r.l_sep) { r.bl = pool + p * FMD_BLK_U4; r.tl = fmd_pool_xor(p); }
        } else r.two_phase = true;
    }
    r.nk = ok_ + 1; r.nl = ol_ + 1;
    fmd_fetch_wait();
    return r;
}

// Second phase of a two-phase step (no-op otherwise).  All 64 lanes together, after every lane
// has finished reading its k-side image: lanes whose l side lives in another block get it in the
// dense slot; the others keep their k block, which is also their l block.
template <int AUX = FMD_GLDS_AUX>
__device__ __forceinline__ void fmd_wave_l_ready(const FmdIndexView &ix, uint4 *lds, FmdRank2c &r)
{
    if (!r.two_phase) return;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // k-side LDS reads have returned
    fmd_fetch_slot<0, AUX>(ix, lds, r.blk_l, r.l_sep);
    fmd_fetch_wait();
    r.bl = r.bk; r.tl = r.t;
}
