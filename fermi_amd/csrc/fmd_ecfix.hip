// fmd_ecfix.hip -- the correction pass of `fermi correct` (ec_fix1 / ec_fix, correct.c:121-256) on the GPU.
//
// The reference corrects a read by a best-first search over partial paths: a path ends at some base of the read,
// carries the k-mer that ends there, a score, and a back-pointer into a trace of the choices made.  The next path
// to extend is the one with the smallest key
//        score (16 bits) | trace index (32) | bases still ahead (16)          compared as a SIGNED 64-bit number
// (correct.c:104, ku128_ylt mag.c:22).  Every path has its own trace index, so the order is total: any priority
// queue pops the paths in the reference's order, and the result is a function of (read, qualities, table) alone.
// That makes a read an independent work item -- one lane each, two passes (reverse-complement strand, then forward:
// correct.c:237-243), refilled from a ticket queue as lanes finish (reads with errors take 10-100x the steps of clean
// ones).  A lane's queue (<= 256 paths, correct.c:114) and trace live in its slice of an HBM work area; the table is
// a device hash table with one 8-byte slot per solid k-mer: a look-up is ONE random 8-byte load where the host table
// (sorted buckets) needs a search and the reference's khash two dependent ones.
//
// Nothing here touches the FMD index: the table is the output of fmd_kmer_collect_dev (fmd_kmer.hip).
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "fmd_kernel_common.h"

// correct.c:112-119
#define EC_RATIO_FACTOR 10
#define EC_DIFF_FACTOR 13
#define EC_MAX_HEAP 256
#define EC_MAX_SC_DIFF 60
#define EC_MAX_QUAL 40
#define EC_MISS_PENALTY 10
#define EC_MIN_OCC 5
#define EC_MIN_OCC_RATIO 0.8
#define EC_HEAP_SLOTS 264          // 256 + the slack of one expansion
#define EC_INFO_TRACE_FULL 0x80000000u

struct fmd_ectab {
    int device, w, suf_len;
    uint64_t n_slots;              // power of two
    uint64_t *slots;               // device: kmer << 10 | val << 2 | best base, or EC_EMPTY
    uint32_t *queue;               // device: ticket counter of the persistent kernel (+ EC_FULL_FLAG)
    void *buf[5]; size_t buf_bytes[5]; int buf_busy;   // device buffers of fmd_ecfix_batch, kept between calls (EcBuf)
};
#define EC_EMPTY (~0ull)
#ifndef EC_NT_TABLE
#define EC_NT_TABLE 0
#endif
// What a lane keeps in LDS decides how many waves a CU holds (160 KiB), and this kernel lives on resident waves (profiles/r6_ecfix): three knobs.
#ifndef EC_STAGE_QUAL  // 1: the qualities of a staged read wait in LDS beside its bases (8 KiB per wave)
#define EC_STAGE_QUAL 0
#endif
#ifndef EC_TR_LINE     // trace entries per piece written / read at once through an LDS piece per lane: 0 = entry by entry, 4 = 32-byte pieces, 8 = whole 64-byte lines
#define EC_TR_LINE 4
#endif
#ifndef EC_LDS_H       // entries of the lane's queue (a binary heap) that live in LDS, the rest in its slice of HBM
#define EC_LDS_H 4
#endif
#ifndef EC_QWIN        // 1: the lane keeps the 16 aligned bytes of qualities around the last position it asked for in registers (a strand is searched position by position)
#define EC_QWIN 1
#endif
#ifndef EC_HOP_LOOP
#define EC_HOP_LOOP 1      // the hop chains in a loop of their own (below); 0: every hop a turn of the general form
#endif
#ifndef EC_HOP_MIN
#define EC_HOP_MIN 16      // ... which runs (once at least, if anybody hops) while at least this many lanes of the wave are hopping: 1 -> 218 ms (a last hopping lane holds 63 up), 8 -> 173, 16 -> 160, 24 -> 162, 32 and more -> 172 (profiles/r6_hop)
#endif
#ifndef EC_HOP_ONLY
#define EC_HOP_ONLY 1      // 1: the general turn holds no hop at all (a hopping lane sits it out; the loop gives every one of them a hop per turn at least)
#endif
#ifndef EC_PUSH_AB
#define EC_PUSH_AB 1       // 1: a turn's pushes through two shared calls
#endif
#ifndef EC_GATE
#define EC_GATE 1          // taking reads in / seeding strands / closing strands wait until EC_GATE_IN / EC_GATE_CL lanes of the wave want to (k_ecfix)
#endif
#ifndef EC_GATE_IN
#define EC_GATE_IN 16      // 0 (no gates): 143.8 ms per 5*10^7 reads; 4: 130.4; 8: 117.9; 16: 114.2; 24: 122.1 (profiles/r6_hop/ab_gate.txt)
#endif
#ifndef EC_GATE_CL
#define EC_GATE_CL 8
#endif
#ifndef EC_HOP_WARM
#define EC_HOP_WARM 0      // 1: the hop loop asks for the NEXT hop's table line beside this hop's (speculation: most hops stand)
#endif
#ifndef EC_HOP_BATCH   // 1: the bases a hop passes over are taken from the lane's LDS words in one piece, not one LDS read and one 64-bit shift per base
#define EC_HOP_BATCH 1
#endif
#define EC_STAGE (EC_TR_LINE > 0)
// One triple encodes to EC_EMPTY itself: the 27-mer of 27 Ts (54 one bits) with the largest packed depths (255) and best base T.
// k_ectab_fill does not store it -- it would read as an empty slot and the solid k-mer as a miss -- but raises queue[EC_FULL_FLAG],
// and a look-up of that k-mer that runs into an empty slot answers from the flag.
#define EC_FULL_FLAG 1
// queue + EC_STAT_U32 (as u64[3]): the instrumented build's counters (-DFMD_COUNT_LINES=1, libfmdhip_count.so; fmd_ectab_line_count): table slots
// probed (8 bytes each), queue entries read or written (16 bytes each), trace entries read or written (8 bytes each)
#define EC_STAT_U32 4
struct EcCount {
#if FMD_COUNT_LINES
    unsigned long long c[3];
    __device__ __forceinline__ EcCount() { c[0] = c[1] = c[2] = 0; }
    __device__ __forceinline__ void add(int kind, int n = 1) { c[kind] += (unsigned long long)n; }
    __device__ __forceinline__ void flush(uint32_t *queue) { for (int k = 0; k < 3; ++k) if (c[k]) atomicAdd((unsigned long long *)(queue + EC_STAT_U32) + k, c[k]); }
#else
    __device__ __forceinline__ void add(int, int = 1) {}
    __device__ __forceinline__ void flush(uint32_t *) {}
#endif
};

__device__ __forceinline__ uint64_t ec_hash(uint64_t x)   // splitmix64 finaliser: the k-mers of a genome are anything but uniform
{
    x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ull;
    x ^= x >> 27; x *= 0x94d049bb133111ebull;
    return x ^ (x >> 31);
}

// The table is bucketed by 64-byte lines of eight slots: a k-mer goes into the first free slot of its home line, the next line when that one is full.
// Nothing is ever removed, so a line with a free slot has never been full: a look-up that finds neither the k-mer nor a free slot in a line goes on to the
// next, anything else ends there -- one line, fetched as four 16-byte loads in flight together, for all but the ~1 % of the look-ups whose home line is
// full (load <= 1/2), where linear probing over single slots made 1.7 dependent round trips per look-up, and the slowest lane's chain is the wave's.
#define EC_LINE 8
__global__ void k_ectab_fill(uint64_t n, int suf_len, const uint32_t *__restrict__ bucket, const uint32_t *__restrict__ key, const uint8_t *__restrict__ val,
                             uint64_t *__restrict__ slots, uint64_t mask, uint32_t *__restrict__ full_flag)
{
    const uint64_t step = (uint64_t)gridDim.x * blockDim.x, lmask = mask >> 3;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += step) {
        // the k-mer a look-up will present (correct.c:156-157): bucket = its low 2*suf_len bits, key >> 2 = the rest
        const uint64_t x = (uint64_t)(key[i] >> 2) << (2 * suf_len) | bucket[i];
        const uint64_t e = x << 10 | (uint64_t)val[i] << 2 | (key[i] & 3);
        if (e == EC_EMPTY) { *full_flag = 1u; continue; }
        uint64_t ln = ec_hash(x) & lmask;
        for (bool placed = false; !placed; ln = (ln + 1) & lmask)
            for (int k = 0; k < EC_LINE && !placed; ++k)
                placed = atomicCAS((unsigned long long *)(slots + ln * EC_LINE + k), (unsigned long long)EC_EMPTY, (unsigned long long)e) == EC_EMPTY;
    }
}

// kh_get(solid, h, key) of correct.c:156-157: -1, or val << 2 | best base
__device__ __forceinline__ int ec_lookup(const uint64_t *__restrict__ slots, uint64_t mask, uint64_t x, bool full, EcCount &C)
{
    const uint64_t lmask = mask >> 3;
    uint64_t ln = ec_hash(x) & lmask;
    for (;;) {
        const uint4 *q = (const uint4 *)(slots + ln * EC_LINE);
#if EC_NT_TABLE   // a table line is asked for once: "nt" keeps it from pushing the lanes' own reads out of the caches (profiles/r6_ecfix)
        typedef uint32_t ec_u32x4 __attribute__((ext_vector_type(4)));
        const ec_u32x4 a_ = __builtin_nontemporal_load((const ec_u32x4 *)q), b_ = __builtin_nontemporal_load((const ec_u32x4 *)q + 1),
                       c_ = __builtin_nontemporal_load((const ec_u32x4 *)q + 2), d_ = __builtin_nontemporal_load((const ec_u32x4 *)q + 3);
        const uint4 a = make_uint4(a_.x, a_.y, a_.z, a_.w), b = make_uint4(b_.x, b_.y, b_.z, b_.w), c = make_uint4(c_.x, c_.y, c_.z, c_.w), d = make_uint4(d_.x, d_.y, d_.z, d_.w);
#else
        const uint4 a = q[0], b = q[1], c = q[2], d = q[3];
#endif
        C.add(0, EC_LINE);
        int hit = -2;
        bool free_slot = false;
        // (slot by slot in registers: as an array of eight the compiler has put the line into scratch)
#define EC_SLOT(lo_, hi_) do { const uint64_t e_ = (uint64_t)(hi_) << 32 | (lo_); if ((e_ >> 10) == x && e_ != EC_EMPTY) hit = (int)(e_ & 0x3ff); free_slot |= e_ == EC_EMPTY; } while (0)
        EC_SLOT(a.x, a.y); EC_SLOT(a.z, a.w); EC_SLOT(b.x, b.y); EC_SLOT(b.z, b.w); EC_SLOT(c.x, c.y); EC_SLOT(c.z, c.w); EC_SLOT(d.x, d.y); EC_SLOT(d.z, d.w);
#undef EC_SLOT
        if (hit >= 0) return hit;
        if (free_slot) return (full && x == (EC_EMPTY >> 10)) ? 0x3ff : -1;
        ln = (ln + 1) & lmask;
    }
}

struct EcNode { uint64_t x; int64_t y; };

// ---- where a lane keeps what it works on -------------------------------------------------------------------------------------------
// The bases of a read of at most EC_LDS_BASES bases are staged in LDS when the lane takes it, 4 bits each, in read order (word w of lane l at
// [w * 64 + l]: lanes side by side, no bank conflicts when they ask for the same word).  The first EC_LDS_H entries of the lane's queue -- a binary
// heap: the top is what every expansion reads -- live in LDS as well; a read with a few errors never has more paths than that (per error: the path
// that leaves the read there and the path that stays), so only error-ridden reads ever touch the slice of HBM behind it.  A quality is needed once
// per expansion, of a position known before the table is asked: its byte is loaded beside the table's line, and it rides in the trace entry for the
// walk back (correct.c:213-218 reads it again there).  What is left for HBM per turn: the table's line (the point of the exercise), that byte, one
// 8-byte store to the trace.  Round 4's kernel kept all of it in HBM behind byte loads -- ~17 dependent round trips per expansion, 409 ms per 5*10^7 reads.
#define EC_LDS_BASES 128
#define EC_LDS_BW (EC_LDS_BASES / 8)          // words of bases per lane
#define EC_LDS_QW (EC_LDS_BASES / 4)          // words of quality bytes per lane
#define EC_LDS_BYTES (EC_LDS_H * 64 * 16 + EC_LDS_BW * 64 * 4 + (EC_STAGE_QUAL ? EC_LDS_QW * 64 * 4 : 0) + EC_TR_LINE * 64 * 8)   // 4 + 4 + 0 + 2 = 10 KiB per wave: 16 waves per CU

struct EcHeap {                                // entry k of the lane's queue
    uint4 *lds; uint4 *hbm;                    // lds + k * 64 for k < EC_LDS_H, hbm + k beyond
    __device__ __forceinline__ uint4 get(uint32_t k, EcCount &C) const { if (k < EC_LDS_H) return lds[k * 64]; C.add(1); return hbm[k]; }
    __device__ __forceinline__ void put(uint32_t k, const uint4 v, EcCount &C) const { if (k < EC_LDS_H) lds[k * 64] = v; else { C.add(1); hbm[k] = v; } }
};
__device__ __forceinline__ int64_t ec_key(const uint4 v) { return (int64_t)((uint64_t)v.w << 32 | v.z); }

// binary min-heap on y (any queue gives the reference's order: the keys are distinct)
__device__ __forceinline__ void ec_push(const EcHeap &H, uint32_t &hn, uint64_t x, int64_t y, EcCount &C)
{
    uint32_t k = hn++;
    while (k) {
        const uint32_t p = (k - 1) >> 1;
        const uint4 v = H.get(p, C);
        if (ec_key(v) <= y) break;
        H.put(k, v, C); k = p;
    }
    H.put(k, make_uint4((uint32_t)x, (uint32_t)(x >> 32), (uint32_t)(uint64_t)y, (uint32_t)((uint64_t)y >> 32)), C);
}
__device__ __forceinline__ EcNode ec_pop(const EcHeap &H, uint32_t &hn, EcCount &C)
{
    const uint4 top = H.get(0, C);
    EcNode r; r.x = (uint64_t)top.y << 32 | top.x; r.y = ec_key(top);
    const uint4 last = H.get(--hn, C);
    const int64_t ly = ec_key(last);
    uint32_t i = 0;
    for (;;) {
        uint32_t c = 2 * i + 1;
        if (c >= hn) break;
        uint4 cv = H.get(c, C);
        int64_t cy = ec_key(cv);
        if (c + 1 < hn) {
            const uint4 dv = H.get(c + 1, C);
            const int64_t dy = ec_key(dv);
            if (dy < cy) { cv = dv; cy = dy; ++c; }
        }
        if (ly <= cy) break;
        H.put(i, cv, C); i = c;
    }
    if (hn) H.put(i, last, C);
    return r;
}

// One read, seen from one strand.  Pass 0 works on the reverse complement with the qualities reversed (correct.c:237-238):
// logical position i is byte len-1-i, bases complemented; pass 1 is the read as stored.
struct EcRead {
    uint8_t *s, *q; int len; bool rc, staged;
    uint32_t *lb;                              // the lane's words of bases in LDS (staged reads)
    uint32_t *lq;                              // EC_STAGE_QUAL: the lane's words of quality bytes in LDS, in read order
    __device__ __forceinline__ int at(int i) const { return rc ? len - 1 - i : i; }
    __device__ __forceinline__ int base(int i) const
    {
        const int j = at(i);
        const int c = staged ? (int)((lb[(j >> 3) * 64] >> (4 * (j & 7))) & 0xfu) : (int)s[j];
        return rc ? comp6(c) : c;
    }
    __device__ __forceinline__ void set_base(int i, int c)
    {
        const int j = at(i);
        const uint32_t v = (uint32_t)(rc ? comp6(c) : c);
        s[j] = (uint8_t)v;
        if (staged) { uint32_t *w = lb + (j >> 3) * 64; *w = (*w & ~(0xfu << (4 * (j & 7)))) | v << (4 * (j & 7)); }
    }
#if EC_STAGE_QUAL
    __device__ __forceinline__ int qual(int i) const { const int j = at(i); return staged ? (int)((lq[(j >> 2) * 64] >> (8 * (j & 3))) & 0xffu) : (int)q[j]; }
    __device__ __forceinline__ void set_qual(int i, int v)
    {
        const int j = at(i);
        q[j] = (uint8_t)v;
        if (staged) { uint32_t *w = lq + (j >> 2) * 64; *w = (*w & ~(0xffu << (8 * (j & 3)))) | (uint32_t)(v & 0xff) << (8 * (j & 3)); }
    }
#else
#if EC_QWIN
    // The quality of position i is asked for once per expansion, of a position a few bases from the last one: the 16 aligned bytes around it stay in four
    // registers, and three accesses in four are answered from them.  (Byte by byte every access paid a 64-byte line that the table's lines had pushed out of
    // the caches since the last one: 155 GB of table lines pass through them per 5*10^7 reads.  profiles/r6_ecfix.)
    // (the window lives in the kernel's own registers, EcQwin: as members of this struct the whole struct went to scratch)
    __device__ __forceinline__ int qual(int i) const { return (int)q[at(i)]; }
    __device__ __forceinline__ void set_qual(int i, int v) { q[at(i)] = (uint8_t)v; }
#else
    __device__ __forceinline__ int qual(int i) const { return (int)q[at(i)]; }
    __device__ __forceinline__ void set_qual(int i, int v) { q[at(i)] = (uint8_t)v; }
#endif
#endif
};
struct EcQwin { uint32_t w0, w1, w2, w3; uint64_t tag; };
__device__ __forceinline__ int ec_qual(const EcRead &r, int i, EcQwin &W)
{
#if EC_QWIN && !EC_STAGE_QUAL
    const uint64_t a = (uint64_t)(uintptr_t)(r.q + r.at(i)), t = a & ~15ull;
    if (t != W.tag) { const uint4 v = *(const uint4 *)(uintptr_t)t; W.w0 = v.x; W.w1 = v.y; W.w2 = v.z; W.w3 = v.w; W.tag = t; }
    const uint32_t o = (uint32_t)a & 15u, wv = (o >> 2) == 0 ? W.w0 : (o >> 2) == 1 ? W.w1 : (o >> 2) == 2 ? W.w2 : W.w3;
    return (int)((wv >> (8 * (o & 3))) & 0xffu);
#else
    return r.qual(i);
#endif
}
// The read into the lane's LDS words: aligned 4-byte loads funnel-shifted into place, whatever the alignment of its first byte; a word is loaded only if
// it holds a byte of the read (bytes past the read's end inside its last word: whatever follows, never looked at).
__device__ __forceinline__ void ec_stage_words(const uint8_t *p, int len, uint32_t *dst, bool nibbles)
{
    const uint32_t *al = (const uint32_t *)((uintptr_t)p & ~(uintptr_t)3);
    const uint32_t sh = 8u * (uint32_t)((uintptr_t)p & 3);
    const uint32_t *last = (const uint32_t *)((uintptr_t)(p + len - 1) & ~(uintptr_t)3);   // the word of the read's last byte
    const int nw = (len + 3) >> 2;
    uint32_t carry = al[0], acc = 0;
    for (int w = 0; w < nw; ++w) {
        const uint32_t nxt = al + w + 1 <= last ? al[w + 1] : 0u;
        const uint32_t d = sh ? __builtin_amdgcn_alignbit(nxt, carry, sh) : carry;
        carry = nxt;
        if (!nibbles) dst[w * 64] = d;
        else {
            uint32_t t = (d | d >> 4) & 0x00ff00ffu; t = (t | t >> 8) & 0xffffu;   // four bases -> four nibbles
            if (w & 1) dst[(w >> 1) * 64] = acc | t << 16; else acc = t;
        }
    }
    if (nibbles && (nw & 1)) dst[(nw >> 1) * 64] = acc;
}
__device__ __forceinline__ void ec_stage(EcRead &r)
{
    r.staged = r.len > 0 && r.len <= EC_LDS_BASES;
    if (!r.staged) return;
    ec_stage_words(r.s, r.len, r.lb, true);
#if EC_STAGE_QUAL
    ec_stage_words(r.q, r.len, r.lq, false);
#endif
}

// The trace of a lane (correct.c:104: one entry per path, each naming its parent) is written in order and read back along ONE chain of parents.  Entry by
// entry that is an 8-byte access to a line of the lane's own (64 lanes, 64 lines per wave instruction: every entry pays a 64-byte line -- 92 GB of traffic for
// 11.5 GB of entries on 5*10^7 reads, profiles/r5_final).  EC_STAGE: the open line -- EC_TR_LINE entries -- waits in LDS (entry e of lane l at [e * 64 + l]) and
// leaves whole; the walk back loads the line of the entry it wants into the same LDS line and takes every entry of its chain that the line holds.
struct EcTrace {
    uint64_t *hbm;                             // the lane's slice
    uint64_t *ln;                              // its LDS line (lds + lane)
    uint32_t tag;                              // walk back: the line of the slice the LDS line holds
    __device__ __forceinline__ void put(uint32_t t, uint64_t v, EcCount &C)
    {
#if EC_STAGE
        ln[(t & (EC_TR_LINE - 1)) * 64] = v;
        if ((t & (EC_TR_LINE - 1)) == EC_TR_LINE - 1) flush(t, C);
#else
        hbm[t] = v; C.add(2);
#endif
    }
    __device__ __forceinline__ void flush(uint32_t t, EcCount &C)     // the line that holds entry t, whole
    {
#if EC_STAGE
        uint4 *dst = (uint4 *)(hbm + (t & ~(uint32_t)(EC_TR_LINE - 1)));
#pragma unroll
        for (int e = 0; e < (EC_TR_LINE > 0 ? EC_TR_LINE : 2); e += 2) {
            const uint64_t a = ln[e * 64], b = ln[(e + 1) * 64];
            dst[e >> 1] = make_uint4((uint32_t)a, (uint32_t)(a >> 32), (uint32_t)b, (uint32_t)(b >> 32));
        }
        C.add(2, EC_TR_LINE);
#endif
    }
};

// a new path: `par` extended by base code c (0..3; an N in the read is followed as A, correct.c:101) at a cost
// (qv: the quality byte of the position the path steps over -- the walk back wants it, correct.c:215-217, and takes it from the trace entry)
__device__ __forceinline__ bool ec_branch(const EcHeap &H, uint32_t &hn, EcTrace &trace, uint32_t &tn, uint32_t trace_cap, const EcNode &par, int c, int cost,
                                          int shift, int has_match, int qv, EcCount &C)
{
    if (tn >= trace_cap) return false;
    if (cost < 0) cost = 0;
    if (c >= 4) c = 0;
    const uint64_t py = (uint64_t)par.y, left = (py & 0xffff) - 1;
    const uint64_t y = ((py >> 48) + (uint64_t)cost) << 48 | (uint64_t)tn << 16 | left;
    trace.put(tn, ((uint64_t)(uint32_t)qv << 16 | left) << 32 | (uint64_t)((uint32_t)c << 29 | (uint32_t)has_match << 28 | (uint32_t)(py >> 16)), C); // quality, position | base | matched | parent
    ++tn;
    ec_push(H, hn, (uint64_t)c << shift | par.x >> 2, (int64_t)y, C);
    return true;
}

// price of leaving the read's base for the table's best one, from the packed depth byte (correct.c:164-171)
__device__ __forceinline__ int ec_swap_penalty(int v)
{
    const int rest = v & 7, best = rest ? rest * (v >> 3) : v >> 3;
    int p = (best - rest) * EC_DIFF_FACTOR;
    if (best - rest < 1) p = 1;
    const int by_ratio = rest ? (v >> 3) * EC_RATIO_FACTOR : 10000, by_rest = (7 - rest) * EC_DIFF_FACTOR;
    p = p < by_ratio ? p : by_ratio;
    p = p < by_rest ? p : by_rest;
    return p < 1 ? 1 : p;
}
__device__ __forceinline__ int ec_depth(int v) { return (v & 7) ? (v & 7) * ((v >> 3) + 1) : v >> 3; }

// ec_fix1 (correct.c:121-220) on the strand `r`: the state of its best-first search
struct EcSearch {
    uint32_t hn, tn;           // paths in the queue, trace entries
    int n_done, no_hits;
    int64_t done_y0, done_y1;  // keys of the (up to two) best finished paths
};

__device__ __forceinline__ bool ec_seed(const EcRead &r, int w, EcSearch &S, const EcHeap &H, EcTrace &trace, EcCount &C)   // false: ec_fix1 returns 0xffff
{
    const int shift = (w - 1) << 1;
    if (r.len <= w) return false;
    uint64_t x = 0;
    int i, l;
    for (i = r.len - 1, l = 0; i > 0 && l < w; --i) {      // the k-mer at the end of the strand (windows restart after an N)
        const int c = r.base(i);
        if (c == 5) { x = 0; l = 0; }
        else { x = (uint64_t)(c - 1) << shift | x >> 2; ++l; }
    }
    if (i == 0) return false;
    S.hn = 0; S.tn = 0; S.n_done = 0; S.no_hits = 1; S.done_y0 = S.done_y1 = 0;
    trace.put(S.tn, 0, C);
    ++S.tn;
    ec_push(H, S.hn, x, (int64_t)(i + 1), C);
    return true;
}

// ec_fix's word for a read from the two strands' (correct.c:247-252)
__device__ __forceinline__ int ec_combine(int ret0, int ret)
{
    int out = ((ret0 & 0xffff) + (ret & 0xffff)) | ((ret0 >> 18 < ret >> 18 ? ret0 >> 18 : ret >> 18) << 18);
    if ((ret0 >> 17 & 1) && (ret >> 17 & 1)) out |= 1 << 16;
    return out;
}
// EL_JUMP: the path z `step` bases on (correct.c:183-185) -> the position of the base its next k-mer ends with
__device__ __forceinline__ int ec_hop(const EcRead &r, EcNode &z, int step, int shift)
{
    int i;
    int l;
    i = (int)((uint64_t)z.y & 0xffff) - 1;
    const int nav = i < step ? (i < 0 ? 0 : i) : step;             // bases the loop below would take if none of them is an N (i >= 1, l < step)
    bool batched = false;
#if EC_HOP_BATCH
    if (r.staged && nav > 0 && nav <= 8 && 2 * (nav - 1) <= shift) {
        // all of them at once: their nibbles are one field of the lane's LDS words (positions i - nav + 1 .. i of the strand; on the reverse strand the
        // bytes run the other way and the bases are complemented), A/C/G/T -> 2 bits each, into the k-mer in the order the loop shifts them in
        const int jl = r.rc ? r.len - 1 - i : i - nav + 1;
        const uint32_t wq = (uint32_t)jl >> 3, sh4 = 4u * ((uint32_t)jl & 7u);
        const uint32_t lo_w = r.lb[wq * 64], hi_w = wq + 1 < EC_LDS_BW ? r.lb[(wq + 1) * 64] : 0u;
        const uint32_t fm = nav == 8 ? ~0u : (1u << (4 * nav)) - 1u;
        const uint32_t F = (uint32_t)(((uint64_t)hi_w << 32 | lo_w) >> sh4) & fm;
        const uint32_t t = F - (0x11111111u & fm);                    // nibbles 0..3 for A/C/G/T, 4 for an N
        if (((t >> 2) & 0x11111111u & fm) == 0) {                     // no N among them
            uint32_t y = (t | t >> 2) & 0x0f0f0f0fu;                  // pairs of bases per byte
            y = (y | y >> 4) & 0x00ff00ffu; y = (y | y >> 8) & 0xffffu;   // nav bases, 2 bits each, the one at the lowest byte address first
            if (r.rc) y ^= (1u << (2 * nav)) - 1u;                    // complemented; the loop's first base is the one at the lowest address
            else { uint32_t rv = __brev(y) >> (32 - 2 * nav); y = ((rv >> 1) & 0x55555555u) | ((rv & 0x55555555u) << 1); }   // ... at the highest address: the 2-bit groups reversed
            z.x = (uint64_t)y << (shift - 2 * (nav - 1)) | z.x >> (2 * nav);
            i -= nav;
            batched = true;
        }
    }
#endif
    if (!batched)
    for (l = 0; i >= 1 && l < step; --i, ++l) {
        const int c = r.base(i);
        if (c >= 5) break;
        z.x = (uint64_t)(c - 1) << shift | z.x >> 2;
    }
    return i;
}
// ... and the verdict on it (correct.c:186-195): the hop stands if the table agrees, unambiguously and deep enough; then the path is kept as it is now
__device__ __forceinline__ bool ec_hop_good(int hit, int b, int i, int qv, EcNode &z, EcNode &keep, int &keep_i, int &keep_q, int &depth_last)
{
    bool good = hit >= 0 && b == (hit & 3) + 1;
    if (good) {
        const int v2 = hit >> 2, depth = ec_depth(v2);
        good = (v2 & 7) <= 1 && depth >= EC_MIN_OCC && (double)depth / depth_last >= EC_MIN_OCC_RATIO;
        if (good) {
            z.y = (int64_t)((uint64_t)z.y >> 16 << 16 | (uint64_t)(i + 1));
            keep = z; keep_i = i; keep_q = qv; depth_last = depth;
        }
    }
    return good;
}

// One lane = one read at a time.  The search of correct.c:141-206 and the walk back along the best path (:207-219) are cut into TURNS of the wave's loop, and
// a turn asks memory for ONE thing per lane -- a table slot (EL_POP: the best path's k-mer; EL_JUMP: the k-mer `step` bases on, correct.c:182-196) or a trace
// entry (EL_CLOSE) -- so that the 64 requests of a wave are in flight together whatever its lanes are doing: the reference's inner loops (up to 20 dependent
// look-ups while a clean read hops along, up to 100 dependent trace entries at the end) are turns here, not loops inside a turn that 63 lanes wait for.
// A lane whose read is finished draws the next one at once (reads with errors take 10-100x the expansions of clean ones: a wave never waits for its slowest read).
enum { EL_IDLE = 0, EL_POP, EL_JUMP, EL_CLOSE, EL_HOPEND, EL_SEED };
#ifndef EC_LB
#define EC_LB 4
#endif
__global__ __launch_bounds__(64, EC_LB) void k_ecfix(size_t n, uint8_t *__restrict__ seqs, uint8_t *__restrict__ quals, const uint64_t *__restrict__ off, int w, int step,
                                                 const uint64_t *__restrict__ slots, uint64_t mask, int32_t *__restrict__ info, uint4 *heaps, uint64_t *traces,
                                                 uint32_t trace_cap, uint32_t *__restrict__ queue)
{
    __shared__ uint4 lds_heap[EC_LDS_H * 64];
    __shared__ uint32_t lds_bases[EC_LDS_BW * 64];
#if EC_STAGE_QUAL
    __shared__ uint32_t lds_quals[EC_LDS_QW * 64];
#endif
#if EC_STAGE
    __shared__ uint64_t lds_trace[EC_TR_LINE * 64];
#endif
    const int lane = (int)threadIdx.x;
    const size_t slice = (size_t)blockIdx.x * 64 + threadIdx.x;
    EcHeap H; H.lds = lds_heap + lane; H.hbm = heaps + slice * EC_HEAP_SLOTS;
    EcTrace trace; trace.hbm = traces + slice * (size_t)trace_cap; trace.tag = 0;
#if EC_STAGE
    trace.ln = lds_trace + lane;
#else
    trace.ln = nullptr;
#endif
    const int shift = (w - 1) << 1;
    EcRead r; r.s = nullptr; r.q = nullptr; r.len = 0; r.rc = true; r.staged = false; r.lb = lds_bases + lane;
    EcQwin QW; QW.w0 = QW.w1 = QW.w2 = QW.w3 = 0; QW.tag = ~0ull;
#if EC_STAGE_QUAL
    r.lq = lds_quals + lane;
#else
    r.lq = nullptr;
#endif
    EcSearch S; S.hn = S.tn = 0; S.n_done = 0; S.no_hits = 1; S.done_y0 = S.done_y1 = 0;
    EcNode z, keep;                            // EL_JUMP: the path as the hop in progress leaves it, and as the last accepted hop left it
    z.x = 0; z.y = 0; keep = z;
    int keep_i = 0, keep_q = 0, depth_last = 0;
    uint32_t ct = 0;                           // EL_CLOSE: the trace entry to apply next
    int qsum = 0, score_diff = 0;
    size_t cur = 0;
    int ret0 = 0, st = EL_IDLE;
    bool drained = false;
    const bool full = queue[EC_FULL_FLAG] != 0;   // the one triple the table cannot hold (EC_FULL_FLAG)
    FmdTickets tk;
    EcCount C;
    fmd_tickets_init(tk, queue);
    for (;;) {
#if EC_GATE
        // ---- what a lane does ONCE per strand -- take a read and stage it, find the strand's seed k-mer; close a strand: walk its best path back, hand the
        // result on -- costs the wave its whole code every turn in which ANY lane does it, and with 64 lanes and ~18 turns per strand some lane nearly always
        // does.  So these wait for company: reads are taken in and strands seeded when EC_GATE_IN lanes want it, strands closed when EC_GATE_CL do -- or
        // when no lane of the wave is searching (then everything goes ahead: nothing can wait for ever).  Per lane the sequence of operations is unchanged.
        const uint64_t busy_m = __ballot(st == EL_POP || st == EL_JUMP || st == EL_HOPEND);
        const bool gate_in = busy_m == 0 || __popcll(__ballot((st == EL_IDLE && !drained) || st == EL_SEED)) >= EC_GATE_IN;
        const bool gate_cl = busy_m == 0 || __popcll(__ballot(st == EL_CLOSE)) >= EC_GATE_CL;
#else
        const bool gate_in = true, gate_cl = true;
#endif
        const size_t my = fmd_tickets_take(tk, queue, gate_in && st == EL_IDLE && !drained);
        if (gate_in) {
            if (st == EL_IDLE && !drained) {
                if (my < n) {
                    cur = my;
                    r.s = seqs + off[my]; r.q = quals + off[my]; r.len = (int)(off[my + 1] - off[my]); r.rc = true;
                    QW.tag = ~0ull;
                    ec_stage(r);
                    st = EL_SEED;
                } else drained = true;
            }
            if (st == EL_SEED) {                                           // the reverse-complement strand of a new read, or (r.rc false) the read as given after it
                if (ec_seed(r, w, S, H, trace, C)) st = EL_POP;
                else {
                    // first strand: too short, or no clean k-mer (correct.c:242-246); second: ec_fix1 returns 0xffff and ec_fix combines it all the same
                    info[cur] = r.rc ? 0xffff : ec_combine(ret0, 0xffff);
                    st = EL_IDLE;
                }
            }
        }
        if (__ballot(st != EL_IDLE) == 0) { if (__ballot(!drained) == 0) { C.flush(queue); break; } else continue; }

#if EC_HOP_LOOP
        // ---- hop chains first.  A strand in agreement with the table hops `step` bases at a time, one look-up per hop, up to (len - w) / step hops in a row
        // (correct.c:182-196) -- three turns in four of an error-free read -- and a turn of the general form below costs the wave everything any lane might
        // be doing: the queue's sift loops, the branches, the walk back.  So the lanes that are hopping hop on in a loop that holds nothing but the hop, its
        // look-up and the verdict, until fewer than EC_HOP_MIN of them still are (the rest wait; a chain that ends leaves its strand in EL_HOPEND, and the
        // turn below pushes the kept path and pops the next one as before).  Per lane the sequence of operations is the same as without the loop.
        if (__ballot(st == EL_JUMP)) {
            do {
                if (st == EL_JUMP) {
                    const int ih = ec_hop(r, z, step, shift);
                    const int bh = r.base(ih);
                    bool ends = bh == 5;
                    if (!ends) {
                        const int qh = ec_qual(r, ih, QW);
#if EC_HOP_WARM
                        // most hops stand, and then the next look-up is the k-mer `step` bases further on: its line is asked for NOW, beside this hop's
                        // (one word of it is loaded and never looked at: the line is in the caches by the time the next turn asks for it)
                        if (ih > 0) {
                            EcNode zs = z;
                            zs.y = (int64_t)((uint64_t)z.y >> 16 << 16 | (uint64_t)(ih + 1));       // (as ec_hop_good leaves a hop that stands)
                            (void)ec_hop(r, zs, step, shift);
                            const uint32_t w_ = *(const volatile uint32_t *)(slots + (ec_hash(zs.x) & (mask >> 3)) * EC_LINE);
                            (void)w_;
                        }
                        const int hh = ec_lookup(slots, mask, z.x, full, C);
#else
                        const int hh = ec_lookup(slots, mask, z.x, full, C);
#endif
                        ends = !ec_hop_good(hh, bh, ih, qh, z, keep, keep_i, keep_q, depth_last) || keep_i <= 0;
                    }
                    if (ends) st = EL_HOPEND;
                }
            } while (__popcll(__ballot(st == EL_JUMP)) >= EC_HOP_MIN);
        }
#endif
        // ---- what the lane wants from memory this turn
        bool want = false, pass_done = false, hop_end = false, overflow = false;
        int i = 0, b = 0;
        if (st == EL_HOPEND) {                                             // the path goes on from where the last good hop left it (correct.c:198); then the queue's best
            st = EL_POP;
            if (!ec_branch(H, S.hn, trace, S.tn, trace_cap, keep, r.base(keep_i) - 1, 0, shift, 1, keep_q, C)) { info[cur] = (int32_t)EC_INFO_TRACE_FULL; st = EL_IDLE; }
        }
        if (st == EL_POP) {
            if (S.hn == 0) pass_done = true;
            else {
                z = ec_pop(H, S.hn, C);
                const uint64_t zy = (uint64_t)z.y;
                if ((zy & 0xffff) == 0) {                                  // a path that reached the start of the strand
                    if (S.n_done == 0) S.done_y0 = z.y; else S.done_y1 = z.y;
                    ++S.n_done;
                    pass_done = S.n_done == 2;
                } else if (S.n_done && (int)(zy >> 48) > (int)((uint64_t)S.done_y0 >> 48) + EC_MAX_SC_DIFF) pass_done = true;
                else { i = (int)(zy & 0xffff) - 1; b = r.base(i); want = true; }
            }
        }
#if !(EC_HOP_LOOP && EC_HOP_ONLY)
        else if (st == EL_JUMP) {                                          // `step` bases on (correct.c:183-185)
            i = ec_hop(r, z, step, shift);
            b = r.base(i);
            if (b == 5) hop_end = true; else want = true;
        }
#endif
        if (pass_done) {                                                   // correct.c:207-212
            score_diff = S.n_done == 1 ? EC_MAX_SC_DIFF : (int)((uint64_t)S.done_y1 >> 48) - (int)((uint64_t)S.done_y0 >> 48);
            if (score_diff >= EC_MAX_SC_DIFF) score_diff = EC_MAX_SC_DIFF;
            qsum = 0;
            ct = ((uint64_t)S.done_y0 >> 48) == 0 ? 0u : (uint32_t)((uint64_t)S.done_y0 >> 16);   // nothing to change: no walk back
            st = EL_CLOSE;
#if EC_STAGE
            if (ct && S.tn) { trace.flush(S.tn - 1, C); trace.tag = (S.tn - 1) / EC_TR_LINE; }   // the open line leaves too (whole: what lies behind its last entry is never read); the LDS line still holds it
#endif
        }
        // ---- the turn's one request per lane
        int hit = -1, qv = 0;
        uint64_t te = 0;
        if (want) { qv = ec_qual(r, i, QW); hit = ec_lookup(slots, mask, z.x, full, C); }   // (the byte's load is in flight beside the line's)
#if EC_STAGE
        uint4 tl[EC_TR_LINE / 2];
        const bool tload = st == EL_CLOSE && gate_cl && ct && ct / EC_TR_LINE != trace.tag;
        if (tload) {
            const uint4 *src = (const uint4 *)(trace.hbm + (ct & ~(uint32_t)(EC_TR_LINE - 1)));
#pragma unroll
            for (int e = 0; e < EC_TR_LINE / 2; ++e) tl[e] = src[e];
            C.add(2, EC_TR_LINE);
        }
#else
        if (st == EL_CLOSE && gate_cl && ct) { te = trace.hbm[ct]; C.add(2); }
#endif
        // ---- what came back
#if EC_PUSH_AB
        // The paths a turn adds to the queue -- none, one (a miss; the kept path where a hop chain cannot start) or two (the read's base and the table's) -- are
        // described first and pushed by TWO shared calls: four inlined copies of the push (trace entry + sift loop) cost the wave four of them per turn
        // whenever its lanes disagree about which one they need.
        bool pa = false, pb = false, a_keep = false;
        int ac = 0, acost = 0, amatch = 1, bc = 0, bcost = 0;
        if (want && st == EL_POP) {
            int q = qv - 33;
            q = q < EC_MAX_QUAL ? q : EC_MAX_QUAL;
            q = q < 3 ? 3 : q;
            if (hit < 0) { pa = true; ac = b - 1; acost = EC_MISS_PENALTY + (EC_MAX_QUAL - q); amatch = 0; }
            else {
                const int best = (hit & 3) + 1, v = hit >> 2;
                S.no_hits = 0;
                if (b != best) {                                     // the table prefers another base: follow both, within the queue's budget
                    const int pen = ec_swap_penalty(v);
                    if (b != 5 && (S.hn + 2 <= EC_MAX_HEAP || pen < q)) { pa = true; ac = b - 1; acost = pen; }
                    if (b == 5 || S.hn + (pa ? 1u : 0u) + 2 <= EC_MAX_HEAP || pen > q) { pb = true; bc = best - 1; bcost = q; }   // (the budget as it stands after the first push)
                } else {                                             // agreement: hop `step` bases at a time while the k-mers stay deep and unambiguous
                    keep = z; keep_i = i; keep_q = qv; depth_last = ec_depth(v);
                    if ((v & 7) <= 0 && step > 1 && keep_i > 0) st = EL_JUMP;
                    else hop_end = true;
                }
            }
        }
#if !(EC_HOP_LOOP && EC_HOP_ONLY)
        else if (want) {                                                   // EL_JUMP: is the hop good?  (correct.c:186-195)
            const bool good = ec_hop_good(hit, b, i, qv, z, keep, keep_i, keep_q, depth_last);
            if (!good || keep_i <= 0) hop_end = true;
        }
#endif
        if (hop_end) { pa = true; a_keep = true; ac = r.base(keep_i) - 1; acost = 0; st = EL_POP; }   // the path goes on from where the last good hop left it (correct.c:198)
        if (pa) overflow = !ec_branch(H, S.hn, trace, S.tn, trace_cap, a_keep ? keep : z, ac, acost, shift, amatch, a_keep ? keep_q : qv, C);
        if (pb && !overflow) overflow = !ec_branch(H, S.hn, trace, S.tn, trace_cap, z, bc, bcost, shift, 1, qv, C);
#else
        if (want && st == EL_POP) {
            int q = qv - 33;
            q = q < EC_MAX_QUAL ? q : EC_MAX_QUAL;
            q = q < 3 ? 3 : q;
            bool ok = true;
            if (hit < 0) ok = ec_branch(H, S.hn, trace, S.tn, trace_cap, z, b - 1, EC_MISS_PENALTY + (EC_MAX_QUAL - q), shift, 0, qv, C);
            else {
                const int best = (hit & 3) + 1, v = hit >> 2;
                S.no_hits = 0;
                if (b != best) {                                     // the table prefers another base: follow both, within the queue's budget
                    const int pen = ec_swap_penalty(v);
                    if (b != 5 && (S.hn + 2 <= EC_MAX_HEAP || pen < q)) ok = ec_branch(H, S.hn, trace, S.tn, trace_cap, z, b - 1, pen, shift, 1, qv, C);
                    if (ok && (b == 5 || S.hn + 2 <= EC_MAX_HEAP || pen > q)) ok = ec_branch(H, S.hn, trace, S.tn, trace_cap, z, best - 1, q, shift, 1, qv, C);
                } else {                                             // agreement: hop `step` bases at a time while the k-mers stay deep and unambiguous
                    keep = z; keep_i = i; keep_q = qv; depth_last = ec_depth(v);
                    if ((v & 7) <= 0 && step > 1 && keep_i > 0) st = EL_JUMP;
                    else hop_end = true;
                }
            }
            overflow = !ok;
        }
#if !(EC_HOP_LOOP && EC_HOP_ONLY)
        else if (want) {                                                   // EL_JUMP: is the hop good?  (correct.c:186-195)
            const bool good = ec_hop_good(hit, b, i, qv, z, keep, keep_i, keep_q, depth_last);
            if (!good || keep_i <= 0) hop_end = true;
        }
#endif
        if (hop_end) {                                                     // the path goes on from where the last good hop left it (correct.c:198)
            overflow = !ec_branch(H, S.hn, trace, S.tn, trace_cap, keep, r.base(keep_i) - 1, 0, shift, 1, keep_q, C);
            st = EL_POP;
        }
#endif
        if (overflow) { info[cur] = (int32_t)EC_INFO_TRACE_FULL; st = EL_IDLE; continue; }
        if (st != EL_CLOSE || !gate_cl) continue;
#if EC_STAGE
        if (tload) {                                                       // the line of the chain's next entry into the lane's LDS line
#pragma unroll
            for (int e = 0; e < EC_TR_LINE / 2; ++e) { trace.ln[2 * e * 64] = (uint64_t)tl[e].y << 32 | tl[e].x; trace.ln[(2 * e + 1) * 64] = (uint64_t)tl[e].w << 32 | tl[e].z; }
            trace.tag = ct / EC_TR_LINE;
        }
        while (ct && ct / EC_TR_LINE == trace.tag) {                       // every choice of the best path that this line holds (correct.c:213-218)
            te = trace.ln[(ct & (EC_TR_LINE - 1)) * 64];
#else
        if (ct) {                                                          // one choice of the best path applied (correct.c:213-218)
#endif
            const int pos = (int)((te >> 32) & 0xffffu), qq = (int)((te >> 48) & 0xffu);
            const uint32_t lo = (uint32_t)te, c = lo >> 29;
            if ((uint32_t)(r.base(pos) - 1) != c) { qsum += qq - 33; r.set_base(pos, (int)c + 1); }
            else if ((lo >> 28 & 1) && qq < 37) { r.set_qual(pos, 37); QW.tag = ~0ull; }
            ct = lo << 4 >> 4;
#if !EC_STAGE
            if (ct) continue;
#endif
        }
#if EC_STAGE
        if (ct) continue;
#endif
        {                                                                  // the strand is done
            int ret = ((uint64_t)S.done_y0 >> 48) == 0 ? score_diff << 18 : (qsum | score_diff << 18 | S.no_hits << 17);
            if (r.rc) { ret0 = ret; r.rc = false; st = EL_SEED; continue; }   // the reverse-complement strand is done: now the read as given (seeded at the top, in company)
            info[cur] = ec_combine(ret0, ret);
            st = EL_IDLE;
        }
    }
}

// ------------------------------------------------------------------------------------------------ host entries
extern "C" void fmd_ectab_free(fmd_ectab_t *t)
{
    if (!t) return;
    hipSetDevice(t->device);
    hipFree(t->slots); hipFree(t->queue);
    for (int i = 0; i < 5; ++i) if (t->buf[i]) hipFree(t->buf[i]);
    free(t);
}

// What the kernels launched on this table requested since the last reset -- {table slots probed, queue entries moved, trace entries moved} --
// when the library is the instrumented build (*counting = 1; zeros otherwise).  Synchronises the device.
extern "C" int fmd_ectab_line_count(fmd_ectab_t *t, uint64_t counts[3], int reset, int *counting)
{
    if (!t || !counts) return FMD_E_ARG;
    FMD_HIP_TRY(hipSetDevice(t->device));
    FMD_HIP_TRY(hipDeviceSynchronize());
    uint32_t host[16];
    FMD_HIP_TRY(hipMemcpy(host, t->queue, sizeof(host), hipMemcpyDeviceToHost));
    memcpy(counts, host + EC_STAT_U32, 24);
    if (reset) FMD_HIP_TRY(hipMemset(t->queue + EC_STAT_U32, 0, 24));
    if (counting) *counting = FMD_COUNT_LINES;
    return FMD_OK;
}

extern "C" int fmd_ectab_build_dev(int device, void *stream_, int w, int suf_len, uint64_t n, const uint32_t *d_bucket, const uint32_t *d_key,
                                   const uint8_t *d_val, fmd_ectab_t **out)
{
    if (!out || w < 2 || w > 27 || suf_len < 1 || 2 * w - 2 * suf_len > 30 || (n && (!d_bucket || !d_key || !d_val))) return FMD_E_ARG;
    if (fmd_device_count() <= 0) return FMD_E_NODEV;
    FMD_HIP_TRY(hipSetDevice(device));
    fmd_ectab *t = (fmd_ectab *)calloc(1, sizeof(fmd_ectab));
    if (!t) return FMD_E_NOMEM;
    t->device = device; t->w = w; t->suf_len = suf_len;
    t->n_slots = 1024;
    while (t->n_slots < 2 * n) t->n_slots <<= 1;              // load factor <= 1/2
    hipStream_t st = (hipStream_t)stream_;
    if (hipMalloc((void **)&t->slots, t->n_slots * 8) != hipSuccess || hipMalloc((void **)&t->queue, 64) != hipSuccess) {
        fmd_set_hip_error(hipGetLastError(), "hipMalloc(k-mer table)");
        fmd_ectab_free(t);
        return FMD_E_NOMEM;
    }
    FMD_HIP_TRY(hipMemsetAsync(t->slots, 0xff, t->n_slots * 8, st));
    FMD_HIP_TRY(hipMemsetAsync(t->queue, 0, 64, st));
    if (n) {
        size_t blocks = (size_t)((n + 255) / 256);
        if (blocks > (1u << 20)) blocks = 1u << 20;
        k_ectab_fill<<<(unsigned)blocks, 256, 0, st>>>(n, suf_len, d_bucket, d_key, d_val, t->slots, t->n_slots - 1, t->queue + EC_FULL_FLAG);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { fmd_set_hip_error(e, "k_ectab_fill"); fmd_ectab_free(t); return FMD_E_HIP; }
    *out = t;
    return FMD_OK;
}

extern "C" int fmd_ectab_build(int device, int w, int suf_len, uint64_t n, const uint32_t *bucket, const uint32_t *key, const uint8_t *val, fmd_ectab_t **out)
{
    if (n && (!bucket || !key || !val)) return FMD_E_ARG;
    if (fmd_device_count() <= 0) return FMD_E_NODEV;
    FMD_HIP_TRY(hipSetDevice(device));
    void *db = nullptr, *dk = nullptr, *dv = nullptr;
    int rc = FMD_OK;
    if (hipMalloc(&db, n * 4 + 16) != hipSuccess || hipMalloc(&dk, n * 4 + 16) != hipSuccess || hipMalloc(&dv, n + 16) != hipSuccess) rc = FMD_E_NOMEM;
    if (rc == FMD_OK && n && (hipMemcpy(db, bucket, n * 4, hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(dk, key, n * 4, hipMemcpyHostToDevice) != hipSuccess ||
                              hipMemcpy(dv, val, n, hipMemcpyHostToDevice) != hipSuccess)) rc = FMD_E_HIP;
    if (rc == FMD_OK) rc = fmd_ectab_build_dev(device, nullptr, w, suf_len, n, (uint32_t *)db, (uint32_t *)dk, (uint8_t *)dv, out);
    if (rc == FMD_OK && hipDeviceSynchronize() != hipSuccess) { rc = FMD_E_HIP; fmd_ectab_free(*out); *out = nullptr; }
    hipFree(db); hipFree(dk); hipFree(dv);
    return rc;
}

static int ec_grid(int device, size_t n)   // the resident set: a workgroup (one wave) holds EC_LDS_BYTES of the CU's 160 KiB
{
    hipDeviceProp_t prop;
    int cus = 256;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) cus = prop.multiProcessorCount;
    const size_t per_cu = (160 * 1024) / (((size_t)EC_LDS_BYTES + 1279) / 1280 * 1280);
    size_t waves = (size_t)cus * per_cu, need = (n + 63) / 64;
    return (int)(need < waves ? (need ? need : 1) : waves);
}

extern "C" size_t fmd_ecfix_work_bytes(const fmd_ectab_t *t, size_t n, uint32_t trace_cap)
{
    if (!t) return 0;
    const size_t lanes = (size_t)ec_grid(t->device, n) * 64;
    return lanes * (EC_HEAP_SLOTS * sizeof(uint4) + (size_t)trace_cap * 8) + 256;
}

extern "C" int fmd_ecfix_dev(fmd_ectab_t *t, void *stream_, size_t n, uint8_t *d_seqs, uint8_t *d_quals, const uint64_t *d_off, int step, uint32_t trace_cap,
                             int32_t *d_info, void *d_work, size_t work_bytes)
{
    if (!t || (n && (!d_seqs || !d_quals || !d_off || !d_info || !d_work)) || trace_cap < 16) return FMD_E_ARG;
    if (n == 0) return FMD_OK;
    if (n >= 0xffffff00ull || work_bytes < fmd_ecfix_work_bytes(t, n, trace_cap)) return FMD_E_ARG;
    FMD_HIP_TRY(hipSetDevice(t->device));
    hipStream_t st = (hipStream_t)stream_;
    const int grid = ec_grid(t->device, n);
    if (EC_TR_LINE > 0) trace_cap &= ~(uint32_t)((EC_TR_LINE > 0 ? EC_TR_LINE : 1) - 1);               // slices of whole 64-byte lines (the work area was sized by the caller's number: no smaller)
    uint4 *heaps = (uint4 *)(((uintptr_t)d_work + 255) & ~(uintptr_t)255);
    uint64_t *traces = (uint64_t *)(heaps + (size_t)grid * 64 * EC_HEAP_SLOTS);
    FMD_HIP_TRY(hipMemsetAsync(t->queue, 0, 4, st));
    k_ecfix<<<grid, 64, 0, st>>>(n, d_seqs, d_quals, d_off, t->w, step, t->slots, t->n_slots - 1, d_info, heaps, traces, trace_cap, t->queue);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { fmd_set_hip_error(e, "k_ecfix"); return FMD_E_HIP; }
    return FMD_OK;
}

// Device buffers of the host form, kept in the table handle between calls (`correct` calls it once per 10^6 reads: the work
// area alone is 256 CUs x 16 waves x 64 lanes x ~12 KB).  One caller at a time owns them; a concurrent call allocates its own.
struct EcBuf {
    fmd_ectab *t; bool cached; void *p[5]; size_t bytes[5];
    explicit EcBuf(fmd_ectab *t_) : t(t_), cached(false)
    {
        for (int i = 0; i < 5; ++i) { p[i] = nullptr; bytes[i] = 0; }
        int expect = 0;
        if (__atomic_compare_exchange_n(&t->buf_busy, &expect, 1, false, __ATOMIC_ACQUIRE, __ATOMIC_RELAXED)) {
            cached = true;
            for (int i = 0; i < 5; ++i) { p[i] = t->buf[i]; bytes[i] = t->buf_bytes[i]; }
        }
    }
    void *get(int i, size_t need)   // at least `need` bytes in buffer i (contents are not kept), nullptr when the device is out of memory
    {
        if (bytes[i] >= need && p[i]) return p[i];
        if (p[i]) hipFree(p[i]);
        p[i] = nullptr; bytes[i] = 0;
        const size_t want = need + need / 8 + 256;
        if (hipMalloc(&p[i], want) != hipSuccess) { (void)hipGetLastError(); p[i] = nullptr; return nullptr; }
        bytes[i] = want;
        return p[i];
    }
    ~EcBuf()
    {
        if (cached) {
            for (int i = 0; i < 5; ++i) { t->buf[i] = p[i]; t->buf_bytes[i] = bytes[i]; }
            __atomic_store_n(&t->buf_busy, 0, __ATOMIC_RELEASE);
        } else for (int i = 0; i < 5; ++i) if (p[i]) hipFree(p[i]);
    }
};

// Host form: reads whose trace overflows are run again, from their original bytes, with the trace four times as long.
extern "C" int fmd_ecfix_batch(fmd_ectab_t *t, size_t n, uint8_t *seqs, uint8_t *quals, const uint64_t *off, int step, int32_t *info)
{
    if (!t || (n && (!seqs || !quals || !off || !info))) return FMD_E_ARG;
    if (n == 0) return FMD_OK;
    FMD_HIP_TRY(hipSetDevice(t->device));
    const uint64_t total = off[n] - off[0];
    uint32_t cap = 1024;
    EcBuf B(t);
    void *ds = B.get(0, total + 16), *dq = B.get(1, total + 16), *doff = B.get(2, (n + 1) * 8), *dinfo = B.get(3, n * 4);
    if (!ds || !dq || !doff || !dinfo) return FMD_E_NOMEM;
    std::vector<uint64_t> rel(n + 1);   // offsets relative to the first read
    for (size_t i = 0; i <= n; ++i) rel[i] = off[i] - off[0];
    uint8_t *s0 = seqs + off[0], *q0 = quals + off[0];
    FMD_HIP_TRY(hipMemcpy(ds, s0, total, hipMemcpyHostToDevice));
    FMD_HIP_TRY(hipMemcpy(dq, q0, total, hipMemcpyHostToDevice));
    FMD_HIP_TRY(hipMemcpy(doff, rel.data(), (n + 1) * 8, hipMemcpyHostToDevice));
    {
        const size_t wb = fmd_ecfix_work_bytes(t, n, cap);
        void *dwork = B.get(4, wb);
        if (!dwork) return FMD_E_NOMEM;
        const int rc = fmd_ecfix_dev(t, nullptr, n, (uint8_t *)ds, (uint8_t *)dq, (uint64_t *)doff, step, cap, (int32_t *)dinfo, dwork, wb);
        if (rc != FMD_OK) return rc;
    }
    FMD_HIP_TRY(hipMemcpy(info, dinfo, n * 4, hipMemcpyDeviceToHost));
    // the reads to run again (rare: a trace of 1024 entries is enough for all but the most error-ridden reads): their ORIGINAL bytes
    // are taken from the caller's arrays now, before the corrected batch is copied over them
    std::vector<size_t> again;
    for (size_t i = 0; i < n; ++i) if ((uint32_t)info[i] == EC_INFO_TRACE_FULL) again.push_back(i);
    std::vector<uint64_t> ko(again.size() + 1, 0);
    for (size_t k = 0; k < again.size(); ++k) ko[k + 1] = ko[k] + (rel[again[k] + 1] - rel[again[k]]);
    std::vector<uint8_t> keep_s(ko.back() + 16), keep_q(ko.back() + 16);
    for (size_t k = 0; k < again.size(); ++k) {
        memcpy(keep_s.data() + ko[k], s0 + rel[again[k]], ko[k + 1] - ko[k]);
        memcpy(keep_q.data() + ko[k], q0 + rel[again[k]], ko[k + 1] - ko[k]);
    }
    FMD_HIP_TRY(hipMemcpy(s0, ds, total, hipMemcpyDeviceToHost));
    FMD_HIP_TRY(hipMemcpy(q0, dq, total, hipMemcpyDeviceToHost));
    // re-runs: `todo` indexes `again` (and so the kept originals)
    std::vector<size_t> todo(again.size());
    for (size_t k = 0; k < todo.size(); ++k) todo[k] = k;
    for (int attempt = 0; attempt < 12 && !todo.empty(); ++attempt) {
        cap *= 4;
        const size_t m = todo.size();
        std::vector<uint64_t> o2(m + 1, 0);
        for (size_t k = 0; k < m; ++k) o2[k + 1] = o2[k] + (ko[todo[k] + 1] - ko[todo[k]]);
        std::vector<uint8_t> s2(o2[m] + 16), q2(o2[m] + 16);
        for (size_t k = 0; k < m; ++k) {
            memcpy(s2.data() + o2[k], keep_s.data() + ko[todo[k]], o2[k + 1] - o2[k]);
            memcpy(q2.data() + o2[k], keep_q.data() + ko[todo[k]], o2[k + 1] - o2[k]);
        }
        std::vector<int32_t> i2(m);
        const size_t wb = fmd_ecfix_work_bytes(t, m, cap);
        void *dwork = B.get(4, wb);
        if (!dwork) return FMD_E_NOMEM;
        FMD_HIP_TRY(hipMemcpy(ds, s2.data(), o2[m], hipMemcpyHostToDevice));
        FMD_HIP_TRY(hipMemcpy(dq, q2.data(), o2[m], hipMemcpyHostToDevice));
        FMD_HIP_TRY(hipMemcpy(doff, o2.data(), (m + 1) * 8, hipMemcpyHostToDevice));
        const int rc = fmd_ecfix_dev(t, nullptr, m, (uint8_t *)ds, (uint8_t *)dq, (uint64_t *)doff, step, cap, (int32_t *)dinfo, dwork, wb);
        if (rc != FMD_OK) return rc;
        FMD_HIP_TRY(hipMemcpy(s2.data(), ds, o2[m], hipMemcpyDeviceToHost));
        FMD_HIP_TRY(hipMemcpy(q2.data(), dq, o2[m], hipMemcpyDeviceToHost));
        FMD_HIP_TRY(hipMemcpy(i2.data(), dinfo, m * 4, hipMemcpyDeviceToHost));
        std::vector<size_t> left;
        for (size_t k = 0; k < m; ++k) {
            const size_t i = again[todo[k]];
            info[i] = i2[k];
            if ((uint32_t)i2[k] == EC_INFO_TRACE_FULL) { left.push_back(todo[k]); continue; }
            memcpy(s0 + rel[i], s2.data() + o2[k], o2[k + 1] - o2[k]);
            memcpy(q0 + rel[i], q2.data() + o2[k], o2[k + 1] - o2[k]);
        }
        todo.swap(left);
    }
    return todo.empty() ? FMD_OK : FMD_E_OVERFLOW;
}
