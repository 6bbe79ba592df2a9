// fmd_smem.hip -- super-maximal exact matches: fm6_smem1_core (smem.c:13-80) driven as fm6_smem
// does (smem.c:397-410), i.e. what `fermi exact` prints (cmd.c:319-327).  One lane per read on the
// compact wave engine (12 waves/CU); the forward sweep is a chain of forward extensions, the
// backward sweep walks the candidate list once per base.  Candidate lists live in an HBM work
// area owned by the persistent LANE (two lists of 2*max_len+2 entries, reused read after read, so the
// area does not grow with the batch); SMEMs are written to the caller's array in the reference's order.
//
// The lists and the memory system.  One list entry is 32 bytes at an address no other lane is near, so
// every push and every pick used to be a request of its own to the fabric -- a third of all requests
// of this kernel, which runs at the fabric's request rate.  Now: (1) the last two entries pushed sit in
// LDS (two 32-byte slots per lane) and go to HBM as ONE aligned 64-byte burst when a pair is complete;
// the first entry of the backward sweep (the last one pushed) is taken from its slot; (2) picks fetch the
// next entry NEEDED under the current gather; (3) the last round of the backward
// sweep (i = -1) writes no list at all -- nobody reads `curr` after it (smem.c:52, :75) -- and skips
// every entry that can no longer matter: once curr is non-empty, an entry that is not followed by a
// sentinel (x[1] >= n_seq) can neither be a full-length match nor be kept (smem.c:61-62), so its
// fm6_extend would be dead work.  Which entries are sentinel-closed is known when they are pushed: a
// bit per entry in a 64-bit register (entries beyond the 64th are simply processed).
#include <stdlib.h>
#include <string.h>
#include "fmd_internal.h"
#include "fmd_kernel_common.h"

// forward-sweep push (the list is written back to front); a full list marks the item as overflowed
#define SM_PUSH_FWD(a_, b_, c_, d_) do { if (curr_n < cap) { sm_push(la, slots, pend_e, list_units, cap - 1 - curr_n, a_, b_, c_, d_); fmask = fmask << 1 | (uint64_t)((b_) < ix.n_seq); ++curr_n; } else overflow = true; } while (0)

#define SM_NONE 0xffffffffu
// the two LDS slots of a lane (slot = entry index & 1), planes of 64 lanes x 16 bytes: no bank conflicts
#define SM_SLOT(s_, p_) slots[(((s_) * 2 + (p_)) << 6) + fmd_lane()]

// (units: list entries moved to or from HBM, 32 bytes each -- read by the instrumented build only, dead code otherwise)
__device__ __forceinline__ void sm_flush_single(fmd_intv_t *la, uint4 *slots, uint32_t &pend_e, uint32_t &units)
{
    ++units;
    uint4 *dst = (uint4 *)(la + pend_e);
    const uint32_t s = pend_e & 1;
    dst[0] = SM_SLOT(s, 0); dst[1] = SM_SLOT(s, 1);
    pend_e = SM_NONE;
}

// entry e of the lane's area (la[e]; lb = la + cap) := (a, b, c, d).  It is written to its LDS slot; HBM gets it together with
// its pair partner (e ^ 1) as one aligned 64-byte burst when that one was the push before, else when its slot is needed again.
__device__ __forceinline__ void sm_push(fmd_intv_t *la, uint4 *slots, uint32_t &pend_e, uint32_t &units, uint32_t e, uint64_t a, uint64_t b, uint64_t c, uint64_t d)
{
    if (pend_e != SM_NONE && pend_e != (e ^ 1u)) sm_flush_single(la, slots, pend_e, units);
    const uint32_t s = e & 1;
    const uint4 v0 = make_uint4((uint32_t)a, (uint32_t)(a >> 32), (uint32_t)b, (uint32_t)(b >> 32));
    const uint4 v1 = make_uint4((uint32_t)c, (uint32_t)(c >> 32), (uint32_t)d, (uint32_t)(d >> 32));
    SM_SLOT(s, 0) = v0; SM_SLOT(s, 1) = v1;
    if (pend_e == (e ^ 1u)) {
        const uint4 p0 = SM_SLOT(s ^ 1, 0), p1 = SM_SLOT(s ^ 1, 1);
        uint4 *dm = (uint4 *)(la + e), *dp = (uint4 *)(la + (e ^ 1u));   // (addresses, not data, depend on which half this is)
        dm[0] = v0; dm[1] = v1; dp[0] = p0; dp[1] = p1;
        pend_e = SM_NONE; units += 2;
    } else pend_e = e;
}

// first entry at or after j that the last round still needs (bit = the entry is sentinel-closed; entries from the 64th on: all)
__device__ __forceinline__ uint32_t sm_next_needed(uint32_t j, uint32_t n, uint64_t mask)
{
    if (j >= 64) return j;
    const uint64_t rem = mask >> j;
    if (rem) return j + (uint32_t)__builtin_ctzll(rem);
    return n < 64 ? n : 64;
}

enum { SM_IDLE = 0, SM_START, SM_BEGIN_BWD, SM_BWD_PICK, SM_FWD, SM_FWD_END, SM_BWD };

#define SMEM_LDS_BYTES (FMD_COMPACT_LDS_U4 * 16 + 4 * 64 * 16)   // the engine + the two list slots per lane
#define SMEM_MAX_WAVES 4096   // the candidate lists belong to the persistent lane, not to the read

__global__ __launch_bounds__(64, 3) void k_smem(FmdIndexView ix, size_t n, const uint8_t *__restrict__ seqs, const uint64_t *__restrict__ off,
                                             int self_match, uint32_t cap, fmd_intv_t *__restrict__ work, uint32_t max_mem,
                                             fmd_intv_t *__restrict__ mem_out, uint32_t *__restrict__ n_mem_out, uint32_t *__restrict__ queue,
                                             int refill_min, const fmd_smem_win_t *__restrict__ wins, const uint32_t *__restrict__ mixed, int patience)
{
    // reads of one length (k_smem_mixed, same stream): the wave waits for ALL its lanes before it takes new reads -- while that works
    bool patient = mixed && *mixed == 0;
    int drain = 0;                               // wave steps since the first lane of the wave went idle
    FMD_DECLARE_COMPACT_LDS();
    __shared__ uint4 slots[4 * 64];
    size_t rid = 0;
    const uint8_t *q = nullptr;
    int st = SM_IDLE, len = 0, x = 0, i = 0, ret = 0, stop = 0;
    uint32_t prev_n = 0, curr_n = 0, j = 0, n_mem = 0, call_base = 0;   // SMEMs found (the algorithm's count)
    uint32_t n_out = 0, out_base = 0;                                    // SMEMs written (full_only drops some)
    bool full_only = false;
    // two candidate lists of `cap` entries per lane (HBM; this lane's area is reused read after read): la = entries [0, cap) of
    // the area, lb = [cap, 2 cap).  prev / curr are entry indices into it.
    fmd_intv_t *const la = work + ((size_t)blockIdx.x * 64 + fmd_lane()) * 2 * (size_t)cap;
    uint32_t prev_e0 = 0, curr_e0 = 0, pend_e = SM_NONE, pf_e = SM_NONE, list_units = 0;
    uint64_t fmask = 0, kmask = 0, cmask = 0;        // sentinel-closed entries of the list being pushed / of prev / of curr
    uint64_t kx0 = 0, kx1 = 0, ksz = 0, kinfo = 0;   // ik (forward sweep) / p (backward sweep)
    uint64_t last_curr_sz = 0, last_mem_beg = 0;
    uint64_t sbase = 0;                              // off[rid]
    uint32_t cw = 0; uint64_t cw_at = ~0ull;         // four bases of the read around the position in use
    uint4 pfa = make_uint4(0, 0, 0, 0), pfb = pfa;   // entry pf_e of the area, fetched under a gather
    bool exhausted = false, overflow = false;

    FmdTickets tk_;
    fmd_tickets_init(tk_, queue, 64, n);   // guided chunks of up to 64 reads: a launch of 5*10^7 reads in chunks of 16 is 3*10^6 atomics on one counter
    for (;;) {
        // ---- refill
        // Reads are taken in groups of at least refill_min lanes: a new read starts with ~log4(n) steps on
        // wide intervals (two six-symbol block ranks), and a wave pays for that code path whenever ONE
        // lane is in it; starting reads together keeps most steps free of it.
        const uint64_t idle = __ballot(st == SM_IDLE);
        drain = idle ? drain + 1 : 0;
        if (drain > patience) patient = false;   // the lanes' work differs too much (several calls per read, long backward sweeps): groups from now on
        const bool take = !exhausted && ((!patient && __popcll(idle) >= refill_min) || idle == ~0ull);
        const size_t my = fmd_tickets_take(tk_, queue, st == SM_IDLE && take, n);
        if (st == SM_IDLE && take) {
            if (my < n) {
                rid = my; cw_at = ~0ull; n_mem = 0; n_out = 0; overflow = false; full_only = false;
                if (wins) { // a window of a long sequence: start positions [start, stop) of the fm6_smem chain
                    const fmd_smem_win_t wn = wins[my];
                    sbase = wn.seq_off; len = (int)wn.seq_len; x = (int)wn.start; stop = (int)(wn.stop < wn.seq_len ? wn.stop : wn.seq_len);
                    full_only = (wn.reserved & FMD_SMEM_WIN_F_FULL) != 0;
                } else { sbase = off[my]; len = (int)(off[my + 1] - sbase); x = 0; stop = len; }
                q = seqs + sbase;
                if (len <= 0 || x >= stop) n_mem_out[rid] = 0;
                else if (!wins && 2 * (uint32_t)len + 2 > cap) n_mem_out[rid] = 0x80000000u; // longer than max_len
                else st = SM_START;
            } else exhausted = true;
        }
        // ---- transitions that need no rank
        bool again = st == SM_START || st == SM_BEGIN_BWD || st == SM_BWD_PICK;
        while (again) {
            again = false;
            if (st == SM_START) { // fm6_smem1_core prologue (smem.c:19-21)
                const int c = q[x];
                kx0 = ix.cnt[c]; kx1 = ix.cnt[comp6(c)]; ksz = ix.cnt[c + 1] - ix.cnt[c]; kinfo = (uint64_t)(x + 1);
                curr_n = 0; call_base = n_mem; out_base = n_out; i = x + 1; fmask = 0; pend_e = SM_NONE;
                if (ksz == 0) { // the reference dereferences an empty list here (undefined); stop this read
                    n_mem_out[rid] = n_out | (overflow ? 0x80000000u : 0);
                    st = SM_IDLE;
                } else if (i < len) st = SM_FWD;
                else { // x is the last base: push the interval (smem.c:35-37); list is written back to front
                    SM_PUSH_FWD(kx0, kx1, ksz, kinfo);
                    st = self_match ? SM_BEGIN_BWD : SM_FWD_END;
                    again = st == SM_BEGIN_BWD;
                }
            } else if (st == SM_BEGIN_BWD) { // the forward list, already reversed, becomes prev (smem.c:45-50)
                if (curr_n == 0) { // nothing was pushed (every occurrence of q[x] ends its sequence): the reference reads a[0] of an empty list; stop this read
                    n_mem_out[rid] = n_out | (overflow ? 0x80000000u : 0);
                    st = SM_IDLE;
                } else {
                    prev_e0 = cap - curr_n; prev_n = curr_n; kmask = fmask;
                    { // prev[0] = the last entry pushed: still in its slot
                        const uint4 a = SM_SLOT(prev_e0 & 1, 0), b = SM_SLOT(prev_e0 & 1, 1);
                        kx0 = (uint64_t)a.y << 32 | a.x; kx1 = (uint64_t)a.w << 32 | a.z;
                        ksz = (uint64_t)b.y << 32 | b.x; kinfo = (uint64_t)b.w << 32 | b.z;
                    }
                    ret = (int)kinfo;
                    pend_e = SM_NONE;   // (the only entry that can still be waiting for its partner is that one)
                    curr_e0 = cap; curr_n = 0; cmask = 0; j = 0; i = x - 1; last_mem_beg = 0; pf_e = SM_NONE;
                    st = SM_BWD;
                }
            } else if (st == SM_BWD_PICK) {
                if (i == -1 && curr_n != 0) j = sm_next_needed(j, prev_n, kmask);
                if (j < prev_n) {
                    const uint32_t e = prev_e0 + j;
                    if (e == pf_e) {
                        kx0 = (uint64_t)pfa.y << 32 | pfa.x; kx1 = (uint64_t)pfa.w << 32 | pfa.z;
                        ksz = (uint64_t)pfb.y << 32 | pfb.x; kinfo = (uint64_t)pfb.w << 32 | pfb.z;
                    } else { load_entry(la + e, kx0, kx1, ksz, kinfo); ++list_units; }
                    st = SM_BWD;
                } else if (curr_n != 0 && i != -1) { // next base to the left (smem.c:76-77)
                    if (pend_e != SM_NONE) sm_flush_single(la, slots, pend_e, list_units);
                    prev_e0 = curr_e0; prev_n = curr_n; kmask = cmask; cmask = 0;
                    curr_e0 = prev_e0 == cap ? 0 : cap; // lists start at index 0 of their areas from now on
                    curr_n = 0; j = 0; --i; pf_e = SM_NONE; again = true;
                } else { // this call is over: fm_reverse_fmivec(mem) (smem.c:79), then the next start (smem.c:404-409)
                    if (n_out <= max_mem)
                        for (uint32_t a = out_base, b = n_out; a + 1 < b; ++a) {
                            --b;
                            fmd_intv_t *pa = mem_out + rid * (size_t)max_mem + a, *pb = mem_out + rid * (size_t)max_mem + b;
                            uint64_t a0, a1, a2, a3, b0, b1, b2, b3;
                            load_entry(pa, a0, a1, a2, a3); load_entry(pb, b0, b1, b2, b3);
                            store_entry(pa, b0, b1, b2, b3); store_entry(pb, a0, a1, a2, a3);
                        }
                    x = ret;
                    if (x < stop) { st = SM_START; again = true; }
                    else { n_mem_out[rid] = n_out | (overflow ? 0x80000000u : 0); st = SM_IDLE; }
                }
            }
        }
        if (__ballot(st != SM_IDLE) == 0) { if (__ballot(!exhausted) == 0) break; else continue; }


        // ---- rank2a request: the two ends of the extension of [a, a + size)
        const bool act = st != SM_IDLE;
        const bool fwd = st == SM_FWD || st == SM_FWD_END;
        const uint64_t a0 = fwd ? kx1 : kx0;
        // the symbol this step moves to (FWD_END only looks at '$'); the read is touched one aligned
        // word per four bases, and that load rides under the gather as well
        int c = 0;
        if (st == SM_FWD || (st == SM_BWD && i >= 0)) {
            const uint64_t a = sbase + (uint64_t)i;
            if ((a & ~3ull) != cw_at) { cw_at = a & ~3ull; cw = *(const uint32_t *)(seqs + cw_at); }
        }
        if (st == SM_BWD) { // the next entry needed rides under this gather (last round: curr is non-empty after the first entry in all but odd cases)
            const uint32_t jn = i == -1 ? sm_next_needed(j + 1, prev_n, kmask) : j + 1;
            if (jn < prev_n) {
                const uint4 *pq = (const uint4 *)(la + (prev_e0 + jn));
                pfa = pq[0]; pfb = pq[1]; pf_e = prev_e0 + jn; ++list_units;
            }
        }
        FmdRank2c r = fmd_wave_rank2_fetch_compact(ix, fmd_lds, act ? a0 - 1 : NONE64, act ? a0 - 1 + ksz : NONE64);
        if (st == SM_FWD || (st == SM_BWD && i >= 0)) c = (int)((cw >> (8 * ((sbase + (uint64_t)i) & 3))) & 0xff);
        if (st == SM_FWD) c = comp6(c);
        // narrow interval: everything comes from one 64-position window; a lane whose window straddles
        // two blocks in a two-phase step (the dense slot is reused for the l side) takes the general path
        const bool narrow = act && ksz <= 63 && !(r.two_phase && r.l_sep);
        // the last round of a backward sweep (i = -1) extends by '$' alone: of the six counts of a wide interval it needs that one
        const bool only0 = st == SM_BWD && i == -1;
        uint64_t tk[6] = {0, 0, 0, 0, 0, 0};
        if (r.two_phase && act && !narrow && r.hk) {
            if (only0) tk[0] = fmd_block_rank1(r.bk, r.t, r.nk, 0, r.blk_k);
            else fmd_block_rank6<false>(r.bk, r.t, r.nk, tk, r.blk_k);
        }
        const bool was_two_phase = r.two_phase;
        fmd_wave_l_ready(ix, fmd_lds, r);
        if (!act) continue;

        uint64_t s[6], tkc, tk0;
        bool have_tk0 = false;
        if (narrow) { // all six child sizes from one 64-position window of BWT[a, a + size), one absolute rank
            const uint32_t sh = (uint32_t)a0 & 31;
            uint4 wa, wb, wc;
            grp_window(r.bk, r.t, r.bl, r.tl, r.blk_k, r.blk_l, r.hk, r.l_sep, r.blk_k, r.nk - 1, wa, wb, wc); // window at a0 = (a0 - 1) + 1
            const uint64_t m = (1ull << (int)ksz) - 1;
            const uint64_t X = win64(wa.x, wb.x, wc.x, sh), Y = win64(wa.y, wb.y, wc.y, sh), Z = win64(wa.z, wb.z, wc.z, sh);
            const uint64_t lo = ~Z & m, hi = Z & ~Y & m;
            s[0] = __popcll(lo & ~Y & ~X); s[1] = __popcll(lo & ~Y & X); s[2] = __popcll(lo & Y & ~X); s[3] = __popcll(lo & Y & X);
            s[4] = __popcll(hi & ~X); s[5] = __popcll(hi & X);
            tkc = r.hk ? fmd_block_rank1(r.bk, r.t, r.nk, c, r.blk_k) : 0;
            tk0 = tkc; have_tk0 = c == 0;
        } else {
            uint64_t tl[6] = {0, 0, 0, 0, 0, 0};
            if (only0) {
                if (!was_two_phase && r.hk) tk[0] = fmd_block_rank1(r.bk, r.t, r.nk, 0, r.blk_k);
                if (r.hl) tl[0] = fmd_block_rank1(r.bl, r.tl, r.nl, 0, r.blk_l);
            } else {
                if (!was_two_phase && r.hk) fmd_block_rank6<false>(r.bk, r.t, r.nk, tk, r.blk_k);
                if (r.hl) fmd_block_rank6<false>(r.bl, r.tl, r.nl, tl, r.blk_l);
            }
#pragma unroll
            for (int b = 0; b < 6; ++b) s[b] = tl[b] - tk[b];
            tkc = sel6(c, tk[0], tk[1], tk[2], tk[3], tk[4], tk[5]);
            tk0 = tk[0]; have_tk0 = true;
        }
        const uint64_t sc = sel6(c, s[0], s[1], s[2], s[3], s[4], s[5]);
        // other-strand coordinate of child c: running sum in the order $,T,G,C,A,N (exact.c:81-86)
        const uint64_t base = fwd ? kx0 : kx1;
        uint64_t before = 0;
        if (c != 0) before += s[0];
        if (c == 3 || c == 2 || c == 1 || c == 5) before += s[4];
        if (c == 2 || c == 1 || c == 5) before += s[3];
        if (c == 1 || c == 5) before += s[2];
        if (c == 5) before += s[1];
        const uint64_t rc = base + before;                 // child c
        const uint64_t nxc = ix.cnt[c] + tkc;              // its coordinate on the extended strand
        // the '$' child of a forward extension (pushed by the forward sweep when !self_match)
        if ((st == SM_FWD || st == SM_FWD_END) && !self_match && s[0] && !have_tk0) tk0 = r.hk ? fmd_block_rank1(r.bk, r.t, r.nk, 0, r.blk_k) : 0;

        if (st == SM_FWD) {
            if (sc != ksz) { // change of the interval size (smem.c:25-31)
                if (ksz != s[0]) SM_PUSH_FWD(kx0, kx1, ksz, kinfo);
                if (!self_match && s[0]) SM_PUSH_FWD(base, ix.cnt[0] + tk0, s[0], (uint64_t)i);
            }
            if ((!self_match && sc == 0) || (self_match && sc < 2)) st = SM_BEGIN_BWD; // cannot be extended
            else {
                kx1 = nxc; kx0 = rc; ksz = sc; kinfo = (uint64_t)(i + 1);
                ++i;
                if (i == len) { // reached the end: always push (smem.c:35-37)
                    SM_PUSH_FWD(kx0, kx1, ksz, kinfo);
                    st = self_match ? SM_BEGIN_BWD : SM_FWD_END;
                }
            }
        } else if (st == SM_FWD_END) { // is the last interval terminated by a sentinel? (smem.c:38-43)
            if (s[0]) SM_PUSH_FWD(base, ix.cnt[0] + tk0, s[0], (uint64_t)len);
            st = SM_BEGIN_BWD;
        } else { // SM_BWD: one interval of the list against base q[i] (smem.c:53-74)
            const bool fl_match = s[0] && kx1 < ix.n_seq;
            const bool cont = self_match ? sc > 1 : sc != 0;
            if (!cont || fl_match || i == -1) {
                if (curr_n == 0 || fl_match) {
                    if (fl_match || n_mem == call_base || (uint64_t)(i + 1) < last_mem_beg) { // skip contained matches
                        const uint64_t inf = kinfo | (uint64_t)(s[0] != 0) << 63 | (uint64_t)(i + 1) << 32;
                        if (!full_only || fl_match) { // fl_match: closed by a sentinel on both sides = a whole sequence of the index
                            if (n_out < max_mem) store_entry(mem_out + rid * (size_t)max_mem + n_out, kx0, kx1, ksz, inf);
                            else overflow = true;
                            ++n_out;
                        }
                        ++n_mem;
                        last_mem_beg = (uint64_t)(i + 1);
                    }
                }
            }
            if (cont && (kx1 < ix.n_seq || curr_n == 0 || sc != last_curr_sz)) {
                if (i != -1) { // (nobody reads the list of the last round)
                    sm_push(la, slots, pend_e, list_units, curr_e0 + curr_n, nxc, rc, sc, kinfo);
                    if (curr_n < 64) cmask |= (uint64_t)(kx1 < ix.n_seq) << curr_n;
                }
                last_curr_sz = sc;
                ++curr_n;
            }
            ++j;
            st = SM_BWD_PICK;
        }
    }
#if FMD_COUNT_LINES
    if (list_units) atomicAdd(ix.stat + (blockIdx.x & (FMD_STAT_SLOTS - 1)) * FMD_STAT_STRIDE + 1, (unsigned long long)list_units);   // kind 1 in this kernel: 32-byte list entries
#endif
}

// *mixed := 1 when the n sequences do not all have the length of the first
__global__ void k_smem_mixed(size_t n, const uint64_t *__restrict__ off, uint32_t *__restrict__ mixed)
{
    const uint64_t len0 = off[1] - off[0];
    bool any = false;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) any |= off[i + 1] - off[i] != len0;
    if (__ballot(any) && fmd_lane() == 0) *mixed = 1;
}

static size_t smem_lanes(size_t n)
{
    const size_t waves = (n + 63) / 64;
    return (waves < SMEM_MAX_WAVES ? waves : SMEM_MAX_WAVES) * 64;
}

extern "C" size_t fmd_smem_work_bytes(size_t n, uint32_t max_len)
{
    return smem_lanes(n) * 2 * (size_t)(2 * max_len + 2) * sizeof(fmd_intv_t) + 256;
}

extern "C" int fmd_smem_dev(fmd_dev_t *h, void *stream_, size_t n, const uint8_t *d_seqs, const uint64_t *d_off, int self_match,
                            uint32_t max_len, uint32_t max_mem, fmd_intv_t *d_mem, uint32_t *d_n_mem, void *d_work, size_t work_bytes)
{
    if (!h || (n && (!d_seqs || !d_off || !d_mem || !d_n_mem || !d_work)) || max_len == 0 || max_mem == 0) return FMD_E_ARG;
    if (n == 0) return FMD_OK;
    if (n >= 0xffffff00ull || work_bytes < fmd_smem_work_bytes(n, max_len) || ((uintptr_t)d_seqs & 3)) return FMD_E_ARG;
    FMD_HIP_TRY(hipSetDevice(h->device));
    hipStream_t st = (hipStream_t)stream_;
    uint32_t *q = fmd_next_queue(h, st);
    int grid = fmd_grid_for_lds(h, n, SMEM_LDS_BYTES);
    if (grid > SMEM_MAX_WAVES) grid = SMEM_MAX_WAVES;
    // When do idle lanes take new reads?  Reads of ONE length that start together stay together: 64 forward sweeps in step, then 64
    // backward sweeps, and a wave step pays for one of the two code paths (and for the wide-interval path only while every lane is
    // on it) instead of for all of them -- worth more than the lanes that wait for the longest list of the wave (50 M x 100 bp:
    // 204 ms against 221).  Reads of mixed lengths drift apart anyway and would only wait: groups of 8; and so would reads of one
    // length whose work differs (self_match: nine calls per read, 471 ms against 377 for 2*10^7 reads), so a wave whose lanes
    // do not all come back within `patience` steps of the first gives the waiting up for the rest of the launch.  The lengths are
    // looked at on the device (one streaming pass over `off`); FMD_SMEM_REFILL=k fixes the group size (64: always wait).
    const char *rf = getenv("FMD_SMEM_REFILL"), *pt = getenv("FMD_SMEM_PATIENCE");
    uint32_t *uniform = (uint32_t *)((uint8_t *)d_work + fmd_smem_work_bytes(n, max_len) - 16);   // (inside the slack behind the lists)
    if (!rf) {
        FMD_HIP_TRY(hipMemsetAsync(uniform, 0, 4, st));
        k_smem_mixed<<<512, 256, 0, st>>>(n, d_off, uniform);
    }
    k_smem<<<grid, 64, 0, st>>>(fmd_view(h), n, d_seqs, d_off, self_match ? 1 : 0, 2 * max_len + 2,
                                (fmd_intv_t *)(((uintptr_t)d_work + 63) & ~(uintptr_t)63), max_mem, d_mem, d_n_mem, q, rf ? atoi(rf) : 8, nullptr,
                                rf ? nullptr : uniform, pt ? atoi(pt) : 96);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { fmd_set_hip_error(e, "k_smem"); return FMD_E_HIP; }
    return FMD_OK;
}

// Windows of long sequences (contigs): item i runs the fm6_smem chain (smem.c:397-410, = fm6_miter_next,
// smem.c:96-102) from `start` while x < stop over the sequence at d_seqs + seq_off.  Every
// fm6_smem1_core call is a pure function of (sequence, x); the calls the reference makes are those at
// the positions of the chain from 0 (x -> x + reach[x], fmd_reach_dev), one item each.  max_len bounds
// the length of a match (the longest sequence in the index + 1).
extern "C" int fmd_smem_win_dev(fmd_dev_t *h, void *stream_, size_t n, const uint8_t *d_seqs, const fmd_smem_win_t *d_wins, int self_match,
                                uint32_t max_len, uint32_t max_mem, fmd_intv_t *d_mem, uint32_t *d_n_mem, void *d_work, size_t work_bytes)
{
    if (!h || (n && (!d_seqs || !d_wins || !d_mem || !d_n_mem || !d_work)) || max_len == 0 || max_mem == 0) return FMD_E_ARG;
    if (n == 0) return FMD_OK;
    if (n >= 0xffffff00ull || work_bytes < fmd_smem_work_bytes(n, max_len) || ((uintptr_t)d_seqs & 3)) return FMD_E_ARG;
    FMD_HIP_TRY(hipSetDevice(h->device));
    hipStream_t st = (hipStream_t)stream_;
    uint32_t *q = fmd_next_queue(h, st);
    int grid = fmd_grid_for_lds(h, n, SMEM_LDS_BYTES);
    if (grid > SMEM_MAX_WAVES) grid = SMEM_MAX_WAVES;
    k_smem<<<grid, 64, 0, st>>>(fmd_view(h), n, d_seqs, nullptr, self_match ? 1 : 0, 2 * max_len + 2,
                                (fmd_intv_t *)(((uintptr_t)d_work + 63) & ~(uintptr_t)63), max_mem, d_mem, d_n_mem, q, 1, d_wins, nullptr, 0);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { fmd_set_hip_error(e, "k_smem (windows)"); return FMD_E_HIP; }
    return FMD_OK;
}

struct SBuf { void *p = nullptr; int alloc(size_t b) { return hipMalloc(&p, b ? b : 16) == hipSuccess ? FMD_OK : FMD_E_NOMEM; } ~SBuf() { if (p) hipFree(p); } };

extern "C" int fmd_smem_batch(fmd_dev_t *h, size_t n, const uint8_t *seqs, const uint64_t *off, int self_match, uint32_t max_len,
                              uint32_t max_mem, fmd_intv_t *mem, uint32_t *n_mem)
{
    if (!h || (n && (!seqs || !off || !mem || !n_mem))) return FMD_E_ARG;
    if (n == 0) return FMD_OK;
    FMD_HIP_TRY(hipSetDevice(h->device));
    SBuf ds, doff, dm, dn, dw;
    const size_t wb = fmd_smem_work_bytes(n, max_len);
    if (ds.alloc(off[n] + 8) || doff.alloc((n + 1) * 8) || dm.alloc(n * (size_t)max_mem * sizeof(fmd_intv_t)) || dn.alloc(n * 4) || dw.alloc(wb))
        return FMD_E_NOMEM;
    FMD_HIP_TRY(hipMemcpy(ds.p, seqs, off[n], hipMemcpyHostToDevice));
    FMD_HIP_TRY(hipMemcpy(doff.p, off, (n + 1) * 8, hipMemcpyHostToDevice));
    FMD_HIP_TRY(hipMemset(dm.p, 0, n * (size_t)max_mem * sizeof(fmd_intv_t)));
    int rc = fmd_smem_dev(h, nullptr, n, (uint8_t *)ds.p, (uint64_t *)doff.p, self_match, max_len, max_mem, (fmd_intv_t *)dm.p, (uint32_t *)dn.p, dw.p, wb);
    if (rc) return rc;
    FMD_HIP_TRY(hipMemcpy(mem, dm.p, n * (size_t)max_mem * sizeof(fmd_intv_t), hipMemcpyDeviceToHost));
    FMD_HIP_TRY(hipMemcpy(n_mem, dn.p, n * 4, hipMemcpyDeviceToHost));
    return FMD_OK;
}

extern "C" int fmd_smem_win_batch(fmd_dev_t *h, size_t n, const uint8_t *seqs, uint64_t seq_bytes, const fmd_smem_win_t *wins, int self_match,
                                  uint32_t max_len, uint32_t max_mem, fmd_intv_t *mem, uint32_t *n_mem)
{
    if (!h || (n && (!seqs || !wins || !mem || !n_mem))) return FMD_E_ARG;
    if (n == 0) return FMD_OK;
    FMD_HIP_TRY(hipSetDevice(h->device));
    SBuf ds, dwin, dm, dn, dw;
    const size_t wb = fmd_smem_work_bytes(n, max_len);
    if (ds.alloc(seq_bytes + 8) || dwin.alloc(n * sizeof(fmd_smem_win_t)) || dm.alloc(n * (size_t)max_mem * sizeof(fmd_intv_t)) || dn.alloc(n * 4) || dw.alloc(wb))
        return FMD_E_NOMEM;
    FMD_HIP_TRY(hipMemcpy(ds.p, seqs, seq_bytes, hipMemcpyHostToDevice));
    FMD_HIP_TRY(hipMemcpy(dwin.p, wins, n * sizeof(fmd_smem_win_t), hipMemcpyHostToDevice));
    int rc = fmd_smem_win_dev(h, nullptr, n, (uint8_t *)ds.p, (fmd_smem_win_t *)dwin.p, self_match, max_len, max_mem, (fmd_intv_t *)dm.p, (uint32_t *)dn.p, dw.p, wb);
    if (rc) return rc;
    FMD_HIP_TRY(hipMemcpy(mem, dm.p, n * (size_t)max_mem * sizeof(fmd_intv_t), hipMemcpyDeviceToHost));
    FMD_HIP_TRY(hipMemcpy(n_mem, dn.p, n * 4, hipMemcpyDeviceToHost));
    return FMD_OK;
}
