// fmd_smem.hip -- super-maximal exact matches: fm6_smem1_core (smem.c:13-80) driven as fm6_smem
// does (smem.c:397-410), i.e. what `fermi exact` prints (cmd.c:319-327).  One lane per read on the
// wave engine; the forward sweep is a chain of forward extensions, the backward sweep walks the
// whole candidate list once per base.  Candidate lists live in an HBM work area (two lists of
// 2*max_len+2 entries per read); SMEMs are written to the caller's array in the reference's order.
#include <stdlib.h>
#include <string.h>
#include "fmd_internal.h"

#define NONE64 (~0ull)
#define MASK30 0x3fffffffull

__device__ __forceinline__ int s_comp6(int c) { return (c >= 1 && c <= 4) ? 5 - c : c; }

template <class T>
__device__ __forceinline__ T s_sel6(int c, T a0, T a1, T a2, T a3, T a4, T a5)
{
    T r = a0;
    r = c == 1 ? a1 : r; r = c == 2 ? a2 : r; r = c == 3 ? a3 : r; r = c == 4 ? a4 : r; r = c == 5 ? a5 : r;
    return r;
}

__device__ __forceinline__ void s_load(const fmd_intv_t *e, uint64_t &x0, uint64_t &x1, uint64_t &sz, uint64_t &info)
{
    const uint4 *q = (const uint4 *)e;
    const uint4 a = q[0], b = q[1];
    x0 = (uint64_t)a.y << 32 | a.x; x1 = (uint64_t)a.w << 32 | a.z;
    sz = (uint64_t)b.y << 32 | b.x; info = (uint64_t)b.w << 32 | b.z;
}
__device__ __forceinline__ void s_store(fmd_intv_t *e, uint64_t x0, uint64_t x1, uint64_t sz, uint64_t info)
{
    uint4 *q = (uint4 *)e;
    q[0] = make_uint4((uint32_t)x0, (uint32_t)(x0 >> 32), (uint32_t)x1, (uint32_t)(x1 >> 32));
    q[1] = make_uint4((uint32_t)sz, (uint32_t)(sz >> 32), (uint32_t)info, (uint32_t)(info >> 32));
}

enum { SM_IDLE = 0, SM_START, SM_BEGIN_BWD, SM_BWD_PICK, SM_FWD, SM_FWD_END, SM_BWD };

__global__ __launch_bounds__(64) void k_smem(FmdIndexView ix, size_t n, const uint8_t *__restrict__ seqs, const uint64_t *__restrict__ off,
                                             int self_match, uint32_t cap, fmd_intv_t *__restrict__ work, uint32_t max_mem,
                                             fmd_intv_t *__restrict__ mem_out, uint32_t *__restrict__ n_mem_out, uint32_t *__restrict__ queue)
{
    FMD_DECLARE_WAVE_LDS();
    size_t rid = 0;
    const uint8_t *q = nullptr;
    int st = SM_IDLE, len = 0, x = 0, i = 0, ret = 0;
    uint32_t prev_n = 0, curr_n = 0, j = 0, n_mem = 0, call_base = 0;
    fmd_intv_t *la = nullptr, *lb = nullptr, *prev = nullptr, *curr = nullptr;
    uint64_t kx0 = 0, kx1 = 0, ksz = 0, kinfo = 0;   // ik (forward sweep) / p (backward sweep)
    uint64_t last_curr_sz = 0, last_mem_beg = 0;
    bool exhausted = false, overflow = false;

    for (;;) {
        // ---- refill
        const uint64_t want = __ballot(st == SM_IDLE && !exhausted);
        if (want) {
            uint32_t first = 0;
            if (fmd_lane() == 0) first = atomicAdd(queue, (uint32_t)__popcll(want));
            first = (uint32_t)__builtin_amdgcn_readfirstlane((int)first);
            if (st == SM_IDLE && !exhausted) {
                const size_t my = (size_t)first + __popcll(want & ((1ull << fmd_lane()) - 1));
                if (my < n) {
                    rid = my; q = seqs + off[my]; len = (int)(off[my + 1] - off[my]);
                    la = work + rid * 2 * (size_t)cap; lb = la + cap;
                    n_mem = 0; overflow = false; x = 0;
                    if (len <= 0) n_mem_out[rid] = 0;
                    else if (2 * (uint32_t)len + 2 > cap) n_mem_out[rid] = 0x80000000u; // longer than max_len
                    else st = SM_START;
                } else exhausted = true;
            }
        }
        // ---- transitions that need no rank
        bool again = st == SM_START || st == SM_BEGIN_BWD || st == SM_BWD_PICK;
        while (again) {
            again = false;
            if (st == SM_START) { // fm6_smem1_core prologue (smem.c:19-21)
                const int c = q[x];
                kx0 = ix.cnt[c]; kx1 = ix.cnt[s_comp6(c)]; ksz = ix.cnt[c + 1] - ix.cnt[c]; kinfo = (uint64_t)(x + 1);
                curr_n = 0; call_base = n_mem; i = x + 1;
                if (ksz == 0) { // the reference dereferences an empty list here (undefined); stop this read
                    n_mem_out[rid] = n_mem | (overflow ? 0x80000000u : 0);
                    st = SM_IDLE;
                } else if (i < len) st = SM_FWD;
                else { // x is the last base: push the interval (smem.c:35-37); list is written back to front
                    s_store(la + (cap - 1 - curr_n), kx0, kx1, ksz, kinfo); ++curr_n;
                    st = self_match ? SM_BEGIN_BWD : SM_FWD_END;
                    again = st == SM_BEGIN_BWD;
                }
            } else if (st == SM_BEGIN_BWD) { // the forward list, already reversed, becomes prev (smem.c:45-50)
                prev = la + (cap - curr_n); prev_n = curr_n;
                uint64_t t0, t1, t2, t3;
                s_load(prev, t0, t1, t2, t3);
                ret = (int)t3;
                curr = lb; curr_n = 0; j = 0; i = x - 1; last_mem_beg = 0;
                st = SM_BWD_PICK; again = true;
            } else if (st == SM_BWD_PICK) {
                if (j < prev_n) { s_load(prev + j, kx0, kx1, ksz, kinfo); st = SM_BWD; }
                else if (curr_n != 0 && i != -1) { // next base to the left (smem.c:76-77)
                    prev = curr; prev_n = curr_n;
                    curr = (prev == lb) ? la : lb; // lists start at index 0 of their areas from now on
                    curr_n = 0; j = 0; --i; again = true;
                } else { // this call is over: fm_reverse_fmivec(mem) (smem.c:79), then the next start (smem.c:404-409)
                    if (n_mem <= max_mem)
                        for (uint32_t a = call_base, b = n_mem; a + 1 < b; ++a) {
                            --b;
                            fmd_intv_t *pa = mem_out + rid * (size_t)max_mem + a, *pb = mem_out + rid * (size_t)max_mem + b;
                            uint64_t a0, a1, a2, a3, b0, b1, b2, b3;
                            s_load(pa, a0, a1, a2, a3); s_load(pb, b0, b1, b2, b3);
                            s_store(pa, b0, b1, b2, b3); s_store(pb, a0, a1, a2, a3);
                        }
                    x = ret;
                    if (x < len) { st = SM_START; again = true; }
                    else { n_mem_out[rid] = n_mem | (overflow ? 0x80000000u : 0); st = SM_IDLE; }
                }
            }
        }
        if (__ballot(st != SM_IDLE) == 0) { if (__ballot(!exhausted) == 0) break; else continue; }

        // ---- rank2a request
        uint64_t qk = NONE64, ql = NONE64;
        const bool fwd = st == SM_FWD || st == SM_FWD_END;
        if (st != SM_IDLE) { const uint64_t a = fwd ? kx1 : kx0; qk = a - 1; ql = a - 1 + ksz; }
        const FmdRank2 r = fmd_wave_rank2_fetch(ix, fmd_lds, qk, ql);
        if (st != SM_IDLE) {
            uint64_t tk[6] = {0, 0, 0, 0, 0, 0}, tl[6] = {0, 0, 0, 0, 0, 0};
            if (r.hk) fmd_block_rank6<false>(r.bk, r.t, r.nk, tk);
            if (r.hl) fmd_block_rank6<false>(r.bl, r.tl, r.nl, tl);
            uint64_t s[6];
#pragma unroll
            for (int c = 0; c < 6; ++c) s[c] = tl[c] - tk[c];
            // other-strand coordinate of child c: running sum in the order $,T,G,C,A,N (exact.c:81-86)
            const uint64_t base = fwd ? kx0 : kx1;
            const uint64_t r0 = base, r4 = r0 + s[0], r3 = r4 + s[4], r2 = r3 + s[3], r1 = r2 + s[2], r5 = r1 + s[1];

            if (st == SM_FWD) {
                const int c = s_comp6(q[i]);
                const uint64_t sc = s_sel6(c, s[0], s[1], s[2], s[3], s[4], s[5]);
                if (sc != ksz) { // change of the interval size (smem.c:25-31)
                    if (ksz != s[0]) { s_store(la + (cap - 1 - curr_n), kx0, kx1, ksz, kinfo); ++curr_n; }
                    if (!self_match && s[0]) { s_store(la + (cap - 1 - curr_n), r0, ix.cnt[0] + tk[0], s[0], (uint64_t)i); ++curr_n; }
                }
                if ((!self_match && sc == 0) || (self_match && sc < 2)) st = SM_BEGIN_BWD; // cannot be extended
                else {
                    kx1 = s_sel6(c, ix.cnt[0], ix.cnt[1], ix.cnt[2], ix.cnt[3], ix.cnt[4], ix.cnt[5]) +
                          s_sel6(c, tk[0], tk[1], tk[2], tk[3], tk[4], tk[5]);
                    kx0 = s_sel6(c, r0, r1, r2, r3, r4, r5); ksz = sc; kinfo = (uint64_t)(i + 1);
                    ++i;
                    if (i == len) { // reached the end: always push (smem.c:35-37)
                        s_store(la + (cap - 1 - curr_n), kx0, kx1, ksz, kinfo); ++curr_n;
                        st = self_match ? SM_BEGIN_BWD : SM_FWD_END;
                    }
                }
            } else if (st == SM_FWD_END) { // is the last interval terminated by a sentinel? (smem.c:38-43)
                if (s[0]) { s_store(la + (cap - 1 - curr_n), r0, ix.cnt[0] + tk[0], s[0], (uint64_t)len); ++curr_n; }
                st = SM_BEGIN_BWD;
            } else { // SM_BWD: one interval of the list against base q[i] (smem.c:53-74)
                const int c = i < 0 ? 0 : q[i];
                const uint64_t sc = s_sel6(c, s[0], s[1], s[2], s[3], s[4], s[5]);
                const bool fl_match = s[0] && kx1 < ix.n_seq;
                const bool cont = self_match ? sc > 1 : sc != 0;
                if (!cont || fl_match || i == -1) {
                    if (curr_n == 0 || fl_match) {
                        if (fl_match || n_mem == call_base || (uint64_t)(i + 1) < last_mem_beg) { // skip contained matches
                            const uint64_t inf = kinfo | (uint64_t)(s[0] != 0) << 63 | (uint64_t)(i + 1) << 32;
                            if (n_mem < max_mem) s_store(mem_out + rid * (size_t)max_mem + n_mem, kx0, kx1, ksz, inf);
                            else overflow = true;
                            ++n_mem;
                            last_mem_beg = (uint64_t)(i + 1);
                        }
                    }
                }
                if (cont && (kx1 < ix.n_seq || curr_n == 0 || sc != last_curr_sz)) {
                    const uint64_t nx0 = s_sel6(c, ix.cnt[0], ix.cnt[1], ix.cnt[2], ix.cnt[3], ix.cnt[4], ix.cnt[5]) +
                                         s_sel6(c, tk[0], tk[1], tk[2], tk[3], tk[4], tk[5]);
                    const uint64_t nx1 = s_sel6(c, r0, r1, r2, r3, r4, r5);
                    s_store(curr + curr_n, nx0, nx1, sc, kinfo);
                    last_curr_sz = sc;
                    ++curr_n;
                }
                ++j;
                st = SM_BWD_PICK;
            }
        }
    }
}

extern "C" size_t fmd_smem_work_bytes(size_t n, uint32_t max_len)
{
    return n * 2 * (size_t)(2 * max_len + 2) * sizeof(fmd_intv_t) + 256;
}

extern "C" int fmd_smem_dev(fmd_dev_t *h, void *stream_, size_t n, const uint8_t *d_seqs, const uint64_t *d_off, int self_match,
                            uint32_t max_len, uint32_t max_mem, fmd_intv_t *d_mem, uint32_t *d_n_mem, void *d_work, size_t work_bytes)
{
    if (!h || (n && (!d_seqs || !d_off || !d_mem || !d_n_mem || !d_work)) || max_len == 0 || max_mem == 0) return FMD_E_ARG;
    if (n == 0) return FMD_OK;
    if (n >= 0xffffff00ull || work_bytes < fmd_smem_work_bytes(n, max_len)) return FMD_E_ARG;
    FMD_HIP_TRY(hipSetDevice(h->device));
    hipStream_t st = (hipStream_t)stream_;
    uint32_t *q = fmd_next_queue(h, st);
    k_smem<<<fmd_grid_for(h, n), 64, 0, st>>>(fmd_view(h), n, d_seqs, d_off, self_match ? 1 : 0, 2 * max_len + 2,
                                              (fmd_intv_t *)d_work, max_mem, d_mem, d_n_mem, q);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { fmd_set_hip_error(e, "k_smem"); return FMD_E_HIP; }
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
