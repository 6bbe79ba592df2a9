// fmd_kmer.hip -- the k-mer harvest of `fermi correct`: fm6_traverse (exact.c:141-171) +
// ec_collect (correct.c:35-87), 90 % of `correct`'s CPU time.
//
// The reference walks the k-mer trie depth-first with an explicit stack per suffix bucket.  The
// table it builds -- per bucket, key = the remaining bases + best next base, val = ratio/rest
// code -- does not depend on the visiting order (only khash insertion order does), so the GPU
// expands the trie LEVEL BY LEVEL instead: frontier(d) -> one backward extension per node on the
// wave engine -> children with enough occurrences appended to frontier(d+1) by wave-aggregated
// atomics.  No per-lane stacks, every lane busy, every step the same phase.  The k-mer travels in
// the node's info word as K = sum (base_d - 1) << 2d  (base_0 = rightmost base of the k-mer):
// bucket = K mod 4^SUF_LEN (fm6_traverse's index, exact.c:160) and key = (K >> 2*SUF_LEN) << 2 |
// best (correct.c:71-73).
#include <stdlib.h>
#include <string.h>
#include "fmd_internal.h"

#define NONE64 (~0ull)

// counters in device memory: [0..63] frontier sizes per depth, [64] #output, [65] overflow flag,
// [66] cnt[0] (k-mers kept), [67] cnt[1] (informative ones), correct.c:64-69
#define KM_OUT 64
#define KM_OVF 65
#define KM_CNT0 66
#define KM_CNT1 67
#define KM_WORDS 72

__device__ __forceinline__ void km_extend_back(const FmdIndexView &ix, uint4 *lds, bool active, uint64_t x0, uint64_t sz,
                                               uint64_t tk[6], uint64_t s[6])
{
    const FmdRank2 r = fmd_wave_rank2_fetch(ix, lds, active ? x0 - 1 : NONE64, active ? x0 - 1 + sz : NONE64);
    uint64_t tl[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int c = 0; c < 6; ++c) tk[c] = 0;
    if (active) {
        if (r.hk) fmd_block_rank6<false>(r.bk, r.t, r.nk, tk);
        if (r.hl) fmd_block_rank6<false>(r.bl, r.tl, r.nl, tl);
    }
#pragma unroll
    for (int c = 0; c < 6; ++c) s[c] = tl[c] - tk[c];
}

// one trie level: nodes at depth d -> children at depth d+1
__global__ __launch_bounds__(64) void k_kmer_level(FmdIndexView ix, int d, int suf_len, int min_occ, const fmd_intv_t *__restrict__ in,
                                                   fmd_intv_t *__restrict__ out, uint64_t cap, unsigned long long *__restrict__ ctr)
{
    FMD_DECLARE_WAVE_LDS();
    const int lane = fmd_lane();
    const uint64_t n = ctr[d] < cap ? ctr[d] : cap;   // an overflowing level counted more than it stored
    const uint64_t thr = (d + 1 <= suf_len) ? 1 : (uint64_t)min_occ; // exact.c:159 vs correct.c:78
    const uint64_t stride = (uint64_t)gridDim.x * 64;
    for (uint64_t base = (uint64_t)blockIdx.x * 64; base < n; base += stride) {
        const uint64_t i = base + lane;
        const bool act = i < n;
        uint64_t x0 = 0, x1 = 0, sz = 0, K = 0;
        if (act) {
            const uint4 *q = (const uint4 *)(in + i);
            const uint4 a = q[0], b = q[1];
            x0 = (uint64_t)a.y << 32 | a.x; x1 = (uint64_t)a.w << 32 | a.z;
            sz = (uint64_t)b.y << 32 | b.x; K = (uint64_t)b.w << 32 | b.z;
        }
        uint64_t tk[6], s[6];
        km_extend_back(ix, fmd_lds, act, x0, sz, tk, s);
        // children c = 1..4 (ambiguous bases are skipped, correct.c:77); x[1] = running sum in
        // the order $,T,G,C,A (exact.c:81-86)
        const bool has1 = act && s[1] >= thr, has2 = act && s[2] >= thr, has3 = act && s[3] >= thr, has4 = act && s[4] >= thr;
        const uint64_t m1 = __ballot(has1), m2 = __ballot(has2), m3 = __ballot(has3), m4 = __ballot(has4);
        const uint32_t tot = (uint32_t)(__popcll(m1) + __popcll(m2) + __popcll(m3) + __popcll(m4));
        if (tot == 0) continue;
        unsigned long long first = 0;
        if (lane == 0) first = atomicAdd(&ctr[d + 1], (unsigned long long)tot);
        first = ((unsigned long long)__builtin_amdgcn_readfirstlane((int)(first >> 32)) << 32) |
                (unsigned int)__builtin_amdgcn_readfirstlane((int)first);
        const uint64_t lt = (1ull << lane) - 1;
        uint64_t o1 = first + __popcll(m1 & lt);
        uint64_t o2 = first + __popcll(m1) + __popcll(m2 & lt);
        uint64_t o3 = first + __popcll(m1) + __popcll(m2) + __popcll(m3 & lt);
        uint64_t o4 = first + __popcll(m1) + __popcll(m2) + __popcll(m3) + __popcll(m4 & lt);
        const uint64_t x1_4 = x1 + s[0], x1_3 = x1_4 + s[4], x1_2 = x1_3 + s[3], x1_1 = x1_2 + s[2];
#define KM_PUSH(c, has, o, x1c)                                                                   \
        if (has) {                                                                                \
            if (o < cap) {                                                                        \
                const uint64_t nx0 = ix.cnt[c] + tk[c], nk = K | (uint64_t)(c - 1) << (2 * d);    \
                uint4 *q = (uint4 *)(out + o);                                                    \
                q[0] = make_uint4((uint32_t)nx0, (uint32_t)(nx0 >> 32), (uint32_t)(x1c), (uint32_t)((x1c) >> 32)); \
                q[1] = make_uint4((uint32_t)s[c], (uint32_t)(s[c] >> 32), (uint32_t)nk, (uint32_t)(nk >> 32));     \
            } else ctr[KM_OVF] = 1;                                                               \
        }
        KM_PUSH(1, has1, o1, x1_1) KM_PUSH(2, has2, o2, x1_2) KM_PUSH(3, has3, o3, x1_3) KM_PUSH(4, has4, o4, x1_4)
#undef KM_PUSH
    }
}

// nodes at depth w: pick the most frequent next base and emit (bucket, key, val), correct.c:56-75
__global__ __launch_bounds__(64) void k_kmer_emit(FmdIndexView ix, int w, int suf_len, int min_occ, const fmd_intv_t *__restrict__ in,
                                                  uint32_t *__restrict__ o_bucket, uint32_t *__restrict__ o_key,
                                                  uint8_t *__restrict__ o_val, uint64_t cap, unsigned long long *__restrict__ ctr)
{
    FMD_DECLARE_WAVE_LDS();
    const int lane = fmd_lane();
    const uint64_t n = ctr[w] < cap ? ctr[w] : cap;
    const uint64_t stride = (uint64_t)gridDim.x * 64;
    for (uint64_t base = (uint64_t)blockIdx.x * 64; base < n; base += stride) {
        const uint64_t i = base + lane;
        const bool act = i < n;
        uint64_t x0 = 0, sz = 0, K = 0;
        if (act) {
            const uint4 *q = (const uint4 *)(in + i);
            const uint4 a = q[0], b = q[1];
            x0 = (uint64_t)a.y << 32 | a.x;
            sz = (uint64_t)b.y << 32 | b.x; K = (uint64_t)b.w << 32 | b.z;
        }
        uint64_t tk[6], s[6];
        km_extend_back(ix, fmd_lds, act, x0, sz, tk, s);
        uint64_t mx = 0; int max_c = 6;
#pragma unroll
        for (int c = 1; c <= 4; ++c) if (s[c] > mx) { mx = s[c]; max_c = c; }
        const bool keep = act && mx >= (uint64_t)min_occ;
        const uint64_t rest = sz - mx - s[0] - s[5];
        double r = rest == 0 ? (double)mx : (double)mx / (double)rest;   // IEEE double divide, as on the host
        if (r > 31.) r = 31.;
        const bool informative = keep && rest <= 7 && r >= (double)min_occ;
        const uint64_t mk = __ballot(keep), mi = __ballot(informative);
        if (mk == 0) continue;
        unsigned long long first = 0;
        if (lane == 0) {
            first = atomicAdd(&ctr[KM_OUT], (unsigned long long)__popcll(mk));
            atomicAdd(&ctr[KM_CNT0], (unsigned long long)__popcll(mk));
            if (mi) atomicAdd(&ctr[KM_CNT1], (unsigned long long)__popcll(mi));
        }
        first = ((unsigned long long)__builtin_amdgcn_readfirstlane((int)(first >> 32)) << 32) |
                (unsigned int)__builtin_amdgcn_readfirstlane((int)first);
        if (keep) {
            const uint64_t o = first + __popcll(mk & ((1ull << lane) - 1));
            if (o < cap) {
                o_bucket[o] = (uint32_t)(K & ((1ull << (2 * suf_len)) - 1));
                o_key[o] = (uint32_t)(K >> (2 * suf_len)) << 2 | (uint32_t)(max_c - 1);
                o_val[o] = (uint8_t)((int)(r + .499) << 3 | (int)(rest < 7 ? rest : 7));
            } else ctr[KM_OVF] = 1;
        }
    }
}

extern "C" size_t fmd_kmer_work_bytes(uint64_t cap_frontier)
{
    return 2 * cap_frontier * sizeof(fmd_intv_t) + KM_WORDS * 8 + 512;
}

// d_status (device, 4 x u64): [0] number of (bucket,key,val) triples, [1] overflow flag (then the
// result is incomplete: call again with a larger cap), [2] cnt[0], [3] cnt[1] of correct.c:64-69.
extern "C" int fmd_kmer_collect_dev(fmd_dev_t *h, void *stream_, int w, int min_occ, int suf_len, void *d_work, size_t work_bytes,
                                    uint64_t cap, uint32_t *d_bucket, uint32_t *d_key, uint8_t *d_val, uint64_t *d_status)
{
    if (!h || !d_work || !d_bucket || !d_key || !d_val || !d_status) return FMD_E_ARG;
    if (w < 2 || w > 27 || suf_len < 1 || suf_len >= w || min_occ < 1 || w - suf_len > 15 || cap < 4) return FMD_E_ARG; // MAX_KMER 27 (correct.c:303)
    if (work_bytes < fmd_kmer_work_bytes(cap)) return FMD_E_ARG;
    FMD_HIP_TRY(hipSetDevice(h->device));
    hipStream_t st = (hipStream_t)stream_;
    unsigned long long *ctr = (unsigned long long *)d_work;
    fmd_intv_t *fa = (fmd_intv_t *)(((uintptr_t)((uint8_t *)d_work + KM_WORDS * 8) + 255) & ~(uintptr_t)255);
    fmd_intv_t *fb = fa + cap;
    FMD_HIP_TRY(hipMemsetAsync(ctr, 0, KM_WORDS * 8, st));
    // depth 1: the four single-base intervals (exact.c:153-155: fm6_set_intv for the root)
    fmd_intv_t seed[4]; unsigned long long n1 = 0;
    for (int c = 1; c <= 4; ++c) {
        const uint64_t sz = h->cnt[c + 1] - h->cnt[c];
        if (sz == 0) continue;
        seed[n1].x[0] = h->cnt[c]; seed[n1].x[1] = h->cnt[5 - c]; seed[n1].x[2] = sz; seed[n1].info = (uint64_t)(c - 1);
        ++n1;
    }
    FMD_HIP_TRY(hipMemcpyAsync(fa, seed, n1 * sizeof(fmd_intv_t), hipMemcpyHostToDevice, st));
    FMD_HIP_TRY(hipMemcpyAsync(ctr + 1, &n1, 8, hipMemcpyHostToDevice, st));
    FMD_HIP_TRY(hipStreamSynchronize(st)); // seed[] is a stack buffer
    const FmdIndexView ix = fmd_view(h);
    const int grid = h->n_cu * 10;
    fmd_intv_t *in = fa, *out = fb;
    for (int d = 1; d < w; ++d) {
        k_kmer_level<<<grid, 64, 0, st>>>(ix, d, suf_len, min_occ, in, out, cap, ctr);
        fmd_intv_t *t = in; in = out; out = t;
    }
    k_kmer_emit<<<grid, 64, 0, st>>>(ix, w, suf_len, min_occ, in, d_bucket, d_key, d_val, cap, ctr);
    FMD_HIP_TRY(hipMemcpyAsync(d_status, ctr + KM_OUT, 4 * 8, hipMemcpyDeviceToDevice, st));
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { fmd_set_hip_error(e, "kmer kernels"); return FMD_E_HIP; }
    return FMD_OK;
}

// Host form: grows the capacity until nothing overflows; outputs are malloc'ed (fmd_host_free).
extern "C" int fmd_kmer_collect(fmd_dev_t *h, int w, int min_occ, int suf_len, uint32_t **bucket, uint32_t **key, uint8_t **val,
                                uint64_t *n, int64_t cnt[2])
{
    if (!h || !bucket || !key || !val || !n || !cnt) return FMD_E_ARG;
    FMD_HIP_TRY(hipSetDevice(h->device));
    uint64_t cap = 1u << 20;
    for (int attempt = 0; attempt < 16; ++attempt, cap *= 4) {
        void *work = nullptr, *db = nullptr, *dk = nullptr, *dv = nullptr, *ds = nullptr;
        const size_t wb = fmd_kmer_work_bytes(cap);
        int rc = FMD_OK;
        if (hipMalloc(&work, wb) != hipSuccess || hipMalloc(&db, cap * 4) != hipSuccess || hipMalloc(&dk, cap * 4) != hipSuccess ||
            hipMalloc(&dv, cap) != hipSuccess || hipMalloc(&ds, 32) != hipSuccess) rc = FMD_E_NOMEM;
        uint64_t status[4] = {0, 0, 0, 0};
        if (rc == FMD_OK) rc = fmd_kmer_collect_dev(h, nullptr, w, min_occ, suf_len, work, wb, cap, (uint32_t *)db, (uint32_t *)dk, (uint8_t *)dv, (uint64_t *)ds);
        if (rc == FMD_OK && hipMemcpy(status, ds, 32, hipMemcpyDeviceToHost) != hipSuccess) rc = FMD_E_HIP;
        if (rc == FMD_OK && status[1] == 0) {
            const uint64_t m = status[0];
            *bucket = (uint32_t *)malloc(m * 4 + 4); *key = (uint32_t *)malloc(m * 4 + 4); *val = (uint8_t *)malloc(m + 4);
            if (!*bucket || !*key || !*val) rc = FMD_E_NOMEM;
            else if (m && (hipMemcpy(*bucket, db, m * 4, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(*key, dk, m * 4, hipMemcpyDeviceToHost) != hipSuccess ||
                           hipMemcpy(*val, dv, m, hipMemcpyDeviceToHost) != hipSuccess)) rc = FMD_E_HIP;
            *n = m; cnt[0] = (int64_t)status[2]; cnt[1] = (int64_t)status[3];
        }
        hipFree(work); hipFree(db); hipFree(dk); hipFree(dv); hipFree(ds);
        if (rc != FMD_OK) return rc;
        if (status[1] == 0) return FMD_OK;
    }
    return FMD_E_OVERFLOW;
}
