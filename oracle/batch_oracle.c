/* oracle/batch_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 * Array-at-a-time wrappers over the oracle so that pytest / bench.py can check large batches
 * without a Python loop.  Same record layouts as include/fmd_hip.h so results compare bytewise.
 */
#include <stdlib.h>
#include <string.h>
#include <pthread.h>
#include "fmd_oracle.h"

void orc_rank1a_batch(const orc_rld_t *e, size_t n, const uint64_t *k, uint64_t *ok, int8_t *sym)
{
    size_t i;
    for (i = 0; i < n; ++i) {
        int c = orc_rank1a(e, k[i], ok + 6 * i);
        if (sym) sym[i] = (int8_t)c;
    }
}

void orc_rank2a_batch(const orc_rld_t *e, size_t n, const uint64_t *k, const uint64_t *l, uint64_t *ok, uint64_t *ol)
{
    size_t i;
    for (i = 0; i < n; ++i) orc_rank2a(e, k[i], l[i], ok + 6 * i, ol + 6 * i);
}

void orc_extend_batch(const orc_rld_t *e, size_t n, const orc_intv_t *ik, const uint8_t *is_back, orc_intv_t *ok)
{
    size_t i;
    for (i = 0; i < n; ++i) {
        int c;
        orc_extend(e, &ik[i], ok + 6 * i, is_back[i]);
        for (c = 0; c < 6; ++c) ok[6 * i + c].info = 0; /* fm6_extend leaves info untouched (exact.c:72-88) */
    }
}

/* retrieve: seqs is n rows of `stride` bytes (read order, zero padded), len[i], rank[i] */
void orc_retrieve_batch(const orc_rld_t *e, size_t n, const uint64_t *x, uint8_t *seqs, int stride,
                        int32_t *len, uint64_t *rank)
{
    size_t i;
    for (i = 0; i < n; ++i) {
        int l = 0;
        uint8_t *s = seqs + i * (size_t)stride;
        rank[i] = (uint64_t)orc_retrieve(e, x[i], s, stride, &l);
        len[i] = l;
        orc_reverse(l < stride ? l : stride, s);
    }
}

/* Per-id overlap records, same layout as fmd_ovlp_rec_t / fmd_intv_t of include/fmd_hip.h
 * (restated here so that the oracle does not include product headers). */
typedef struct {
    uint64_t rank, k[3];
    int32_t len, status, n_ovlp, rbeg, ext_len, n_nei;
    uint32_t flags, reserved;
} orc_ovlp_rec_t;

typedef struct { const orc_rld_t *e; size_t n; const uint64_t *ids; int min_match; uint32_t max_nei;
                 orc_ovlp_rec_t *rec; orc_intv_t *nei; uint8_t *seq; uint32_t seq_stride; int start, step, with_cls; } ovj_t;

static void *ov_worker(void *d)
{
    ovj_t *w = (ovj_t *)d;
    orc_intv_v a0 = {0, 0, 0}, a1 = {0, 0, 0}, nei = {0, 0, 0};
    orc_str_t s = {0, 0, 0};
    size_t i, j;
    s.m = 65536; s.s = (uint8_t *)malloc(s.m);
    for (i = (size_t)w->start; i < w->n; i += (size_t)w->step) {
        orc_ovlp_rec_t *r = &w->rec[i];
        orc_intv_t intv;
        int len = 0, ret;
        memset(r, 0, sizeof(*r));
        r->rbeg = -1; r->reserved = 2;
        r->rank = (uint64_t)orc_retrieve(w->e, w->ids[i], s.s, (int)s.m - 1, &len);
        orc_reverse(len, s.s);
        s.n = (size_t)len; s.s[len] = 0;
        r->len = len;
        if (w->seq) memcpy(w->seq + i * (size_t)w->seq_stride, s.s, (size_t)len < w->seq_stride ? (size_t)len : w->seq_stride);
        if (len <= w->min_match) { r->status = -1; continue; }
        a0.n = a1.n = nei.n = 0;
        ret = orc_is_contained(w->e, w->min_match, s.s, len, &intv, &a0);
        r->k[0] = intv.x[0]; r->k[1] = intv.x[1]; r->k[2] = intv.x[2];
        r->n_ovlp = (int32_t)a0.n;
        if (ret < 0) { r->status = -3; continue; }
        if (a0.n) {
            r->rbeg = orc_get_nei(w->e, w->min_match, 0, &s, &nei, &a0, &a1);
            r->ext_len = (int32_t)s.n - len;
            r->n_nei = (int32_t)nei.n;
            for (j = 0; j < nei.n && j < w->max_nei; ++j) w->nei[i * w->max_nei + j] = nei.a[j];
            if (w->seq)
                for (j = (size_t)len; j < s.n && j < w->seq_stride; ++j) w->seq[i * (size_t)w->seq_stride + j] = s.s[j];
            /* rec.reserved: check_left_simple for the edge to a unique neighbour (0 / 1), else 2 */
            r->reserved = 2;
            if (w->with_cls && nei.n == 1 && r->rbeg >= 0)
                r->reserved = orc_check_left_simple(w->e, w->min_match, 0, r->rbeg, s.s, (int)s.n) < 0 ? 1 : 0;
        }
    }
    free(a0.a); free(a1.a); free(nei.a); free(s.s);
    orc_counters_flush();
    return 0;
}

void orc_overlap_batch(const orc_rld_t *e, size_t n, const uint64_t *ids, int min_match, uint32_t max_nei,
                       void *rec, orc_intv_t *nei, uint8_t *seq, uint32_t seq_stride, int n_threads, int with_check_left)
{
    pthread_t *tid = (pthread_t *)calloc((size_t)n_threads, sizeof(pthread_t));
    ovj_t *w = (ovj_t *)calloc((size_t)n_threads, sizeof(ovj_t));
    int t;
    for (t = 0; t < n_threads; ++t) {
        ovj_t x = {e, n, ids, min_match, max_nei, (orc_ovlp_rec_t *)rec, nei, seq, seq_stride, t, n_threads, with_check_left};
        w[t] = x;
        pthread_create(&tid[t], 0, ov_worker, &w[t]);
    }
    for (t = 0; t < n_threads; ++t) pthread_join(tid[t], 0);
    free(tid); free(w);
}

/* fm6_smem (smem.c:397-410) for n fixed-length reads, start/step threads; mem[i*max_mem ..) gets the
 * first max_mem SMEMs of read i and n_mem[i] their true count -- the layout fmd_smem_dev writes. */
typedef struct { const orc_rld_t *e; size_t n; int len; const uint8_t *seqs; int self_match; uint32_t max_mem; orc_intv_t *mem; uint32_t *n_mem; int start, step; } smj_t;
static void *sm_worker(void *d)
{
    smj_t *w = (smj_t *)d;
    orc_intv_v v = {0, 0, 0};
    size_t i, j;
    for (i = (size_t)w->start; i < w->n; i += (size_t)w->step) {
        v.n = 0;
        orc_smem(w->e, w->len, w->seqs + i * (size_t)w->len, &v, w->self_match);
        w->n_mem[i] = (uint32_t)v.n;
        for (j = 0; j < v.n && j < w->max_mem; ++j) w->mem[i * w->max_mem + j] = v.a[j];
    }
    free(v.a);
    orc_counters_flush();
    return 0;
}
void orc_smem_batch(const orc_rld_t *e, size_t n, int len, const uint8_t *seqs, int self_match, uint32_t max_mem,
                    orc_intv_t *mem, uint32_t *n_mem, int n_threads)
{
    pthread_t *tid = (pthread_t *)calloc((size_t)n_threads, sizeof(pthread_t));
    smj_t *w = (smj_t *)calloc((size_t)n_threads, sizeof(smj_t));
    int t;
    for (t = 0; t < n_threads; ++t) {
        smj_t x = {e, n, len, seqs, self_match, max_mem, mem, n_mem, t, n_threads};
        w[t] = x;
        pthread_create(&tid[t], 0, sm_worker, &w[t]);
    }
    for (t = 0; t < n_threads; ++t) pthread_join(tid[t], 0);
    free(tid); free(w);
}
