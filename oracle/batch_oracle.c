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
    uint32_t flags;
    uint16_t reserved, lfork;   /* lfork: EXACT here (orc_left_fork); the product may know less (see include/fmd_hip.h) */
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
        if (w->with_cls) r->lfork = (uint16_t)orc_left_fork(w->e, w->min_match, s.s, len);   /* with the other check_left field: not part of the reference's per-read work that bench.py counts and times */
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

/* ec_collect (correct.c:35-87) over suffix buckets [b0, b1), start/step threads (worker1,
 * correct.c:272-279); triples (bucket, key, val) concatenated in unspecified order. */
#include <time.h>
typedef struct { const orc_rld_t *e; int w, min_occ, suf_len; const orc_intv_t *top; int b0, b1, start, step;
                 uint32_t *B, *K; uint8_t *V; uint64_t n, m; } ecj_t;
static void *ec_worker(void *d)
{
    ecj_t *w = (ecj_t *)d;
    int b;
    for (b = w->b0 + w->start; b < w->b1; b += w->step) {
        orc_solid_t so;
        size_t i;
        memset(&so, 0, sizeof(so));
        orc_ec_collect(w->e, w->w, w->min_occ, w->suf_len, &w->top[b], &so);
        for (i = 0; i < so.n; ++i) {
            if (w->n == w->m) {
                w->m = w->m ? w->m << 1 : 1024;
                w->B = (uint32_t *)realloc(w->B, w->m * 4); w->K = (uint32_t *)realloc(w->K, w->m * 4); w->V = (uint8_t *)realloc(w->V, w->m);
            }
            w->B[w->n] = (uint32_t)b; w->K[w->n] = so.key[i]; w->V[w->n] = so.val[i]; ++w->n;
        }
        free(so.key); free(so.val);
    }
    orc_counters_flush();
    return 0;
}
int orc_ec_range(const orc_rld_t *e, int w, int min_occ, int suf_len, int b0, int b1, int n_threads, uint32_t **o_bucket,
                 uint32_t **o_key, uint8_t **o_val, uint64_t *o_n, double *secs)
{
    orc_intv_t *top = orc_traverse(e, suf_len);
    pthread_t *tid = (pthread_t *)calloc((size_t)n_threads, sizeof(pthread_t));
    ecj_t *ws = (ecj_t *)calloc((size_t)n_threads, sizeof(ecj_t));
    uint64_t n = 0, off = 0;
    struct timespec t0, t1;
    int t;
    if (b1 > (1 << (2 * suf_len))) b1 = 1 << (2 * suf_len);
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (t = 0; t < n_threads; ++t) {
        ws[t].e = e; ws[t].w = w; ws[t].min_occ = min_occ; ws[t].suf_len = suf_len; ws[t].top = top;
        ws[t].b0 = b0; ws[t].b1 = b1; ws[t].start = t; ws[t].step = n_threads;
        pthread_create(&tid[t], 0, ec_worker, &ws[t]);
    }
    for (t = 0; t < n_threads; ++t) pthread_join(tid[t], 0);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    *secs = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
    for (t = 0; t < n_threads; ++t) n += ws[t].n;
    *o_bucket = (uint32_t *)malloc(n * 4 + 4); *o_key = (uint32_t *)malloc(n * 4 + 4); *o_val = (uint8_t *)malloc(n + 4);
    for (t = 0; t < n_threads; ++t) {
        memcpy(*o_bucket + off, ws[t].B, ws[t].n * 4); memcpy(*o_key + off, ws[t].K, ws[t].n * 4); memcpy(*o_val + off, ws[t].V, ws[t].n);
        off += ws[t].n;
        free(ws[t].B); free(ws[t].K); free(ws[t].V);
    }
    *o_n = n;
    free(tid); free(ws); free(top);
    return 0;
}
