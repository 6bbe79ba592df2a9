/* oracle/ref_ec_harness.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * ec_collect (correct.c:35-87) is `static` in the reference, so the only way to pin the oracle's
 * restatement of it is to compile the reference's own correct.c into a harness: this file
 * #includes it FROM /root/reference at build time (make -C oracle ref; nothing is copied) and
 * dumps the content of the per-bucket `solid` hash tables that fm6_ec_correct's first phase
 * builds (correct.c:341-356).  Output: (bucket, key, val) triples in bucket order.
 */
#include "correct.c"

int refec_collect(const char *fn, int w, int min_occ, int suf_len, uint32_t **o_bucket, uint32_t **o_key, uint8_t **o_val,
                  uint64_t *o_n, int64_t cnt[2])
{
    rld_t *e = rld_restore(fn);
    fmecopt_t opt;
    fmintv_t *top;
    uint64_t n = 0, m = 1024;
    uint32_t *B = malloc(m * 4), *K = malloc(m * 4);
    uint8_t *V = malloc(m);
    int b;
    if (e == 0) return -1;
    memset(&opt, 0, sizeof(opt));
    opt.w = w; opt.min_occ = min_occ;
    compute_SUF(suf_len);
    top = fm6_traverse(e, SUF_LEN);
    cnt[0] = cnt[1] = 0;
    for (b = 0; b < SUF_NUM; ++b) {
        shash_t *h = kh_init(solid);
        khint_t k;
        ec_collect(e, &opt, SUF_LEN, &top[b], h, cnt);
        for (k = kh_begin(h); k != kh_end(h); ++k) {
            if (!kh_exist(h, k)) continue;
            if (n == m) { m <<= 1; B = realloc(B, m * 4); K = realloc(K, m * 4); V = realloc(V, m); }
            B[n] = (uint32_t)b; K[n] = kh_key(h, k); V[n] = kh_val(h, k); ++n;
        }
        kh_destroy(solid, h);
    }
    free(top);
    rld_destroy(e);
    *o_bucket = B; *o_key = K; *o_val = V; *o_n = n;
    return 0;
}

/* Bench/parity driver: ec_collect over the suffix buckets [b0, b1) with the reference's own
 * start/step thread interleave (worker1, correct.c:272-279).  Returns the (bucket, key, val) triples
 * of those buckets (order unspecified) and the wall seconds of the collect phase alone. */
#include <time.h>
typedef struct { const rld_t *e; const fmecopt_t *opt; const fmintv_t *top; int b0, b1, start, step;
                 uint32_t *B, *K; uint8_t *V; uint64_t n, m; int64_t cnt[2]; } rr_t;
static void *rr_worker(void *d)
{
    rr_t *w = (rr_t *)d;
    int b;
    for (b = w->b0 + w->start; b < w->b1; b += w->step) {
        shash_t *h = kh_init(solid);
        khint_t k;
        ec_collect(w->e, w->opt, SUF_LEN, &w->top[b], h, w->cnt);
        for (k = kh_begin(h); k != kh_end(h); ++k) {
            if (!kh_exist(h, k)) continue;
            if (w->n == w->m) {
                w->m = w->m ? w->m << 1 : 1024;
                w->B = realloc(w->B, w->m * 4); w->K = realloc(w->K, w->m * 4); w->V = realloc(w->V, w->m);
            }
            w->B[w->n] = (uint32_t)b; w->K[w->n] = kh_key(h, k); w->V[w->n] = kh_val(h, k); ++w->n;
        }
        kh_destroy(solid, h);
    }
    return 0;
}
static double rr_now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

int refec_range(const char *fn, int w, int min_occ, int suf_len, int b0, int b1, int n_threads, uint32_t **o_bucket,
                uint32_t **o_key, uint8_t **o_val, uint64_t *o_n, double *secs)
{
    rld_t *e = rld_restore(fn);
    fmecopt_t opt;
    fmintv_t *top;
    pthread_t *tid;
    rr_t *ws;
    uint64_t n = 0, off = 0;
    int t;
    double t0;
    if (e == 0) return -1;
    memset(&opt, 0, sizeof(opt));
    opt.w = w; opt.min_occ = min_occ;
    compute_SUF(suf_len);
    if (b1 > SUF_NUM) b1 = SUF_NUM;
    top = fm6_traverse(e, SUF_LEN);
    tid = calloc(n_threads, sizeof(pthread_t));
    ws = calloc(n_threads, sizeof(rr_t));
    t0 = rr_now();
    for (t = 0; t < n_threads; ++t) {
        ws[t].e = e; ws[t].opt = &opt; ws[t].top = top; ws[t].b0 = b0; ws[t].b1 = b1; ws[t].start = t; ws[t].step = n_threads;
        pthread_create(&tid[t], 0, rr_worker, &ws[t]);
    }
    for (t = 0; t < n_threads; ++t) pthread_join(tid[t], 0);
    *secs = rr_now() - t0;
    for (t = 0; t < n_threads; ++t) n += ws[t].n;
    *o_bucket = malloc(n * 4 + 4); *o_key = malloc(n * 4 + 4); *o_val = malloc(n + 4);
    for (t = 0; t < n_threads; ++t) {
        memcpy(*o_bucket + off, ws[t].B, ws[t].n * 4); memcpy(*o_key + off, ws[t].K, ws[t].n * 4); memcpy(*o_val + off, ws[t].V, ws[t].n);
        off += ws[t].n;
        free(ws[t].B); free(ws[t].K); free(ws[t].V);
    }
    *o_n = n;
    free(tid); free(ws); free(top);
    rld_destroy(e);
    return 0;
}
void refec_free(void *p) { free(p); }

/* Bench/parity driver for the correction pass: the reference's own ec_fix (correct.c:222-256, static) over n reads of len bases each, on n_threads
 * threads with the reference's interleave (read k -> thread k % n_threads, correct.c:444-446).  The `solid` tables are filled from (bucket, key, val)
 * triples SORTED BY BUCKET (what ec_collect would have put there, correct.c:71-75: the product's harvest, itself checked against ec_collect on a bucket
 * sample) -- a look-up depends on the tables' content only, not on the insertion order.  seqs: ASCII bases, quals: phred + 33, both n x len, rewritten in
 * place as ec_fix leaves them (corrected bases in lower case, their quality 36); info[n].  *secs = wall seconds of the ec_fix threads alone, *n_query =
 * the table look-ups they made. */
typedef struct { const fmecopt_t *opt; shash_t *const *solid; const uint32_t *B, *K; const uint8_t *V; uint64_t t0, t1; } rf_fill_t;
static void *rf_fill(void *d)
{
    rf_fill_t *f = (rf_fill_t *)d;
    uint64_t i;
    for (i = f->t0; i < f->t1; ++i) {
        int absent;
        shash_t *h = (shash_t *)f->solid[f->B[i]];
        khint_t k = kh_put(solid, h, f->K[i], &absent);
        kh_val(h, k) = f->V[i];
    }
    return 0;
}
int refec_fix(int w, int suf_len, int step, double max_corr, uint64_t n_trip, const uint32_t *B, const uint32_t *K, const uint8_t *V,
              int n, int len, char *seqs, char *quals, int *info, int n_threads, double *secs, uint64_t *n_query)
{
    fmecopt_t opt;
    shash_t **solid;
    pthread_t *tid;
    worker2_t *w2;
    rf_fill_t *ff;
    int j, t;
    uint64_t i;
    double t0;
    memset(&opt, 0, sizeof(opt));
    opt.w = w; opt.min_occ = 3; opt.max_corr = (float)max_corr; opt.step = step;
    compute_SUF(suf_len);
    for (i = 1; i < n_trip; ++i) if (B[i] < B[i - 1]) return -2;                 /* not sorted by bucket */
    if (n_trip && B[n_trip - 1] >= (uint32_t)SUF_NUM) return -3;
    solid = calloc(SUF_NUM, sizeof(void *));
    for (j = 0; j < SUF_NUM; ++j) solid[j] = kh_init(solid);
    tid = calloc(n_threads, sizeof(pthread_t));
    ff = calloc(n_threads, sizeof(rf_fill_t));
    for (t = 0; t < n_threads; ++t) {                                             /* the fill, by ranges of whole buckets */
        uint64_t a = n_trip * (uint64_t)t / n_threads, b = n_trip * (uint64_t)(t + 1) / n_threads;
        while (a > 0 && a < n_trip && B[a] == B[a - 1]) ++a;
        while (b > 0 && b < n_trip && B[b] == B[b - 1]) ++b;
        ff[t].opt = &opt; ff[t].solid = solid; ff[t].B = B; ff[t].K = K; ff[t].V = V; ff[t].t0 = a; ff[t].t1 = b;
        pthread_create(&tid[t], 0, rf_fill, &ff[t]);
    }
    for (t = 0; t < n_threads; ++t) pthread_join(tid[t], 0);
    free(ff);
    w2 = calloc(n_threads, sizeof(worker2_t));
    for (t = 0; t < n_threads; ++t) {
        int m = (n + n_threads - 1) / n_threads;
        w2[t].e = 0; w2[t].solid = solid; w2[t].opt = &opt;
        w2[t].seq = calloc(m, sizeof(void *)); w2[t].qual = calloc(m, sizeof(void *)); w2[t].info = calloc(m, sizeof(int));
    }
    {   /* ec_fix works on NUL-terminated strings: private copies, as the reference's batches are (correct.c:446-452) */
        char *cs = malloc((size_t)n * (len + 1)), *cq = malloc((size_t)n * (len + 1));
        for (j = 0; j < n; ++j) {
            worker2_t *x = &w2[j % n_threads];
            memcpy(cs + (size_t)j * (len + 1), seqs + (size_t)j * len, len); cs[(size_t)j * (len + 1) + len] = 0;
            memcpy(cq + (size_t)j * (len + 1), quals + (size_t)j * len, len); cq[(size_t)j * (len + 1) + len] = 0;
            x->seq[x->n_seqs] = cs + (size_t)j * (len + 1); x->qual[x->n_seqs] = cq + (size_t)j * (len + 1); ++x->n_seqs;
        }
        t0 = rr_now();
        for (t = 0; t < n_threads; ++t) pthread_create(&tid[t], 0, worker2, &w2[t]);
        for (t = 0; t < n_threads; ++t) pthread_join(tid[t], 0);
        *secs = rr_now() - t0;
        *n_query = 0;
        for (t = 0; t < n_threads; ++t) { *n_query += w2[t].n_query; w2[t].n_seqs = 0; }
        for (j = 0; j < n; ++j) {
            worker2_t *x = &w2[j % n_threads];
            memcpy(seqs + (size_t)j * len, cs + (size_t)j * (len + 1), len);
            memcpy(quals + (size_t)j * len, cq + (size_t)j * (len + 1), len);
            info[j] = x->info[x->n_seqs++];
        }
        free(cs); free(cq);
    }
    for (t = 0; t < n_threads; ++t) { free(w2[t].seq); free(w2[t].qual); free(w2[t].info); }
    free(w2); free(tid);
    for (j = 0; j < SUF_NUM; ++j) kh_destroy(solid, solid[j]);
    free(solid);
    return 0;
}
