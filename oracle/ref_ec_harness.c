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
