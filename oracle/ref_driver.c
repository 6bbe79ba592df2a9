/* oracle/ref_driver.c -- TEST/BENCH INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * pthread drivers around the COMPILED REFERENCE (oracle/_ref/libfermi_ref.so): the reference has
 * no CLI for fm_backward_search (exact.c:7) and its unitig workers print instead of returning
 * records, so bench.py's cpu_baseline ("kind": "reference") needs a thin harness that calls the
 * reference's own exported functions with the reference's own start/step thread interleave
 * (unitig.c:394-404).  Only prototypes are declared here; no reference code is copied.
 * Built by `make -C oracle ref` into oracle/_ref/libref_driver.so (only where /root/reference exists).
 */
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

struct __rld_t;
typedef struct { uint32_t l, m; char *s; } kstring_t;                 /* fermi.h:44-48 */
typedef struct { uint64_t x[3]; uint64_t info; } fmintv_t;            /* fermi.h:13-16 */
typedef struct { size_t n, m; fmintv_t *a; } fmintv_v;                /* fermi.h:21 */
typedef struct { size_t n, m; int32_t *a; } fm32s_v;                  /* fermi.h:18 */
struct __rld_t *rld_restore(const char *fn);                           /* rld.c:288 */
void rld_destroy(struct __rld_t *e);                                   /* rld.c:81 */
uint64_t fm_backward_search(const struct __rld_t *e, int len, const uint8_t *str, uint64_t *sa_beg, uint64_t *sa_end); /* exact.c:7 */
int64_t fm_retrieve(const struct __rld_t *e, uint64_t x, kstring_t *s); /* exact.c:59 */
void seq_reverse(int l, unsigned char *s);                             /* seq.c:30 */
int fm6_is_contained(const struct __rld_t *e, int min_match, const kstring_t *s, fmintv_t *intv, fmintv_v *ovlp); /* unitig.c:77 */
int fm6_get_nei(const struct __rld_t *e, int min_match, int beg, kstring_t *s, fmintv_v *nei, fmintv_v *prev, fmintv_v *curr,
                fm32s_v *cat, uint64_t *used, const uint64_t *sorted); /* unitig.c:93 */

static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

void *refdrv_load(const char *fn) { return rld_restore(fn); }
void refdrv_free(void *e) { rld_destroy((struct __rld_t *)e); }

typedef struct { const struct __rld_t *e; size_t n; int len; const uint8_t *seqs; uint64_t *cnt, *beg, *end; int start, step; } bs_t;
static void *bs_worker(void *d)
{
    bs_t *w = (bs_t *)d;
    size_t i;
    for (i = (size_t)w->start; i < w->n; i += (size_t)w->step) {
        uint64_t b = 0, e = 0;
        w->cnt[i] = fm_backward_search(w->e, w->len, w->seqs + i * (size_t)w->len, &b, &e);
        w->beg[i] = b; w->end[i] = e;
    }
    return 0;
}
/* n fixed-length reads; returns wall seconds */
double refdrv_bsearch(void *e, size_t n, int len, const uint8_t *seqs, uint64_t *cnt, uint64_t *beg, uint64_t *end, int n_threads)
{
    pthread_t *tid = (pthread_t *)calloc((size_t)n_threads, sizeof(pthread_t));
    bs_t *w = (bs_t *)calloc((size_t)n_threads, sizeof(bs_t));
    int t;
    double t0 = now();
    for (t = 0; t < n_threads; ++t) {
        bs_t x = {(const struct __rld_t *)e, n, len, seqs, cnt, beg, end, t, n_threads};
        w[t] = x;
        pthread_create(&tid[t], 0, bs_worker, &w[t]);
    }
    for (t = 0; t < n_threads; ++t) pthread_join(tid[t], 0);
    t0 = now() - t0;
    free(tid); free(w);
    return t0;
}

/* Per-read overlap records over sequence ids ids[0..n): fm_retrieve + fm6_is_contained +
 * fm6_get_nei(used = NULL), the read-only front of unitig1 (unitig.c:274-300).  Record layout =
 * fmd_ovlp_rec_t of include/fmd_hip.h. */
typedef struct {
    uint64_t rank, k0, k1;     /* sentinel rank; x[0], x[1] of the `$read$` interval */
    int32_t len, status;       /* status: 0 ok, -1 too short, -3 contained */
    int32_t n_ovlp, rbeg, ext_len, n_nei;
    uint64_t nei[4][3];        /* first 4 neighbours: x[0], x[1], overlap length */
} ovlp_rec_t;
typedef struct { const struct __rld_t *e; size_t n; const uint64_t *ids; int min_match; ovlp_rec_t *rec; int start, step; } ov_t;
static void *ov_worker(void *d)
{
    ov_t *w = (ov_t *)d;
    kstring_t s = {0, 0, 0};
    fmintv_v a0 = {0, 0, 0}, a1 = {0, 0, 0}, nei = {0, 0, 0};
    fm32s_v cat = {0, 0, 0};
    size_t i, j;
    for (i = (size_t)w->start; i < w->n; i += (size_t)w->step) {
        ovlp_rec_t *r = &w->rec[i];
        fmintv_t intv;
        int ret;
        memset(r, 0, sizeof(*r));
        r->rbeg = -1;
        r->rank = (uint64_t)fm_retrieve(w->e, w->ids[i], &s);
        seq_reverse((int)s.l, (unsigned char *)s.s);
        r->len = (int32_t)s.l;
        if ((int)s.l <= w->min_match) { r->status = -1; continue; }
        a0.n = a1.n = nei.n = 0;
        ret = fm6_is_contained(w->e, w->min_match, &s, &intv, &a0);
        r->k0 = intv.x[0]; r->k1 = intv.x[1];
        r->n_ovlp = (int32_t)a0.n;
        if (ret < 0) { r->status = -3; continue; }
        if (a0.n) {
            int rbeg = fm6_get_nei(w->e, w->min_match, 0, &s, &nei, &a0, &a1, &cat, 0, 0);
            r->rbeg = rbeg;
            r->ext_len = (int32_t)s.l - r->len;
            r->n_nei = (int32_t)nei.n;
            for (j = 0; j < nei.n && j < 4; ++j) { r->nei[j][0] = nei.a[j].x[0]; r->nei[j][1] = nei.a[j].x[1]; r->nei[j][2] = nei.a[j].info; }
        }
    }
    free(s.s); free(a0.a); free(a1.a); free(nei.a); free(cat.a);
    return 0;
}
double refdrv_overlap(void *e, size_t n, const uint64_t *ids, int min_match, void *rec, int n_threads)
{
    pthread_t *tid = (pthread_t *)calloc((size_t)n_threads, sizeof(pthread_t));
    ov_t *w = (ov_t *)calloc((size_t)n_threads, sizeof(ov_t));
    int t;
    double t0 = now();
    for (t = 0; t < n_threads; ++t) {
        ov_t x = {(const struct __rld_t *)e, n, ids, min_match, (ovlp_rec_t *)rec, t, n_threads};
        w[t] = x;
        pthread_create(&tid[t], 0, ov_worker, &w[t]);
    }
    for (t = 0; t < n_threads; ++t) pthread_join(tid[t], 0);
    t0 = now() - t0;
    free(tid); free(w);
    return t0;
}

/* fm6_smem (smem.c:397) per read; same output layout as fmd_smem_dev. */
int fm6_smem(const struct __rld_t *e, int len, const uint8_t *q, fmintv_v *mem, int self_match); /* smem.c:397 */
typedef struct { const struct __rld_t *e; size_t n; int len; const uint8_t *seqs; int self_match; uint32_t max_mem; fmintv_t *mem; uint32_t *n_mem; int start, step; } sm_t;
static void *sm_worker(void *d)
{
    sm_t *w = (sm_t *)d;
    fmintv_v v = {0, 0, 0};
    size_t i, j;
    for (i = (size_t)w->start; i < w->n; i += (size_t)w->step) {
        v.n = 0;
        fm6_smem(w->e, w->len, w->seqs + i * (size_t)w->len, &v, w->self_match);
        w->n_mem[i] = (uint32_t)v.n;
        for (j = 0; j < v.n && j < w->max_mem; ++j) w->mem[i * w->max_mem + j] = v.a[j];
    }
    free(v.a);
    return 0;
}
double refdrv_smem(void *e, size_t n, int len, const uint8_t *seqs, int self_match, uint32_t max_mem, void *mem, uint32_t *n_mem, int n_threads)
{
    pthread_t *tid = (pthread_t *)calloc((size_t)n_threads, sizeof(pthread_t));
    sm_t *w = (sm_t *)calloc((size_t)n_threads, sizeof(sm_t));
    int t;
    double t0 = now();
    for (t = 0; t < n_threads; ++t) {
        sm_t x = {(const struct __rld_t *)e, n, len, seqs, self_match, max_mem, (fmintv_t *)mem, n_mem, t, n_threads};
        w[t] = x;
        pthread_create(&tid[t], 0, sm_worker, &w[t]);
    }
    for (t = 0; t < n_threads; ++t) pthread_join(tid[t], 0);
    t0 = now() - t0;
    free(tid); free(w);
    return t0;
}
