/* oracle/ref_driver.c -- TEST/BENCH INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * pthread drivers around the COMPILED REFERENCE (oracle/_ref/libfermi_ref.so): the reference has
 * no CLI for fm_backward_search (exact.c:7) and its unitig workers print instead of returning
 * records, so bench.py's cpu_baseline ("kind": "reference") needs a thin harness that calls the
 * reference's own exported functions from a pool of pinned threads (drv_run below; the reference's own
 * start/step interleave, unitig.c:394-404, false-shares every output line between 8 workers).
 * Only prototypes are declared here; no reference code is copied.
 * Built by `make -C oracle ref` into oracle/_ref/libref_driver.so (only where /root/reference exists).
 */
#define _GNU_SOURCE
#include <pthread.h>
#include <sched.h>
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


/* Work distribution shared by the three drivers: the reference hands item i to worker i % n_threads
 * (unitig.c:333, 398-399), which makes 8 workers write into every output cache line; here workers take
 * CHUNKS of consecutive items from one atomic counter (no false sharing, no tail imbalance) and are pinned to
 * the CPUs this process may use, so that the baseline is the reference's functions at their best. */
#define DRV_CHUNK 256
typedef struct { void (*item)(void *ctx, void *scratch, size_t i); void *(*mk)(void); void (*rm)(void *); void *ctx; size_t n; size_t *next; int cpu; } drv_t;
static void *drv_worker(void *d)
{
    drv_t *w = (drv_t *)d;
    if (w->cpu >= 0) { cpu_set_t m; CPU_ZERO(&m); CPU_SET(w->cpu, &m); pthread_setaffinity_np(pthread_self(), sizeof(m), &m); }
    void *scratch = w->mk ? w->mk() : 0;
    for (;;) {
        size_t b = __atomic_fetch_add(w->next, (size_t)DRV_CHUNK, __ATOMIC_RELAXED), e = b + DRV_CHUNK, i;
        if (b >= w->n) break;
        if (e > w->n) e = w->n;
        for (i = b; i < e; ++i) w->item(w->ctx, scratch, i);
    }
    if (w->rm) w->rm(scratch);
    return 0;
}
/* CPUs this process is allowed to run on (affinity mask); what "all cores" means for the baseline */
int refdrv_usable_cpus(void)
{
    cpu_set_t m;
    if (sched_getaffinity(0, sizeof(m), &m) != 0) return 1;
    return CPU_COUNT(&m);
}
static double drv_run(void (*item)(void *, void *, size_t), void *(*mk)(void), void (*rm)(void *), void *ctx, size_t n, int n_threads)
{
    pthread_t *tid = (pthread_t *)calloc((size_t)n_threads, sizeof(pthread_t));
    drv_t *w = (drv_t *)calloc((size_t)n_threads, sizeof(drv_t));
    char *started = (char *)calloc((size_t)n_threads, 1);
    cpu_set_t m;
    int cpus[CPU_SETSIZE], n_cpu = 0, t, c;
    size_t next = 0;
    double t0;
    if (sched_getaffinity(0, sizeof(m), &m) == 0) for (c = 0; c < CPU_SETSIZE; ++c) if (CPU_ISSET(c, &m)) cpus[n_cpu++] = c;
    t0 = now();
    for (t = 0; t < n_threads; ++t) {
        drv_t x = {item, mk, rm, ctx, n, &next, n_cpu ? cpus[t % n_cpu] : -1};
        w[t] = x;
        started[t] = pthread_create(&tid[t], 0, drv_worker, &w[t]) == 0;
    }
    if (!started[0]) drv_worker(&w[0]);   /* could not create any thread: run inline */
    for (t = 0; t < n_threads; ++t) if (started[t]) pthread_join(tid[t], 0);
    t0 = now() - t0;
    free(tid); free(w); free(started);
    return t0;
}

void *refdrv_load(const char *fn) { return rld_restore(fn); }
void refdrv_free(void *e) { rld_destroy((struct __rld_t *)e); }

typedef struct { const struct __rld_t *e; int len; const uint8_t *seqs; uint64_t *cnt, *beg, *end; } bs_t;
static void bs_item(void *d, void *scratch, size_t i)
{
    bs_t *w = (bs_t *)d;
    uint64_t b = 0, e = 0;
    (void)scratch;
    w->cnt[i] = fm_backward_search(w->e, w->len, w->seqs + i * (size_t)w->len, &b, &e);
    w->beg[i] = b; w->end[i] = e;
}
/* n fixed-length reads; returns wall seconds */
double refdrv_bsearch(void *e, size_t n, int len, const uint8_t *seqs, uint64_t *cnt, uint64_t *beg, uint64_t *end, int n_threads)
{
    bs_t x = {(const struct __rld_t *)e, len, seqs, cnt, beg, end};
    return drv_run(bs_item, 0, 0, &x, n, n_threads);
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
typedef struct { const struct __rld_t *e; const uint64_t *ids; int min_match; ovlp_rec_t *rec; } ov_t;
typedef struct { kstring_t s; fmintv_v a0, a1, nei; fm32s_v cat; } ov_scratch_t;
static void *ov_mk(void) { return calloc(1, sizeof(ov_scratch_t)); }
static void ov_rm(void *p) { ov_scratch_t *z = (ov_scratch_t *)p; free(z->s.s); free(z->a0.a); free(z->a1.a); free(z->nei.a); free(z->cat.a); free(z); }
static void ov_item(void *d, void *scratch, size_t i)
{
    ov_t *w = (ov_t *)d;
    ov_scratch_t *z = (ov_scratch_t *)scratch;
    ovlp_rec_t *r = &w->rec[i];
    fmintv_t intv;
    size_t j;
    int ret;
    memset(r, 0, sizeof(*r));
    r->rbeg = -1;
    r->rank = (uint64_t)fm_retrieve(w->e, w->ids[i], &z->s);
    seq_reverse((int)z->s.l, (unsigned char *)z->s.s);
    r->len = (int32_t)z->s.l;
    if ((int)z->s.l <= w->min_match) { r->status = -1; return; }
    z->a0.n = z->a1.n = z->nei.n = 0;
    ret = fm6_is_contained(w->e, w->min_match, &z->s, &intv, &z->a0);
    r->k0 = intv.x[0]; r->k1 = intv.x[1];
    r->n_ovlp = (int32_t)z->a0.n;
    if (ret < 0) { r->status = -3; return; }
    if (z->a0.n) {
        int rbeg = fm6_get_nei(w->e, w->min_match, 0, &z->s, &z->nei, &z->a0, &z->a1, &z->cat, 0, 0);
        r->rbeg = rbeg;
        r->ext_len = (int32_t)z->s.l - r->len;
        r->n_nei = (int32_t)z->nei.n;
        for (j = 0; j < z->nei.n && j < 4; ++j) { r->nei[j][0] = z->nei.a[j].x[0]; r->nei[j][1] = z->nei.a[j].x[1]; r->nei[j][2] = z->nei.a[j].info; }
    }
}
double refdrv_overlap(void *e, size_t n, const uint64_t *ids, int min_match, void *rec, int n_threads)
{
    ov_t x = {(const struct __rld_t *)e, ids, min_match, (ovlp_rec_t *)rec};
    return drv_run(ov_item, ov_mk, ov_rm, &x, n, n_threads);
}

/* fm6_smem (smem.c:397) per read; same output layout as fmd_smem_dev. */
int fm6_smem(const struct __rld_t *e, int len, const uint8_t *q, fmintv_v *mem, int self_match); /* smem.c:397 */
typedef struct { const struct __rld_t *e; int len; const uint8_t *seqs; int self_match; uint32_t max_mem; fmintv_t *mem; uint32_t *n_mem; } sm_t;
static void *sm_mk(void) { return calloc(1, sizeof(fmintv_v)); }
static void sm_rm(void *p) { free(((fmintv_v *)p)->a); free(p); }
static void sm_item(void *d, void *scratch, size_t i)
{
    sm_t *w = (sm_t *)d;
    fmintv_v *v = (fmintv_v *)scratch;
    size_t j;
    v->n = 0;
    fm6_smem(w->e, w->len, w->seqs + i * (size_t)w->len, v, w->self_match);
    w->n_mem[i] = (uint32_t)v->n;
    for (j = 0; j < v->n && j < w->max_mem; ++j) w->mem[i * w->max_mem + j] = v->a[j];
}
double refdrv_smem(void *e, size_t n, int len, const uint8_t *seqs, int self_match, uint32_t max_mem, void *mem, uint32_t *n_mem, int n_threads)
{
    sm_t x = {(const struct __rld_t *)e, len, seqs, self_match, max_mem, (fmintv_t *)mem, n_mem};
    return drv_run(sm_item, sm_mk, sm_rm, &x, n, n_threads);
}
