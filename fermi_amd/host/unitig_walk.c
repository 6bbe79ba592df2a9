/* unitig_walk.c -- the deterministic `fermi unitig -t1` walk, replayed on the host over the per-read
 * overlap table the GPU computed (include/fmd_hip.h: fmd_ovlp_*).
 *
 * In the reference, unitig_core (unitig.c:319-362) seeds from every odd sequence id, and
 * unitig1 / unitig_unidir (unitig.c:227-317) extend a seed through unique irreducible overlaps.
 * Everything in that walk that touches the FM-index is a pure function of one read-strand:
 *   fm_retrieve + fm6_is_contained            -> rec.rank, rec.len, rec.k[], rec.status, seq
 *   fm6_get_nei on the terminal read          -> rec.rbeg, nei[], appended bases (seq tail)
 *   check_left_simple on the edge to the unique neighbour -> rec.reserved
 *   check_left's second look (get_nei on the reverse strand of the neighbour) -> that strand's n_nei
 * so the walk itself is bookkeeping: the used / bend / visited bitmaps (unitig.c:15-36, 238-253,
 * 337-339), coverage strings and the MAG record printer (mag.c:149-174).  With one thread the
 * reference is deterministic and this reproduces its output byte for byte.
 * (Marks that the reference puts on CONTAINED reads -- unitig.c:124, :197 -- are not replayed: a
 * contained seed is rejected with or without them, unitig.c:282-292.)
 */
#define _GNU_SOURCE
#include <errno.h>
#include <pthread.h>
#include <stdio.h>
#include <fcntl.h>
#include <sched.h>
#include <sys/stat.h>
#include <unistd.h>
#include <time.h>
#include <stdlib.h>
#include <string.h>
#include "fmd_host.h"

typedef struct { char *s; size_t l, m; } str_t;
static double wall_s(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec; }

static int str_reserve(str_t *s, size_t need)
{
    if (need <= s->m) return 0;
    size_t m = s->m ? s->m : 256;
    while (m < need) m <<= 1;
    char *p = (char *)realloc(s->s, m);
    if (!p) return -ENOMEM;
    s->s = p; s->m = m;
    return 0;
}

/* (the bitmaps are read by the speculating threads while the committing thread writes them: word-sized relaxed atomics; ONE thread writes) */
static inline int bit_get(const uint64_t *b, uint64_t x) { return (int)(__atomic_load_n(&b[x >> 6], __ATOMIC_RELAXED) >> (x & 63) & 1); }
static inline void bit_set(uint64_t *b, uint64_t x) { __atomic_store_n(&b[x >> 6], b[x >> 6] | 1ull << (x & 63), __ATOMIC_RELAXED); }

/* The three bitmaps of the walk (unitig.c:390-392) are its only state that outlives a seed.  A walk run SPECULATIVELY (the parallel
 * driver at the end of this file) reads them through st_get and writes through st_set: reads that find a bit set in the shared maps
 * are final (bits are never cleared); every other read is logged with the value it saw, writes go to a private overlay and a log, and
 * the shared maps are only touched when the walk is committed in seed order. */
enum { ST_USED = 0, ST_BEND = 1, ST_VIS = 2 };
typedef struct {
    uint64_t *keys; uint32_t cap, n;   /* overlay: open addressing, key + 1 stored (0 = empty) */
    uint64_t *rlog; size_t n_r, m_r;   /* which << 62 | value << 61 | bit */
    uint64_t *wlog; size_t n_w, m_w;   /* which << 62 | bit */
    int err;
    uint32_t budget;                   /* reads the walk in progress may still take in (SPEC_MAX_STEPS per seed) */
} spec_t;

typedef struct {
    const fmdh_slim_t *t;       /* the table: 32 bytes per row the walk steps through, a short variable part for seeds and ends (slim_table.c) */
    uint64_t n_seq;
    int min_match;
    uint64_t *used, *bend, *visited;
    int full_records;           /* records written whole (mag_g_print, mag.c:176-188: fwrite) instead of cut at a NUL (unitig.c:354: fputs) */
    spec_t *sp;                 /* NULL: the maps are read and written directly (the sequential walk, and re-runs at commit) */
    const uint64_t *sorted;     /* optional rank -> (sequence id << 2 | flags) map of `unitig -r` (unitig.c:22-29) */
    /* the neighbour list left behind by the last try_right (unitig.c:181-184): that of row `last` */
    uint64_t last; int n_nei;
    int err;
    int have_far;               /* w[].far is built: the row JUMP_DIST accepted links ahead of each row (prefetch hints only) */
    int no_plain;               /* FMD_WALK_NO_HOP: every step through the general code (the A/B switch, and the tests' second opinion) */
    int seed_hints;             /* the seeds' first hops are prefetched (unless FMD_WALK_NO_JUMP) */
    int timing;                 /* FMD_TIMING: the clock is read around the sections of a walk (four times per seed that walks: not for free at 4*10^7 seeds) */
    double t_uni, t_turn, t_text; uint64_t n_hops;   /* FMD_TIMING: seconds inside unidir, turning the string round, formatting the record; reads appended */
} walk_t;

/* The walk is a pointer chase: the next row is known when w[row] has arrived, one DRAM miss (~90 ns) per read and nothing
 * to overlap it with -- 1.8 s per 10^7 reads.  A chase cannot be prefetched, a chase with a skip list can: w[row].far = the
 * row 2^JUMP_LOG links further on (build_far, all host threads), and visiting a row prefetches the line the
 * visit JUMP_DIST steps later will read; half way there, when that line has arrived, the bitmap words its step will test and set.
 * Hints only: where the walk stops or turns, a prefetch was wasted, nothing else.  The PLAIN step -- one neighbour, check_left decided,
 * the appended bases in the line: all but the last step of every walk -- reads w[row] and the neighbour's w[] (its `$read$` interval),
 * which is the line the next step starts from: one new line per read (rounds 2-4 read seven lines from seven arrays, then one of a
 * 32-byte hop[] built beside the table; the table IS that array now). */
#define JUMP_LOG 3
#define JUMP_DIST (1 << JUMP_LOG)
#define SEED_AHEAD 32           /* ids between a seed and the one whose first hop is prefetched */
typedef struct { fmdh_slim_t *s; uint32_t *tmp; int lv; } far_job_t;
static void far_main(void *p, int tid, int nt)
{
    far_job_t *j = (far_job_t *)p;
    fmdh_wrec_t *W = j->s->w;
    const uint64_t n = j->s->n, a = n * (uint64_t)tid / (uint64_t)nt, b = n * (uint64_t)(tid + 1) / (uint64_t)nt;
    uint64_t i;
    if (j->lv == 1) for (i = a; i < b; ++i) { const uint32_t x = W[i].nxt; j->tmp[i] = x != 0xffffffffu ? W[x].nxt : x; }                    /* two links */
    else if (j->lv == 2) for (i = a; i < b; ++i) { const uint32_t x = j->tmp[i]; W[i].far = x != 0xffffffffu ? j->tmp[x] : x; }              /* four */
    else if (j->lv == 3) for (i = a; i < b; ++i) { const uint32_t x = W[i].far; j->tmp[i] = x != 0xffffffffu ? W[x].far : x; }               /* eight */
    else for (i = a; i < b; ++i) W[i].far = j->tmp[i];
}
/* The same w[].far without the temporary array (4 bytes per id: 5.6 GB at BASELINE's 1.4*10^9 ids, on top of a table of 62 GB): every row follows its
 * own eight links -- w[].nxt only, which nothing writes -- FAR_LANES rows at a time so that their misses overlap.  Eight random lines per row instead of
 * the doubling's three, and no slower for it (2*10^7 rows on 16 threads: 0.09 s; the doubling 0.06-0.12 s): the default.  FMD_FAR_CHASE=0 = the doubling;
 * FMD_FAR_CHECK=1 = both, compared (tests). */
#define FAR_LANES 32
static void far_chase_main(void *p, int tid, int nt)
{
    fmdh_wrec_t *W = ((far_job_t *)p)->s->w;
    const uint64_t n = ((far_job_t *)p)->s->n, a = n * (uint64_t)tid / (uint64_t)nt, b = n * (uint64_t)(tid + 1) / (uint64_t)nt;
    uint64_t i;
    for (i = a; i < b; i += FAR_LANES) {
        const int m = b - i < FAR_LANES ? (int)(b - i) : FAR_LANES;
        uint32_t cur[FAR_LANES];
        int k, h;
        for (k = 0; k < m; ++k) { cur[k] = W[i + k].nxt; if (cur[k] != 0xffffffffu) __builtin_prefetch(&W[cur[k]]); }
        for (h = 1; h < JUMP_DIST; ++h)
            for (k = 0; k < m; ++k) if (cur[k] != 0xffffffffu) { cur[k] = W[cur[k]].nxt; if (cur[k] != 0xffffffffu) __builtin_prefetch(&W[cur[k]]); }
        for (k = 0; k < m; ++k) W[i + k].far = cur[k];
    }
}
static int build_far(fmdh_slim_t *s)
{
    far_job_t j = {s, 0, 0};
    const char *e = getenv("FMD_FAR_CHASE");
    if ((!e || atoi(e) != 0) && !getenv("FMD_FAR_CHECK")) { fmdh_par_for(fmdh_host_threads(), far_chase_main, &j); return 1; }
    j.tmp = (uint32_t *)fmdh_big_alloc((s->n ? s->n : 1) * 4);
    if (!j.tmp) { fmdh_par_for(fmdh_host_threads(), far_chase_main, &j); return 1; }    /* (no room for the temporary: the slower way) */
    for (j.lv = 1; j.lv <= 4; ++j.lv) fmdh_par_for(fmdh_host_threads(), far_main, &j);
    if (getenv("FMD_FAR_CHECK")) {   /* tests: the other construction gives the same array */
        uint64_t i, bad = 0;
        for (i = 0; i < s->n; ++i) j.tmp[i] = s->w[i].far;
        fmdh_par_for(fmdh_host_threads(), far_chase_main, &j);
        for (i = 0; i < s->n; ++i) bad += j.tmp[i] != s->w[i].far;
        if (bad) { fprintf(stderr, "[E::%s] the two constructions of the skip list differ in %llu rows\n", __func__, (unsigned long long)bad); fmdh_big_free(j.tmp); return -1; }
    }
    fmdh_big_free(j.tmp);
    return 1;
}
static inline void prefetch_var(const fmdh_slim_t *t, uint32_t row) { const uint8_t *p = fmdh_slim_var(t, row); __builtin_prefetch(p); __builtin_prefetch(p + 64); }
static inline uint64_t row_rank(const fmdh_slim_t *t, uint64_t row)
{
    const fmdh_wrec_t *h = &t->w[row];
    if (h->bits & FMDH_W_BIG) { fmd_ovlp_rec_t r; memcpy(&r, fmdh_slim_var(t, row), 64); return r.rank; }
    return (uint64_t)h->k0 + h->dr;
}

/* Short unitigs (reads with errors: 10^8 of them at 50 M reads) leave the skip list nothing to look ahead along; what a seed will
 * touch is known from the seeds' own lines, which are read in id order.  Three stages, SEED_AHEAD ids apart, for the two first hops
 * of a seed to come (from the seed's strand and from its reverse): the hop's line -- then its variable part and the lines of the SECOND
 * hops -- then their variable parts. */
/* A seed that is `used` already returns at once (unitig.c:282, :289) and nothing of its rows is read: no hints for it.  On error-free reads that is
 * every seed but a handful -- the genome is one walk -- and thirty wasted line fetches per seed were most of the walk's time there (2*10^6 reads on
 * 8 cores: 1.26 s, of which 0.19 s inside walks).  The bit is read as it is NOW, without the speculative walk's logs: a hint decides nothing. */
static inline int seed_is_used(const walk_t *w, uint64_t j)
{
    const unsigned st = w->t->w[j].bits & FMDH_W_ST_MASK;
    if (st == FMDH_W_ST_SHORT || st == FMDH_W_ST_INVALID) return 1;      /* (returns at once as well, and its line holds no rank) */
    return w->sorted ? bit_get(w->used, j) : bit_get(w->used, row_rank(w->t, j));
}
static inline void seed_hints(const walk_t *w, uint64_t i, int staged)
{
    const fmdh_slim_t *t = w->t;
    const fmdh_wrec_t *W = t->w;
    int d;
    if (!staged) {   /* (FMD_WALK_SEED_STAGES=0: the first hop's line only) */
        if (i + SEED_AHEAD < t->n) for (d = 0; d < 2; ++d) { const uint32_t a = W[i + SEED_AHEAD - (uint64_t)d].nxt; if (a != 0xffffffffu) __builtin_prefetch(&W[a]); }
        return;
    }
    if (i + 3 * SEED_AHEAD < t->n && !seed_is_used(w, i + 3 * SEED_AHEAD))
        for (d = 0; d < 2; ++d) { const uint32_t a = W[i + 3 * SEED_AHEAD - (uint64_t)d].nxt; if (a != 0xffffffffu) __builtin_prefetch(&W[a]); }
    if (i + 2 * SEED_AHEAD < t->n && !seed_is_used(w, i + 2 * SEED_AHEAD))
        for (d = 0; d < 2; ++d) {
            const uint32_t a = W[i + 2 * SEED_AHEAD - (uint64_t)d].nxt;
            if (a == 0xffffffffu) continue;
            prefetch_var(t, a);
            if (W[a].nxt != 0xffffffffu) __builtin_prefetch(&W[W[a].nxt]);
        }
    if (i + SEED_AHEAD < t->n && !seed_is_used(w, i + SEED_AHEAD))
        for (d = 0; d < 2; ++d) {
            const uint32_t a = W[i + SEED_AHEAD - (uint64_t)d].nxt;
            if (a != 0xffffffffu && W[a].nxt != 0xffffffffu) prefetch_var(t, W[a].nxt);
        }
}

/* ---- output: records are formatted by the walk and written by a second thread (two buffers handed back and forth) */
#define OUT_BUF ((size_t)8 << 20)
typedef struct {
    FILE *fp;
    char *buf[2]; size_t len[2];
    int fill;                    /* buffer the walk appends to */
    int pending;                 /* buffer waiting to be written, or -1 */
    int quit, err, threaded;
    pthread_t tid; pthread_mutex_t mu; pthread_cond_t cv;
} outq_t;
static void *outq_main(void *p)
{
    outq_t *q = (outq_t *)p;
    pthread_mutex_lock(&q->mu);
    for (;;) {
        while (q->pending < 0 && !q->quit) pthread_cond_wait(&q->cv, &q->mu);
        if (q->pending < 0) break;
        {
            const int k = q->pending;
            pthread_mutex_unlock(&q->mu);
            if (fwrite(q->buf[k], 1, q->len[k], q->fp) != q->len[k]) q->err = 1;
            pthread_mutex_lock(&q->mu);
            q->len[k] = 0; q->pending = -1;
            pthread_cond_broadcast(&q->cv);
        }
    }
    pthread_mutex_unlock(&q->mu);
    return 0;
}
static int outq_open(outq_t *q, FILE *fp)
{
    memset(q, 0, sizeof(*q));
    q->fp = fp; q->pending = -1;
    q->buf[0] = (char *)malloc(OUT_BUF); q->buf[1] = (char *)malloc(OUT_BUF);
    if (!q->buf[0] || !q->buf[1]) { free(q->buf[0]); free(q->buf[1]); return -ENOMEM; }
    pthread_mutex_init(&q->mu, 0); pthread_cond_init(&q->cv, 0);
    q->threaded = pthread_create(&q->tid, 0, outq_main, q) == 0;   /* no thread: the walk writes itself */
    return 0;
}
static void outq_flush(outq_t *q) /* hand the filled buffer over, continue in the other one */
{
    const int k = q->fill;
    if (q->len[k] == 0) return;
    if (!q->threaded) { if (fwrite(q->buf[k], 1, q->len[k], q->fp) != q->len[k]) q->err = 1; q->len[k] = 0; return; }
    pthread_mutex_lock(&q->mu);
    while (q->pending >= 0) pthread_cond_wait(&q->cv, &q->mu);     /* the other buffer is still being written */
    q->pending = k; q->fill = k ^ 1;
    pthread_cond_broadcast(&q->cv);
    pthread_mutex_unlock(&q->mu);
}
static int outq_put(outq_t *q, const char *s, size_t l)
{
    if (l > OUT_BUF) { /* a record larger than a buffer (a chromosome-long unitig): drain, then write it directly */
        outq_flush(q);
        if (q->threaded) { pthread_mutex_lock(&q->mu); while (q->pending >= 0) pthread_cond_wait(&q->cv, &q->mu); pthread_mutex_unlock(&q->mu); }
        if (fwrite(s, 1, l, q->fp) != l) q->err = 1;
        return q->err ? -EIO : 0;
    }
    if (q->len[q->fill] + l > OUT_BUF) outq_flush(q);
    memcpy(q->buf[q->fill] + q->len[q->fill], s, l);
    q->len[q->fill] += l;
    return q->err ? -EIO : 0;
}
static int outq_close(outq_t *q)
{
    outq_flush(q);
    if (q->threaded) {
        pthread_mutex_lock(&q->mu);
        while (q->pending >= 0) pthread_cond_wait(&q->cv, &q->mu);
        q->quit = 1;
        pthread_cond_broadcast(&q->cv);
        pthread_mutex_unlock(&q->mu);
        pthread_join(q->tid, 0);
    }
    pthread_mutex_destroy(&q->mu); pthread_cond_destroy(&q->cv);
    free(q->buf[0]); free(q->buf[1]);
    return q->err ? -EIO : 0;
}

static inline uint64_t *st_map(const walk_t *w, int which) { return which == ST_USED ? w->used : which == ST_BEND ? w->bend : w->visited; }
static inline uint64_t ov_hash(uint64_t k) { k ^= k >> 31; k *= 0x9E3779B97F4A7C15ull; return k ^ (k >> 29); }
static int ov_has(const spec_t *sp, uint64_t key)
{
    uint32_t h;
    if (!sp->n) return 0;
    for (h = (uint32_t)ov_hash(key) & (sp->cap - 1); sp->keys[h]; h = (h + 1) & (sp->cap - 1)) if (sp->keys[h] == key + 1) return 1;
    return 0;
}
static int ov_add(spec_t *sp, uint64_t key) /* 1 = new */
{
    uint32_t h;
    if (2 * (sp->n + 1) > sp->cap) {
        const uint32_t ncap = sp->cap ? 2 * sp->cap : 1024;
        uint64_t *nk = (uint64_t *)calloc(ncap, 8);
        uint32_t i;
        if (!nk) { sp->err = -ENOMEM; return 0; }
        for (i = 0; i < sp->cap; ++i) if (sp->keys[i]) { uint32_t g = (uint32_t)ov_hash(sp->keys[i] - 1) & (ncap - 1); while (nk[g]) g = (g + 1) & (ncap - 1); nk[g] = sp->keys[i]; }
        free(sp->keys); sp->keys = nk; sp->cap = ncap;
    }
    for (h = (uint32_t)ov_hash(key) & (sp->cap - 1); sp->keys[h]; h = (h + 1) & (sp->cap - 1)) if (sp->keys[h] == key + 1) return 0;
    sp->keys[h] = key + 1; ++sp->n;
    return 1;
}
static inline void log_push(uint64_t **a, size_t *n, size_t *m, uint64_t v, int *err)
{
    if (*n == *m) { const size_t nm = *m ? 2 * *m : 4096; uint64_t *q = (uint64_t *)realloc(*a, nm * 8); if (!q) { *err = -ENOMEM; return; } *a = q; *m = nm; }
    (*a)[(*n)++] = v;
}
static inline int st_get(walk_t *w, int which, uint64_t x)
{
    spec_t *sp = w->sp;
    if (bit_get(st_map(w, which), x)) return 1;              /* set in the shared map: final */
    if (!sp) return 0;
    {
        const uint64_t key = (uint64_t)which << 62 | x;
        const int v = ov_has(sp, key);                          /* set by an earlier walk of this chunk, not committed yet */
        log_push(&sp->rlog, &sp->n_r, &sp->m_r, key | (uint64_t)v << 61, &sp->err);
        return v;
    }
}
static inline void st_set(walk_t *w, int which, uint64_t x)
{
    spec_t *sp = w->sp;
    if (!sp) { bit_set(st_map(w, which), x); return; }
    if (bit_get(st_map(w, which), x)) return;
    /* logged whether or not an earlier seed of the chunk put the bit into the overlay first: that seed may fail validation and not set it again when it is
     * re-run, and this seed's log must then still carry the bit (a write without a read in front -- mark_used, the bend of a forward bifurcation -- is not
     * caught by the read log).  A key logged twice is set twice. */
    { const uint64_t key = (uint64_t)which << 62 | x; ov_add(sp, key); log_push(&sp->wlog, &sp->n_w, &sp->m_w, key, &sp->err); }
}

static void mark_used(walk_t *w, const uint64_t x[3]) /* set_bits, unitig.c:22-36 (sorted == NULL) */
{
    uint64_t k;
    if (w->sorted) for (k = 0; k < x[2]; ++k) { st_set(w, ST_USED, w->sorted[x[0] + k] >> 2); st_set(w, ST_USED, w->sorted[x[1] + k] >> 2); }
    else for (k = 0; k < x[2]; ++k) { st_set(w, ST_USED, x[0] + k); st_set(w, ST_USED, x[1] + k); }
}

/* Coverage string (unitig.c:251-255: '"' = one read, one more per read that covers the base, capped at '~').  Every accepted extension adds one read
 * over [rbeg, new end) -- min('~', base + reads) is the same whenever the cap is applied, so there are two ways to add it and each walk gets the one
 * that suits it.  The first COV_DIRECT reads of a walk are added at once: a saturating byte increment over ~100 bytes, a handful of vector operations
 * (reads with errors: 4*10^7 unitigs of one to three reads; a difference array materialised by a scalar prefix sum over the whole string, twice per seed,
 * was a sixth of a walker's time there).  A walk that goes on is a long one: from then on the reads go into a difference array (two integer updates per
 * read instead of a hundred bytes: the genome-long walks of error-free reads measured 43 ns per read this way and 68 ns the other), materialised over the
 * stretch they touched when the string is needed. */
#define COV_DIRECT 8
typedef struct { char *s; int32_t *d; size_t l, m, d_m, lo, hi; int n_add; } cov_t;   /* d: pending reads over [lo, hi) as differences (hi = 0: none) */
static int cov_reserve(cov_t *c, size_t need)
{
    if (need <= c->m) return 0;
    size_t m = c->m ? c->m : 256;
    while (m < need) m <<= 1;
    char *p = (char *)realloc(c->s, m);
    if (!p) return -ENOMEM;
    c->s = p;
    memset(c->s + c->m, '!', m - c->m);                   /* '!' = no read yet */
    c->m = m;
    return 0;
}
static inline int cov_add(cov_t *c, size_t from, size_t to) /* one more read over [from, to) */
{
    if (cov_reserve(c, to + 2)) return -ENOMEM;
    if (c->n_add < COV_DIRECT) {
        unsigned char *q = (unsigned char *)c->s;
        size_t i;
        for (i = from; i < to; ++i) { const unsigned char v = (unsigned char)(q[i] + 1); q[i] = v > '~' ? '~' : v; }
        ++c->n_add;
    } else {
        if (c->d_m < c->m) {
            int32_t *q = (int32_t *)realloc(c->d, c->m * sizeof(int32_t));
            if (!q) return -ENOMEM;
            memset(q + c->d_m, 0, (c->m - c->d_m) * sizeof(int32_t));
            c->d = q; c->d_m = c->m;
        }
        ++c->d[from]; --c->d[to];
        if (c->hi == 0 || from < c->lo) c->lo = from;
        if (to > c->hi) c->hi = to;
    }
    if (to > c->l) c->l = to;
    return 0;
}
static void cov_flush(cov_t *c, size_t l) /* materialise what is pending, keep [0, l); everything beyond is dropped */
{
    if (c->hi) {
        size_t i;
        int32_t run = 0;
        for (i = c->lo; i < c->hi; ++i) {
            run += c->d[i]; c->d[i] = 0;
            { const int v = (unsigned char)c->s[i] + run; c->s[i] = (char)(v > '~' ? '~' : v); }
        }
        c->d[c->hi] = 0;                                    /* (the last read's closing -1) */
        c->hi = c->lo = 0;
    }
    if (c->l > l) memset(c->s + l, '!', c->l - l);
    c->l = l; c->n_add = 0;
}

/* unitig_unidir, unitig.c:227-262.  `cur` = table row of the read at the right end of s. */
static int unidir(walk_t *w, uint64_t cur, str_t *s, cov_t *cov, int beg0, uint64_t k0, uint64_t *end, int *is_loop)
{
    const fmdh_slim_t *t = w->t;
    const fmdh_wrec_t *W = t->w;
    int beg = beg0, ori_l = (int)s->l, n_reads = 0;
    uint32_t ahead[JUMP_DIST];              /* ahead[i % JUMP_DIST] = the row step i + JUMP_DIST will visit, as far as known */
    uint64_t step = 0;
    int q;
    for (q = 0; q < JUMP_DIST; ++q) ahead[q] = 0xffffffffu;
    *is_loop = 0;
    for (;; ++step) {
        const fmdh_wrec_t *h = &W[cur];
        if (w->sp && w->sp->budget-- == 0) { w->err = -EAGAIN; return -1; }      /* too long to speculate on: this seed is walked at commit */
        if (w->have_far) {   /* hints: the line of the step JUMP_DIST links on; half way there, when that line has arrived, the bitmap words its step will test and set */
            const uint32_t far = h->far, mid = ahead[(step + JUMP_DIST / 2) % JUMP_DIST];
            if (far != 0xffffffffu) __builtin_prefetch(&W[far]);
            if (mid != 0xffffffffu) {
                const fmdh_wrec_t *m = &W[mid];
                __builtin_prefetch(&w->bend[m->k0 >> 6]);
                __builtin_prefetch(&w->used[m->k0 >> 6]); if (((uint64_t)mid ^ 1) < t->n) __builtin_prefetch(&w->used[W[mid ^ 1].k0 >> 6]);   /* (k[1] = k[0] of the other strand: the same 64 bytes) */
            }
            ahead[step % JUMP_DIST] = far;
        }
        if ((h->bits & FMDH_W_PLAIN) && !w->no_plain) {                          /* the plain step, from one line (fmdh_wrec_t) and the neighbour's */
            const fmdh_wrec_t *nb = &W[h->nxt];
            const uint64_t kx[3] = {nb->k0, W[h->nxt ^ 1].k0, nb->k2};
            const int ext = h->ext_len, rbeg = beg + (int)h->rbeg;
            int k;
            w->last = cur; w->n_nei = 1;
            if (str_reserve(s, (size_t)ori_l + (size_t)ext + 1)) return -1;
            for (k = 0; k < ext; ++k) s->s[ori_l + k] = (char)(((h->ext[k >> 2] >> (2 * (k & 3))) & 3) + 1);
            s->l = (size_t)ori_l + (size_t)ext;
            if (kx[0] == *end) break;
            if (st_get(w, ST_BEND, kx[0]) || (h->bits & FMDH_W_CL)) { st_set(w, ST_BEND, kx[0]); break; }
            if (kx[0] == k0) { *is_loop = 1; break; }
            if (kx[1] == *end) { w->n_nei = 0; break; }
            *end = kx[1];
            mark_used(w, kx);
            ++n_reads;
            if (cov_add(cov, (size_t)rbeg, s->l)) return -1;
            beg = rbeg; ori_l = (int)s->l;
            cur = h->nxt;
            continue;
        }
        {
            fmdh_rowv_t r;
            uint64_t kx[3];
            const uint32_t nxt = h->nxt;
            int rbeg;
            fmdh_slim_row(t, cur, &r);
            w->last = cur; w->n_nei = r.n_nei;                                       /* the list try_right leaves behind (unitig.c:181-184) */
            if (r.status != 0 || r.rbeg < 0) { w->n_nei = r.status == 0 ? r.n_nei : 0; break; }   /* try_right < 0 */
            rbeg = beg + r.rbeg;
            if (r.n_nei > 1) { st_set(w, ST_BEND, *end); break; }                     /* forward bifurcation */
            /* the `$neighbour$` interval: the neighbour's own row has it (the next one the walk reads anyway).  A row with one neighbour and no
             * link to it is a row of an incomplete table (the neighbour is a non-contained read longer than min_match: it has a row) */
            if (r.n_nei != 1 || nxt == 0xffffffffu) { w->err = -EDOM; return -1; }
            if (W[nxt].bits & FMDH_W_BIG) { fmdh_rowv_t nb; fmdh_slim_row(t, nxt, &nb); kx[0] = nb.k[0]; kx[1] = nb.k[1]; kx[2] = nb.k[2]; }
            else { kx[0] = W[nxt].k0; kx[1] = ((uint64_t)nxt ^ 1) < t->n ? W[nxt ^ 1].k0 : ~0ull; kx[2] = W[nxt].k2; }
            /* the bases fm6_get_nei appended (unitig.c:139) */
            if (str_reserve(s, (size_t)ori_l + (size_t)r.ext_len + 1)) return -1;
            fmdh_slim_ext(t, cur, &r, s->s + ori_l);
            s->l = (size_t)ori_l + (size_t)r.ext_len;
            if (kx[0] == *end) break;                                                /* b>>c>>a><a */
            if (!st_get(w, ST_BEND, kx[0])) {                                        /* check_left (unitig.c:206-225), decided when the table was linked */
                if (h->bits & FMDH_W_UNDEC) { w->err = -EDOM; return -1; }           /* the table is incomplete: the link passes leave no such row */
                if (h->bits & FMDH_W_CL) { st_set(w, ST_BEND, kx[0]); break; }       /* backward bifurcation */
            } else { st_set(w, ST_BEND, kx[0]); break; }
            if (kx[0] == k0) { *is_loop = 1; break; }                                /* a>>b>>c>>a */
            if (kx[1] == *end) { w->n_nei = 0; break; }                              /* b>>c>>a>>a: cut the last link */
            *end = kx[1];
            mark_used(w, kx);
            ++n_reads;
            if (cov_add(cov, (size_t)rbeg, s->l)) return -1;                          /* ++ over [rbeg, ori_l), '"' for the new bases */
            beg = rbeg; ori_l = (int)s->l;
            cur = nxt;
        }
    }
    s->l = (size_t)ori_l;
    cov_flush(cov, (size_t)ori_l);
    return n_reads;
}

/* eight bytes at a time from both ends (byte swap), the middle byte by byte */
static void reverse(size_t l, char *s)
{
    size_t i = 0, j = l;
    while (j - i >= 16) {
        uint64_t a, b;
        memcpy(&a, s + i, 8); memcpy(&b, s + j - 8, 8);
        a = __builtin_bswap64(a); b = __builtin_bswap64(b);
        memcpy(s + i, &b, 8); memcpy(s + j - 8, &a, 8);
        i += 8; j -= 8;
    }
    for (; i + 1 < j; ++i, --j) { char t = s[i]; s[i] = s[j - 1]; s[j - 1] = t; }
}
/* reverse complement of nt6 codes: 1..4 -> 5 - v, 0 and 5 stay.  Words whose bytes are all 1..4 (nearly all) are complemented as
 * 0x0505.. - w (no byte borrows); a word with a 0 or a 5 in it goes byte by byte. */
static inline uint64_t comp6_word(uint64_t w)
{
    const uint64_t ones = 0x0101010101010101ull, five = 0x0505050505050505ull, x = w ^ five;
    if ((((w - ones) & ~w) | ((x - ones) & ~x)) & 0x8080808080808080ull) {   /* some byte is 0 or 5 */
        uint64_t r = 0; int k;
        for (k = 0; k < 8; ++k) { const unsigned v = (unsigned)(w >> (8 * k)) & 0xff; r |= (uint64_t)((v >= 1 && v <= 4) ? 5 - v : v) << (8 * k); }
        return r;
    }
    return five - w;
}
static void revcomp6(size_t l, char *s)
{
    size_t i = 0, j = l;
    while (j - i >= 16) {
        uint64_t a, b;
        memcpy(&a, s + i, 8); memcpy(&b, s + j - 8, 8);
        a = __builtin_bswap64(comp6_word(a)); b = __builtin_bswap64(comp6_word(b));
        memcpy(s + i, &b, 8); memcpy(s + j - 8, &a, 8);
        i += 8; j -= 8;
    }
    for (; i + 1 < j; ++i, --j) {
        int a = s[i], b = s[j - 1];
        s[i] = (char)((b >= 1 && b <= 4) ? 5 - b : b);
        s[j - 1] = (char)((a >= 1 && a <= 4) ? 5 - a : a);
    }
    if (i < j) { int a = s[i]; s[i] = (char)((a >= 1 && a <= 4) ? 5 - a : a); }
}

typedef struct { uint64_t x, y; } link_t;

/* nt6 codes -> the letters mag_v_write prints ("ACGT"[c - 1], mag.c:168: code 5 prints as NUL); returns non-zero if a NUL was
 * written.  Sixteen bases per shuffle where the CPU has SSSE3 (a record of raw reads is half bases). */
#if defined(__x86_64__)
#include <immintrin.h>
__attribute__((target("ssse3"))) static int bases_to_text_ssse3(size_t n, const char *b, char *q)
{
    const __m128i lut = _mm_setr_epi8(0, 'A', 'C', 'G', 'T', 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0), zero = _mm_setzero_si128();
    size_t z = 0;
    int cut = 0;
    for (; z + 16 <= n; z += 16) {
        const __m128i t = _mm_shuffle_epi8(lut, _mm_loadu_si128((const __m128i *)(b + z)));
        _mm_storeu_si128((__m128i *)(q + z), t);
        cut |= _mm_movemask_epi8(_mm_cmpeq_epi8(t, zero));   /* any code that is not A/C/G/T (5 = N; 0 never occurs in a unitig) */
    }
    for (; z < n; ++z) { const unsigned v = (unsigned char)b[z]; q[z] = v < 6 ? "\0ACGT"[v] : 0; cut |= q[z] == 0; }
    return cut;
}
#endif
static int bases_to_text(size_t n, const char *b, char *q)
{
    size_t z;
    int cut = 0;
#if defined(__x86_64__)
    static int have = -1;
    if (have < 0) have = __builtin_cpu_supports("ssse3") ? 1 : 0;
    if (have) return bases_to_text_ssse3(n, b, q);
#endif
    for (z = 0; z < n; ++z) { const unsigned v = (unsigned char)b[z]; q[z] = v < 6 ? "\0ACGT"[v] : 0; cut |= q[z] == 0; }
    return cut;
}

/* decimal digits of v (what "%lld" prints) at p; returns the number of bytes.  A record of raw reads is one header of up to eight
 * numbers over ~200 bytes of text, and printf's parser is most of what such a record costs. */
static inline size_t put_ll(char *p, long long v)
{
    char t[24];
    size_t n = 0, i;
    unsigned long long u = v < 0 ? 0ull - (unsigned long long)v : (unsigned long long)v;
    do { t[n++] = (char)('0' + u % 10); u /= 10; } while (u);
    if (v < 0) t[n++] = '-';
    for (i = 0; i < n; ++i) p[i] = t[n - 1 - i];
    return n;
}

static int put_links(str_t *o, const link_t *a, int n)
{
    int k;
    if (str_reserve(o, o->l + 32 * (size_t)(n + 1) + 8)) return -1;
    o->s[o->l++] = '\t';
    for (k = 0; k < n; ++k) {
        o->l += put_ll(o->s + o->l, (long long)a[k].x); o->s[o->l++] = ',';
        o->l += put_ll(o->s + o->l, (long long)(int32_t)a[k].y); o->s[o->l++] = ';';
    }
    if (n == 0) o->s[o->l++] = '.';
    return 0;
}

/* ---- one seed: unitig1 (unitig.c:274-317), the visited test of unitig_core (unitig.c:336-339) and mag_v_write (mag.c:149-174).
 * The buffers belong to the caller (one set per thread).  Returns 1 with the record in o->s[0 .. *wl), 0 when the seed yields nothing,
 * < 0 on error. */
typedef struct { str_t s, o; cov_t cov; link_t *nei[2]; uint32_t cap_nei; } seedbuf_t;
static int seedbuf_init(seedbuf_t *b, uint32_t cap_nei)
{
    memset(b, 0, sizeof(*b));
    b->cap_nei = cap_nei;
    b->nei[0] = (link_t *)malloc(cap_nei * sizeof(link_t)); b->nei[1] = (link_t *)malloc((cap_nei + 1) * sizeof(link_t));
    return b->nei[0] && b->nei[1] ? 0 : -ENOMEM;
}
static void seedbuf_free(seedbuf_t *b) { free(b->nei[0]); free(b->nei[1]); free(b->s.s); free(b->o.s); free(b->cov.s); free(b->cov.d); }

/* the sequence of row i: its own copy, or the reverse complement of the other strand's (the table keeps the bases of a read once) */
static int seed_seq(const walk_t *w, uint64_t i, const fmdh_rowv_t *r, char *dst)
{
    fmdh_rowv_t p;
    if (fmdh_slim_own_seq(r, dst)) return 0;
    if ((i ^ 1) >= w->t->n) return -EDOM;
    fmdh_slim_row(w->t, i ^ 1, &p);
    if (p.len != r->len || !fmdh_slim_own_seq(&p, dst)) return -EDOM;
    revcomp6((size_t)r->len, dst);
    return 0;
}
/* the neighbours of row `last` as the MAG record lists them: x[0] of `$neighbour$` and the overlap length */
static int end_list(const walk_t *w, int n, link_t *dst, uint32_t cap)
{
    fmdh_rowv_t lr;
    int k;
    if (n <= 0) return 0;
    if ((uint32_t)n > cap) return -ERANGE;          /* more neighbours than any row of this table holds */
    fmdh_slim_row(w->t, w->last, &lr);
    if (n > lr.n_stored) return -ERANGE;
    for (k = 0; k < n; ++k) { uint64_t x1; fmdh_slim_nei(w->t, w->last, &lr, k, &dst[k].x, &x1, &dst[k].y); }
    return 0;
}

static int walk_seed(walk_t *w, uint64_t i, seedbuf_t *b, size_t *wl)
{
    const uint64_t *sorted = w->sorted;
    const int min_match = w->min_match;
    str_t *s = &b->s, *o = &b->o;
    cov_t *cov = &b->cov;
    link_t **nei = b->nei;
    fmdh_rowv_t r;
    uint64_t end[2];
    int n_nei[2] = {0, 0}, n_reads, is_loop = 0, seed_len, done_loop = 0, e;
    fmdh_slim_row(w->t, i, &r);
    if (r.status == FMDH_W_ST_INVALID) return -ERANGE;
    if (sorted && st_get(w, ST_USED, i)) return 0;              /* used (unitig.c:282: by sequence id with -r) */
    if (r.len <= min_match) return 0;                           /* too short */
    if (!sorted && st_get(w, ST_USED, r.rank)) return 0;        /* used (unitig.c:289: by rank) */
    mark_used(w, r.k);
    if (r.status != 0) return 0;                                /* contained */
    seed_len = r.len;
    if (str_reserve(s, (size_t)seed_len + 1)) return -ENOMEM;
    if ((e = seed_seq(w, i, &r, s->s)) != 0) return e;
    s->l = (size_t)seed_len;
    cov_flush(cov, 0);
    if (cov_add(cov, 0, (size_t)seed_len)) return -ENOMEM;
    n_reads = 1;
    end[0] = r.k[1]; end[1] = r.k[0];
    const int tm = w->timing;
    double p0 = tm ? wall_s() : 0, p1;
    if (r.has_ovlp) { /* left-wards extension of the unitig = right-wards of this strand */
        int m = unidir(w, i, s, cov, 0, r.k[0], &end[0], &is_loop);
        if (tm) { p1 = wall_s(); w->t_uni += p1 - p0; p0 = p1; w->n_hops += m > 0 ? (uint64_t)m : 0; }
        if (m < 0) return w->err ? w->err : -ENOMEM;
        n_reads += m;
        if ((e = end_list(w, w->n_nei, nei[0], b->cap_nei)) != 0) return e;
        n_nei[0] = w->n_nei;
        if (is_loop) { nei[1][0].x = end[0]; nei[1][0].y = nei[0][0].y; n_nei[1] = 1; done_loop = 1; }
    }
    if (!done_loop) { /* the other direction, from the reverse strand of the seed (unitig.c:310-315) */
        int m;
        cov_flush(cov, s->l);
        revcomp6(s->l, s->s); reverse(s->l, cov->s);
        if (tm) { p1 = wall_s(); w->t_turn += p1 - p0; p0 = p1; }
        m = unidir(w, i ^ 1, s, cov, (int)s->l - seed_len, r.k[1], &end[1], &is_loop);
        if (tm) { p1 = wall_s(); w->t_uni += p1 - p0; p0 = p1; w->n_hops += m > 0 ? (uint64_t)m : 0; }
        if (m < 0) return w->err ? w->err : -ENOMEM;
        n_reads += m;
        if ((e = end_list(w, w->n_nei, nei[1], b->cap_nei)) != 0) return e;
        n_nei[1] = w->n_nei;
    }
    /* ---- unitig_core: keep each unitig once (unitig.c:336-339) */
    if (st_get(w, ST_VIS, end[0])) return 0;
    st_set(w, ST_VIS, end[0]);
    if (st_get(w, ST_VIS, end[1])) return 0;
    st_set(w, ST_VIS, end[1]);
    /* ---- mag_v_write (mag.c:149-174) */
    o->l = 0;
    if (str_reserve(o, 2 * s->l + 128)) return -ENOMEM;
    o->s[o->l++] = '@'; o->l += put_ll(o->s + o->l, (long long)end[0]); o->s[o->l++] = ':'; o->l += put_ll(o->s + o->l, (long long)end[1]);
    o->s[o->l++] = '\t'; o->l += put_ll(o->s + o->l, (long long)n_reads);
    if (put_links(o, nei[0], n_nei[0]) || put_links(o, nei[1], n_nei[1])) return -ENOMEM;
    if (str_reserve(o, o->l + 2 * s->l + 8)) return -ENOMEM;
    o->s[o->l++] = '\n';
    {
        const int cut = bases_to_text(s->l, s->s, o->s + o->l);      /* cut: a base that prints as NUL (see below) */
        o->l += s->l;
        memcpy(o->s + o->l, "\n+\n", 3); o->l += 3;
        memcpy(o->s + o->l, cov->s, s->l); o->l += s->l;
        o->s[o->l++] = '\n';
        /* the reference prints the record with fputs (unitig.c:354): a base that is not A/C/G/T
         * becomes "ACGT"[4] = NUL (mag.c:168) and cuts the record there.  Reproduced as is. */
        *wl = cut && !w->full_records ? strnlen(o->s, o->l) : o->l;
    }
    if (tm) w->t_text += wall_s() - p0;
    return 1;
}

/* ---- the parallel driver ---------------------------------------------------------------------------------------------------------
 * `fermi unitig -tN` hands the seeds to N threads that race on the bitmaps (unitig.c:319-362, 394-404): fast, and a different MAG on
 * every run.  Here N threads give the MAG of -t1, byte for byte.  The seeds (odd ids, ascending) are cut into chunks; the chunks of a
 * window are walked concurrently and SPECULATIVELY against the bitmaps as the previous window left them (st_get / st_set above: private
 * overlay, read and write logs, the record formatted into the chunk's own buffer); then one thread commits the window in seed order:
 * a walk whose logged reads still hold in the bitmaps as they are NOW did exactly what the sequential walk does at this point (the
 * walk is a function of the table and of the bits it reads; reads that found a bit set are final because bits are never cleared) --
 * its writes are applied and its record goes out; a walk that read something an earlier seed of the window has changed since is run
 * again, directly, then and there.  With reads in sequencer order a window holds a few 10^5 of 10^8 reads, so two walks of one window
 * rarely meet (the log says how often).  One long unitig (error-free reads: the whole genome from the first seed) is one walk and
 * stays one thread's work. */
#define CHUNK_SEEDS 2048
#define SPEC_MAX_STEPS 1024   /* a speculative walk gives up after this many reads: a long unitig is few seeds' work however it is done, and every seed of the
                               * first window would walk the whole of it (nothing is `used` yet in the bitmaps it sees) */
typedef struct { uint32_t n_r, n_w; uint64_t out_len; int rc; } seedres_t;
typedef struct {
    walk_t w; spec_t sp; seedbuf_t b;
    seedres_t res[CHUNK_SEEDS];
    uint64_t q0, nq;            /* seeds q0 .. q0 + nq of the job (seed q = sequence id 2q + 1) */
    char *out; size_t out_l, out_m;
    volatile int pending;       /* slices of `out` the writer has not written yet */
} chunk_t;
typedef struct slice { const char *p; size_t l; chunk_t *owner; char *own; off_t at; } slice_t;   /* own: malloc'ed copy the writer frees; at: where it goes in a regular file */
#define SLICE_RING 4096
#define SLICE_WRITERS 4
typedef struct {
    FILE *fp; slice_t ring[SLICE_RING]; size_t head, tail;   /* head: next to take, tail: next free */
    int quit, err, n_writers, fd;   /* fd >= 0: the output is a regular file, every slice knows its place and several threads pwrite (one thread fills the
                                     * page cache at 2-3 GB/s; `unitig` of 5*10^7 raw reads writes 10 GB); else one thread, fwrite, in order */
    off_t file_off;
    pthread_t tid[SLICE_WRITERS]; pthread_mutex_t mu; pthread_cond_t cv;
} sliceq_t;
static void *sliceq_main(void *p)
{
    sliceq_t *q = (sliceq_t *)p;
    pthread_mutex_lock(&q->mu);
    for (;;) {
        while (q->head == q->tail && !q->quit) pthread_cond_wait(&q->cv, &q->mu);
        if (q->head == q->tail) break;
        {
            slice_t sl = q->ring[q->head % SLICE_RING];
            ++q->head;                                   /* taken (several writers: nobody else takes it; the slot is free again) */
            pthread_cond_broadcast(&q->cv);
            pthread_mutex_unlock(&q->mu);
            if (q->fd >= 0) { size_t done = 0; while (done < sl.l) { const ssize_t k = pwrite(q->fd, sl.p + done, sl.l - done, sl.at + (off_t)done); if (k <= 0) { q->err = 1; break; } done += (size_t)k; } }
            else if (sl.l && fwrite(sl.p, 1, sl.l, q->fp) != sl.l) q->err = 1;
            free(sl.own);
            if (sl.owner) __atomic_fetch_sub(&sl.owner->pending, 1, __ATOMIC_RELEASE);
            pthread_mutex_lock(&q->mu);
        }
    }
    pthread_mutex_unlock(&q->mu);
    return 0;
}
static void sliceq_put(sliceq_t *q, const char *p, size_t l, chunk_t *owner, char *own)
{
    if (owner) __atomic_fetch_add(&owner->pending, 1, __ATOMIC_RELAXED);
    pthread_mutex_lock(&q->mu);
    while (q->tail - q->head == SLICE_RING) pthread_cond_wait(&q->cv, &q->mu);
    q->ring[q->tail % SLICE_RING] = (slice_t){p, l, owner, own, q->file_off};
    q->file_off += (off_t)l;
    ++q->tail;
    pthread_cond_broadcast(&q->cv);
    pthread_mutex_unlock(&q->mu);
}
typedef struct {
    chunk_t *chunks; int n_chunks;          /* the chunks of the current window */
    volatile int next;                      /* next chunk to claim */
    int n_threads, phase_quit;
    pthread_mutex_t mu; pthread_cond_t cv; int generation, running;
    uint64_t n_seq;
} pool_t;
static void chunk_run(chunk_t *c, uint64_t n_seq)
{
    uint64_t k;
    spec_t *sp = &c->sp;
    while (__atomic_load_n(&c->pending, __ATOMIC_ACQUIRE) > 0) sched_yield();     /* its last window's records are still on their way out */
    if (sp->n) { memset(sp->keys, 0, (size_t)sp->cap * 8); sp->n = 0; }
    sp->n_r = sp->n_w = 0;
    c->out_l = 0;
    int gave_up = 0;
    uint32_t spec_steps;
    { const char *e = getenv("FMD_WALK_SPEC_STEPS"); spec_steps = e && atoi(e) > 0 ? (uint32_t)atoi(e) : SPEC_MAX_STEPS; }   /* (tests make it small) */
    for (k = 0; k < c->nq; ++k) {
        const uint64_t i = 2 * (c->q0 + k) + 1;
        const size_t r0 = sp->n_r, w0 = sp->n_w;
        size_t wl = 0;
        int rc;
        if (gave_up) { c->res[k].n_r = c->res[k].n_w = 0; c->res[k].out_len = 0; c->res[k].rc = -EAGAIN; continue; }   /* (walked at commit) */
        sp->budget = spec_steps; c->w.err = 0;
        if (c->w.seed_hints && i < n_seq) seed_hints(&c->w, i, 1);
        rc = i < n_seq ? walk_seed(&c->w, i, &c->b, &wl) : 0;
        if (sp->err && rc >= 0) rc = sp->err;
        /* a walk too long to speculate on: the seeds behind it in this chunk may lie on that very unitig (error-free reads: all of them do) and
         * would each walk it again -- they wait for the commit, where the long walk has happened and marked its reads */
        if (rc == -EAGAIN) gave_up = 1;
        if (rc == 1) {
            if (c->out_l + wl > c->out_m) { size_t m = c->out_m ? c->out_m : (size_t)1 << 20; char *q; while (m < c->out_l + wl) m <<= 1; q = (char *)realloc(c->out, m); if (!q) { rc = -ENOMEM; wl = 0; } else { c->out = q; c->out_m = m; } }
            if (rc == 1) { memcpy(c->out + c->out_l, c->b.o.s, wl); c->out_l += wl; }
        }
        c->res[k].n_r = (uint32_t)(sp->n_r - r0); c->res[k].n_w = (uint32_t)(sp->n_w - w0); c->res[k].out_len = rc == 1 ? wl : 0; c->res[k].rc = rc;
    }
}
static void *pool_main(void *p)
{
    pool_t *P = (pool_t *)p;
    int gen = 0;
    for (;;) {
        pthread_mutex_lock(&P->mu);
        while (P->generation == gen && !P->phase_quit) pthread_cond_wait(&P->cv, &P->mu);
        if (P->phase_quit) { pthread_mutex_unlock(&P->mu); return 0; }
        gen = P->generation;
        pthread_mutex_unlock(&P->mu);
        for (;;) {
            const int k = __atomic_fetch_add(&P->next, 1, __ATOMIC_RELAXED);
            if (k >= P->n_chunks) break;
            chunk_run(&P->chunks[k], P->n_seq);
        }
        pthread_mutex_lock(&P->mu);
        if (--P->running == 0) pthread_cond_broadcast(&P->cv);
        pthread_mutex_unlock(&P->mu);
    }
}

static int walk_threads(void)
{
    const char *e = getenv("FMD_WALK_THREADS");
    int nt = 16;
    if (e && atoi(e) > 0) nt = atoi(e);
    else { e = getenv("FMD_HOST_THREADS"); if (e && atoi(e) > 0) nt = atoi(e); }
    return nt > 256 ? 256 : nt;
}

static uint64_t chunk_seeds(void)   /* FMD_WALK_CHUNK: seeds per chunk (tests make it small, so that a fixture is many windows) */
{
    const char *e = getenv("FMD_WALK_CHUNK");
    const long v = e ? atol(e) : 0;
    return v > 0 && v < CHUNK_SEEDS ? (uint64_t)v : CHUNK_SEEDS;
}
static int walk_parallel(walk_t *w0, uint32_t cap_nei, FILE *out, int nt)
{
    const uint64_t n_seeds = w0->n_seq / 2;          /* odd ids below n_seq: seed q = id 2q + 1, q < n_seq / 2 */
    const uint64_t CS = chunk_seeds();
    const int per_win = 4 * nt;
    chunk_t *ch = (chunk_t *)calloc((size_t)2 * per_win, sizeof(chunk_t));   /* two windows' worth: a window's records leave while the next is walked */
    pthread_t *tid = (pthread_t *)calloc((size_t)nt, sizeof(pthread_t));
    pool_t P;
    sliceq_t Q;
    seedbuf_t mb;
    walk_t wm = *w0;                                  /* the committing thread's own walk: direct mode */
    int rc = 0, k, started = 0, q_open = 0, mb_ok = 0;
    uint64_t q = 0, n_rerun = 0, n_walked = 0, win = 0;
    double t_spec = 0, t_commit = 0;
    const int timing = getenv("FMD_TIMING") != 0;
    memset(&P, 0, sizeof(P)); memset(&Q, 0, sizeof(Q));
    if (!ch || !tid) { rc = -ENOMEM; goto done; }
    wm.sp = 0;
    if ((rc = seedbuf_init(&mb, cap_nei)) != 0) goto done;
    mb_ok = 1;
    for (k = 0; k < 2 * per_win; ++k) { ch[k].w = *w0; ch[k].w.sp = &ch[k].sp; if ((rc = seedbuf_init(&ch[k].b, cap_nei)) != 0) goto done; }
    Q.fp = out; Q.fd = -1; Q.n_writers = 0; pthread_mutex_init(&Q.mu, 0); pthread_cond_init(&Q.cv, 0);
    {   /* a regular file that is not in append mode: slices are placed by offset */
        struct stat sb;
        const int fd = fileno(out);
        fflush(out);
        if (fd >= 0 && !getenv("FMD_WALK_ONE_WRITER") && fstat(fd, &sb) == 0 && S_ISREG(sb.st_mode) && !(fcntl(fd, F_GETFL) & O_APPEND)) {
            const off_t at = lseek(fd, 0, SEEK_CUR);
            if (at >= 0) { Q.fd = fd; Q.file_off = at; }
        }
    }
    for (k = 0; k < (Q.fd >= 0 ? SLICE_WRITERS : 1); ++k) { if (pthread_create(&Q.tid[k], 0, sliceq_main, &Q) != 0) break; ++Q.n_writers; }
    if (Q.n_writers == 0) { rc = -EAGAIN; goto done; }
    q_open = 1;
    pthread_mutex_init(&P.mu, 0); pthread_cond_init(&P.cv, 0);
    P.n_threads = nt; P.n_seq = w0->n_seq;
    for (k = 0; k < nt - 1; ++k) { if (pthread_create(&tid[k], 0, pool_main, &P) != 0) break; ++started; }
    /* Window w + 1 is walked (by the pool) WHILE window w is committed (by this thread): the walkers then see the bitmaps somewhere between "as window w - 1
     * left them" and "as window w leaves them" -- any of which is fine, because a walk is only accepted if what it read still holds when ITS turn comes.
     * The first window is walked by everybody. */
    int nc_next = 0;
    for (k = 0; k < per_win && q < n_seeds; ++k, ++nc_next) { ch[k].q0 = q; ch[k].nq = n_seeds - q < CS ? n_seeds - q : CS; q += ch[k].nq; }
    pthread_mutex_lock(&P.mu);
    P.chunks = ch; P.n_chunks = nc_next; P.next = 0; P.running = started; ++P.generation;
    pthread_cond_broadcast(&P.cv);
    pthread_mutex_unlock(&P.mu);
    for (;;) { const int c = __atomic_fetch_add(&P.next, 1, __ATOMIC_RELAXED); if (c >= nc_next) break; chunk_run(&ch[c], w0->n_seq); }
    pthread_mutex_lock(&P.mu);
    while (P.running > 0) pthread_cond_wait(&P.cv, &P.mu);
    pthread_mutex_unlock(&P.mu);
    for (win = 0; nc_next > 0 && rc == 0; ++win) {
        chunk_t *cw = ch + (win & 1) * per_win, *cn = ch + ((win + 1) & 1) * per_win;
        const int nc = nc_next;
        const double t_win0 = wall_s();
        int async = 0;
        nc_next = 0;
        for (k = 0; k < per_win && q < n_seeds; ++k, ++nc_next) { cn[k].q0 = q; cn[k].nq = n_seeds - q < CS ? n_seeds - q : CS; q += cn[k].nq; }
        if (nc_next && started > 0 && !getenv("FMD_WALK_NO_OVERLAP")) {   /* ---- the next window's chunks, concurrently with the commit below */
            pthread_mutex_lock(&P.mu);
            P.chunks = cn; P.n_chunks = nc_next; P.next = 0; P.running = started; ++P.generation;
            pthread_cond_broadcast(&P.cv);
            pthread_mutex_unlock(&P.mu);
            async = 1;
        }
        /* ---- commit in seed order */
        for (k = 0; k < nc && rc == 0; ++k) {
            chunk_t *c = &cw[k];
            const uint64_t *rl = c->sp.rlog, *wlg = c->sp.wlog;
            size_t ro = 0, wo = 0, oo = 0, run_beg = 0, pf_r = 0, pf_w = 0;
            uint64_t j;
            for (j = 0; j < c->nq && rc == 0; ++j) {
                const seedres_t *sr = &c->res[j];
                size_t z;
                int ok = sr->rc >= 0;
                /* the bits the next seeds' logs name, on their way into the cache */
                while (pf_r < c->sp.n_r && pf_r < ro + 64) { const uint64_t e = rl[pf_r++]; __builtin_prefetch(&st_map(&wm, (int)(e >> 62))[(e & ((1ull << 61) - 1)) >> 6]); }
                while (pf_w < c->sp.n_w && pf_w < wo + 64) { const uint64_t e = wlg[pf_w++]; __builtin_prefetch(&st_map(&wm, (int)(e >> 62))[(e & ((1ull << 61) - 1)) >> 6]); }
                for (z = 0; z < sr->n_r && ok; ++z) { const uint64_t e = rl[ro + z]; ok = bit_get(st_map(&wm, (int)(e >> 62)), e & ((1ull << 61) - 1)) == (int)(e >> 61 & 1); }
                if (sr->n_r || sr->n_w) ++n_walked;
                if (ok) {
                    for (z = 0; z < sr->n_w; ++z) { const uint64_t e = wlg[wo + z]; bit_set(st_map(&wm, (int)(e >> 62)), e & ((1ull << 61) - 1)); }
                } else {   /* something it read has changed (or it failed for want of memory): again, on the bitmaps as they are now */
                    size_t wl = 0;
                    const int r2 = walk_seed(&wm, 2 * (c->q0 + j) + 1, &mb, &wl);
                    ++n_rerun;
                    if (r2 < 0) { rc = r2; break; }
                    if (oo > run_beg) sliceq_put(&Q, c->out + run_beg, oo - run_beg, c, 0);
                    run_beg = oo + sr->out_len;
                    if (r2 == 1) { char *cp = (char *)malloc(wl ? wl : 1); if (!cp) { rc = -ENOMEM; break; } memcpy(cp, mb.o.s, wl); sliceq_put(&Q, cp, wl, 0, cp); }
                }
                ro += sr->n_r; wo += sr->n_w; oo += sr->out_len;
            }
            if (rc == 0 && oo > run_beg) sliceq_put(&Q, c->out + run_beg, oo - run_beg, c, 0);
        }
        if (Q.err) rc = -EIO;
        t_commit += wall_s() - t_win0;
        {
            const double t1 = wall_s();
            if (async) { pthread_mutex_lock(&P.mu); while (P.running > 0) pthread_cond_wait(&P.cv, &P.mu); pthread_mutex_unlock(&P.mu); }
            else if (nc_next && rc == 0) {   /* (no pool, or FMD_WALK_NO_OVERLAP: one after the other, everybody walking) */
                pthread_mutex_lock(&P.mu);
                P.chunks = cn; P.n_chunks = nc_next; P.next = 0; P.running = started; ++P.generation;
                pthread_cond_broadcast(&P.cv);
                pthread_mutex_unlock(&P.mu);
                for (;;) { const int c = __atomic_fetch_add(&P.next, 1, __ATOMIC_RELAXED); if (c >= nc_next) break; chunk_run(&cn[c], w0->n_seq); }
                pthread_mutex_lock(&P.mu);
                while (P.running > 0) pthread_cond_wait(&P.cv, &P.mu);
                pthread_mutex_unlock(&P.mu);
            }
            t_spec += wall_s() - t1;      /* what of the walking did not hide under the commit */
        }
    }
    if (timing) fprintf(stderr, "[M::%s] %d threads, %llu windows of %d chunks x %d seeds: %llu seeds walked speculatively, %llu of them run again at commit; commit %.3f s, waiting for the walkers of the next window after it %.3f s\n", __func__, nt,
                        (unsigned long long)win, per_win, (int)CS, (unsigned long long)n_walked, (unsigned long long)n_rerun, t_commit, t_spec);
    if (timing) fprintf(stderr, "[M::%s] walks run at the commit: %llu reads appended in %.3f s, turning the strings round %.3f s, record text %.3f s\n", __func__,
                        (unsigned long long)wm.n_hops, wm.t_uni, wm.t_turn, wm.t_text);
done:
    if (started || P.generation) {
        pthread_mutex_lock(&P.mu); P.phase_quit = 1; pthread_cond_broadcast(&P.cv); pthread_mutex_unlock(&P.mu);
        for (k = 0; k < started; ++k) pthread_join(tid[k], 0);
    }
    if (q_open) {
        pthread_mutex_lock(&Q.mu); Q.quit = 1; pthread_cond_broadcast(&Q.cv); pthread_mutex_unlock(&Q.mu);
        for (k = 0; k < Q.n_writers; ++k) pthread_join(Q.tid[k], 0);
        if (Q.fd >= 0 && lseek(Q.fd, Q.file_off, SEEK_SET) < 0) Q.err = 1;   /* whoever writes to the stream next goes on behind the records */
        if (Q.err && !rc) rc = -EIO;
    }
    if (ch) for (k = 0; k < 2 * per_win; ++k) { seedbuf_free(&ch[k].b); free(ch[k].sp.keys); free(ch[k].sp.rlog); free(ch[k].sp.wlog); free(ch[k].out); }
    if (mb_ok) seedbuf_free(&mb);
    free(ch); free(tid);
    return rc;
}

int fmdh_unitig_walk(const fmdh_ovlp_table_t *t, uint64_t n_seq, int min_match, const uint64_t *sorted, FILE *out)
{
    return fmdh_unitig_walk_opt(t, n_seq, min_match, sorted, out, 0);
}
/* over a table of packed rows (the tests' tables; a caller that holds fmd_ovlp_packed_batch's output): made slim first, then the one walk there is */
int fmdh_unitig_walk_opt(const fmdh_ovlp_table_t *t, uint64_t n_seq, int min_match, const uint64_t *sorted, FILE *out, int flags)
{
    fmdh_slim_t *s = 0;
    int rc;
    if (!t) return -EINVAL;
    if (n_seq >= 0xffffffffull || t->n >= 0xffffffffull) return -ERANGE;   /* rows are 32-bit ids */
    if ((rc = fmdh_slim_from_table(t, fmdh_host_threads(), &s)) != 0) return rc;
    rc = fmdh_unitig_walk_slim(s, n_seq, min_match, sorted, out, flags);
    fmdh_slim_free(s);
    return rc;
}
int fmdh_unitig_walk_slim(fmdh_slim_t *t, uint64_t n_seq, int min_match, const uint64_t *sorted, FILE *out, int flags)
{
    walk_t w;
    seedbuf_t b;
    uint64_t i, nw = (n_seq + 63) / 64;
    int rc = 0, b_ok = 0;
    double t_begin = wall_s();
    outq_t oq;
    int oq_open = 0;
    uint32_t cap_nei;
    if (!t) return -EINVAL;
    cap_nei = t->max_nei > 1 ? t->max_nei : 1;
    memset(&w, 0, sizeof(w));
    w.t = t; w.n_seq = n_seq; w.min_match = min_match; w.sorted = sorted; w.full_records = (flags & FMDH_WALK_FULL_RECORDS) != 0;
    w.used = (uint64_t *)calloc(nw + 1, 8); w.bend = (uint64_t *)calloc(nw + 1, 8); w.visited = (uint64_t *)calloc(nw + 1, 8);
    if (!w.used || !w.bend || !w.visited) { rc = -ENOMEM; goto done; }
    if (n_seq >= 0xffffffffull || t->n >= 0xffffffffull || n_seq > t->n) { rc = -ERANGE; goto done; }
    const int hints = !getenv("FMD_WALK_NO_JUMP");
    w.seed_hints = hints;
    w.no_plain = getenv("FMD_WALK_NO_HOP") != NULL;
    w.timing = getenv("FMD_TIMING") != NULL;
    /* The skip list serves LONG walks (error-free or corrected reads: unitigs of 10^3 .. 10^7 reads, walked by one thread); on raw reads a
     * walk is one to three steps from its seed, the seeds' own hints cover those, and building it buys nothing.  Which it is shows in the
     * links: the steps in a row from 2048 evenly spaced rows, 64 at most each.  FMD_WALK_LONG=0 / 1 overrides the verdict (tests: both ways
     * on the same fixture). */
    int long_walks = 0;
    if (t->n) {
        const uint64_t ns = t->n < 2048 ? t->n : 2048;
        uint64_t k, tot = 0;
        for (k = 0; k < ns; ++k) {
            uint64_t cur = (uint64_t)((unsigned __int128)t->n * k / ns);
            int len = 0;
            while (len < 64) {
                const fmdh_wrec_t *h = &t->w[cur];
                if ((h->bits & FMDH_W_ST_MASK) != 0 || h->rbeg == 0xffff || h->n_nei != 1 || h->nxt == 0xffffffffu) break;
                cur = h->nxt; ++len;
            }
            tot += (uint64_t)len;
        }
        long_walks = tot >= 16 * ns;
        { const char *e = getenv("FMD_WALK_LONG"); if (e) long_walks = atoi(e) != 0; }
        if (w.timing) fprintf(stderr, "[M::%s] %.1f steps in a row from a sampled row (of 64 at most): %s\n", __func__, (double)tot / (double)ns,
                              long_walks ? "long walks, skip list" : "short walks, the seeds' hints only");
    }
    t_begin = wall_s();
    if (hints && long_walks) { w.have_far = build_far(t); if (w.have_far < 0) return -EDOM; }   /* 0: the plain chase */
    if (w.timing && w.have_far) fprintf(stderr, "[M::%s] skip list over the links: %.3f s; resident set now %.2f GB, peak so far %.2f GB\n", __func__, wall_s() - t_begin, fmdh_rss_gb(0), fmdh_rss_gb(1));
    const int seed_stages = !(getenv("FMD_WALK_SEED_STAGES") && atoi(getenv("FMD_WALK_SEED_STAGES")) == 0);
    {
        const int nt = walk_threads();
        if (nt > 1 && n_seq >= 4 * chunk_seeds()) { rc = walk_parallel(&w, cap_nei, out, nt); goto done; }
    }
    if ((rc = seedbuf_init(&b, cap_nei)) != 0) goto done;
    b_ok = 1;
    if ((rc = outq_open(&oq, out)) != 0) goto done;
    oq_open = 1;
    /* unitig_core with start = 0, step = 1 (unitig.c:333-334): seeds are the odd sequence ids */
    for (i = 1; i < n_seq; i += 2) {
        size_t wl = 0;
        int r1;
        if (hints) seed_hints(&w, i, seed_stages);
        r1 = walk_seed(&w, i, &b, &wl);
        if (r1 < 0) { rc = r1; goto done; }
        if (r1 == 1 && (rc = outq_put(&oq, b.o.s, wl)) != 0) goto done;
    }
done:
    if (oq_open) { const int e = outq_close(&oq); if (!rc) rc = e; }
    if (w.timing && w.n_hops + (uint64_t)(w.t_uni > 0)) fprintf(stderr, "[M::%s] walks of the sequential loop: %llu reads appended in %.3f s, turning the strings round %.3f s, record text %.3f s\n", __func__,
                                                                (unsigned long long)w.n_hops, w.t_uni, w.t_turn, w.t_text);
    if (w.timing) fprintf(stderr, "[M::%s] walk of %llu sequences: %.3f s; resident set now %.2f GB (%.2f GB of it in transparent huge pages), peak so far %.2f GB\n", __func__, (unsigned long long)n_seq, wall_s() - t_begin, fmdh_rss_gb(0), fmdh_thp_gb(), fmdh_rss_gb(1));
    free(w.used); free(w.bend); free(w.visited);
    if (b_ok) seedbuf_free(&b);
    return rc;
}
