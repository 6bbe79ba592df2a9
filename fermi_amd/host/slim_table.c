/* slim_table.c -- the table `fermi unitig`'s walk runs over, as it is kept in host memory.
 *
 * The reference keeps nothing per read: unitig1 (unitig.c:274-317) asks the index again at every seed and every step, and the
 * memory of fm6_unitig is three bitmaps of mcnt[1] bits (unitig.c:390-392).  Here the index work is done once per sequence id on
 * the GPU (include/fmd_hip.h, fmd_ovlp_*), and what the walk reads of a row must be held until the walk has passed -- rounds 1-4
 * held the packed rows as they left the GPU: record 64 B + offset 8 + neighbours 32 each + bases + row map 4 + links 8 (+ 32 of
 * hop[] and 4 of the skip list on long walks), 150-190 bytes per id, 190 GB for BASELINE's 7*10^8 reads.  The walk needs less:
 *
 *   w[id]   32 bytes, ONE line per plain step of the walk: the unique neighbour's id and the overlap with it, the id eight links on
 *           (prefetch hint), k[0] of the `$read$` interval (k[1] is k[0] of row id ^ 1, the other half of the same 64 bytes) and its size,
 *           the rank as its distance from k[0], rbeg, up to 24 appended bases, the verdict of check_left, the place of the variable part;
 *   var     per id: the length where it is not the table's common one; the neighbours where there are several, (x0, overlap) of 6 bytes;
 *           appended bases that did not fit the line; and the bases of the READ -- 2 bits each, once per read: row 2i+1 is the reverse complement of row 2i
 *           (cmd.c:457-469) and the walk already relies on that (unitig.c:310: the seed's other direction is the reverse strand's extension);
 *   a record with a field beyond those widths (an interval of more than 255 identical reads, a sequence of 65 536 bases or more)
 *           is kept whole in its variable part (W_BIG).
 *
 * Rows arrive in chunks as the GPU finishes them (fmdh_slim_add: the fat chunk is a staging buffer that is reused), rows that
 * exceeded a capacity are replaced when they have been computed again (fmdh_slim_replace), links and check_left verdicts come from
 * the device's link pass piece by piece (fmdh_slim_link_fold) or from a host pass over the slim rows (fmdh_slim_link_host: several
 * GPUs, and the tests' tables: until it has run, x[0] of a row's unique neighbour travels in w.nxt and rec.lfork in w.far, the fields it
 * fills), and fmdh_slim_finalize marks the plain steps.  44.5 bytes per id on 100-base reads: the line and a quarter of a byte per base
 * of every other row.
 */
#define _GNU_SOURCE
#include <errno.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "fmd_host.h"

/* ------------------------------------------------------------------------------------------------ a parallel for */
typedef void (*par_fn)(void *ctx, int tid, int nt);
typedef struct { par_fn fn; void *ctx; int tid, nt; } par_job_t;
static void *par_main(void *p) { par_job_t *j = (par_job_t *)p; j->fn(j->ctx, j->tid, j->nt); return 0; }
void fmdh_par_for(int nt, void (*fn)(void *ctx, int tid, int nt), void *ctx)
{
    pthread_t tid[64];
    par_job_t job[64];
    int started[64], k;
    if (nt < 1) nt = 1;
    if (nt > 64) nt = 64;
    for (k = 0; k < nt; ++k) { job[k].fn = fn; job[k].ctx = ctx; job[k].tid = k; job[k].nt = nt; }
    for (k = 1; k < nt; ++k) started[k] = pthread_create(&tid[k], 0, par_main, &job[k]) == 0;
    par_main(&job[0]);
    for (k = 1; k < nt; ++k) { if (started[k]) pthread_join(tid[k], 0); else par_main(&job[k]); }   /* no thread: here, afterwards */
}
/* The resident set of THIS program: VmRSS / VmHWM of /proc/self/status.  (getrusage's ru_maxrss is not that: at execve the kernel folds the high-water mark
 * of the address space that is being replaced into it, so a CLI started by fork + exec from a large process reports its parent's -- a 16 GB Python
 * process made every `unitig` it started "peak at 16.0 GB".) */
double fmdh_rss_gb(int peak)
{
    FILE *f = fopen("/proc/self/status", "r");
    const char *key = peak ? "VmHWM:" : "VmRSS:";
    char line[256];
    double kb = 0.0;
    if (!f) return 0.0;
    while (fgets(line, sizeof(line), f)) if (strncmp(line, key, 6) == 0) { kb = atof(line + 6); break; }
    fclose(f);
    return kb * 1024.0 / 1e9;
}
/* anonymous memory of this process that sits in transparent huge pages (AnonHugePages of /proc/self/smaps_rollup), in GB: fmd_table_alloc asks for them
 * (MADV_HUGEPAGE) and the walk's random reads of an 8 GB table miss the TLB on every step without */
double fmdh_thp_gb(void)
{
    FILE *f = fopen("/proc/self/smaps_rollup", "r");
    char line[256];
    double kb = -1.0;
    if (!f) return -1.0;
    while (fgets(line, sizeof(line), f)) if (strncmp(line, "AnonHugePages:", 14) == 0) { kb = atof(line + 14); break; }
    fclose(f);
    return kb < 0 ? -1.0 : kb * 1024.0 / 1e9;
}
int fmdh_host_threads(void)
{
    const char *e = getenv("FMD_HOST_THREADS");
    int nt = 16;
    if (e && atoi(e) > 0) nt = atoi(e);
    return nt > 64 ? 64 : nt;
}

/* ------------------------------------------------------------------------------------------------ layout */
static inline uint32_t ld32(const uint8_t *p) { uint32_t v; memcpy(&v, p, 4); return v; }
static inline uint16_t ld16(const uint8_t *p) { uint16_t v; memcpy(&v, p, 2); return v; }
static inline void st32(uint8_t *p, uint32_t v) { memcpy(p, &v, 4); }
static inline void st16(uint8_t *p, uint16_t v) { memcpy(p, &v, 2); }

fmdh_slim_t *fmdh_slim_new(uint64_t n, int n_shards, int host_link, uint32_t chunk_shift)
{
    fmdh_slim_t *s;
    if (n_shards < 1 || n >= 0xffffffffull || chunk_shift < 10 || chunk_shift > 26) return 0;
    s = (fmdh_slim_t *)calloc(1, sizeof(*s));
    if (!s) return 0;
    s->n = n; s->n_shards = n_shards; s->chunk_shift = chunk_shift; s->host_link = host_link;
    s->len0 = -1;                                                          /* (no row's length until one has arrived) */
    { const char *e = getenv("FMD_SLIM_BIG_K2"); s->big_k2 = e && atol(e) > 0 ? (uint64_t)atol(e) : 0xff; }   /* (tests lower the width of k[2] so that fixtures have W_BIG rows) */
    { const uint64_t per = (n + (uint64_t)n_shards - 1) / (uint64_t)n_shards; s->cps = (per >> s->chunk_shift) + 1; }
    s->w = (fmdh_wrec_t *)fmdh_big_alloc((n ? n : 1) * sizeof(fmdh_wrec_t));
    s->var = (uint8_t **)calloc((size_t)n_shards * s->cps, sizeof(uint8_t *));
    s->var_len = (uint64_t *)calloc((size_t)n_shards * s->cps, 8);
    pthread_mutex_init(&s->mu, 0);
    if (!s->w || !s->var || !s->var_len) { fmdh_slim_free(s); return 0; }
    return s;
}

static void slim_free_transients(fmdh_slim_t *s)
{
    fmdh_big_free(s->row_of);
    s->row_of = 0;
}

void fmdh_slim_free(fmdh_slim_t *s)
{
    size_t c;
    if (!s) return;
    if (s->var) for (c = 0; c < (size_t)s->n_shards * s->cps; ++c) fmdh_big_free(s->var[c]);
    free(s->var); free(s->var_len); free(s->xvar); free(s->und); free(s->und_rev);
    fmdh_big_free(s->w);
    slim_free_transients(s);
    pthread_mutex_destroy(&s->mu);
    free(s);
}

uint64_t fmdh_slim_bytes(const fmdh_slim_t *s)
{
    uint64_t b = s->n * sizeof(fmdh_wrec_t) + s->x_len;
    size_t c;
    for (c = 0; c < (size_t)s->n_shards * s->cps; ++c) b += s->var_len[c];
    return b;
}

/* ------------------------------------------------------------------------------------------------ one fat row -> one slim row */
typedef struct { const fmd_ovlp_rec_t *rec; const uint64_t *off; const uint8_t *var; uint32_t max_nei; } fat_t;   /* a staging chunk */
static inline fmdh_row_t fat_row(const fat_t *f, uint64_t j)
{
    fmdh_row_t x;
    x.rec = &f->rec[j]; x.var = f->var + f->off[j]; x.nei = (const fmd_intv_t *)x.var; x.max_nei = f->max_nei;
    return x;
}
static inline int row_status(const fmd_ovlp_rec_t *r)
{
    if (r->flags & FMD_OVLP_F_OVERFLOW) return FMDH_W_ST_INVALID;
    return r->status == 0 ? 0 : r->status == FMD_OVLP_SHORT ? FMDH_W_ST_SHORT : FMDH_W_ST_CONTAINED;
}
static int row_is_big(const fmd_ovlp_rec_t *r, int st, uint32_t nn, const fmd_intv_t *nei, uint64_t k2_limit)
{
    uint32_t k;
    if (st == FMDH_W_ST_INVALID) return 0;
    if (r->len < 0 || r->len > 0xffff || r->rank > 0xffffffffull) return 1;
    if (st == FMDH_W_ST_SHORT) return 0;                                  /* nothing else of a short row is ever read */
    if (r->k[0] > 0xffffffffull || r->k[1] > 0xffffffffull || r->k[2] > k2_limit) return 1;
    if (r->rank < r->k[0] || r->rank - r->k[0] > 0xff) return 1;                 /* (never: the read is one of the k[2] <= 255 of its interval) */
    if (st != 0) return 0;
    if (r->rbeg > 0xfffe || r->ext_len < 0 || r->ext_len > 0xff || r->n_nei < 0 || r->n_nei > 0xff) return 1;
    for (k = 0; k < nn; ++k) if (nei[k].x[0] > 0xffffffffull || nei[k].x[1] > 0xffffffffull || nei[k].info > 0xffffull) return 1;
    return 0;
}
/* bytes of the variable part; *inl = the appended bases fit the line (<= 24, all A/C/G/T) */
static inline uint32_t nei_block(uint32_t nn) { return nn == 1 ? 0 : nn * 6; }
static uint32_t row_var_bytes(const fmdh_row_t *x, int st, int big, int own_seed, int32_t len0, int *inl, char *tmp /* >= len + ext_len */)
{
    const fmd_ovlp_rec_t *r = x->rec;
    *inl = 1;
    if (big) {
        const uint32_t nn = fmd_ovlp_row_nei(r, x->max_nei), nb = st == 0 ? (uint32_t)r->len + (uint32_t)r->ext_len : 0;
        *inl = 0;
        return 64 + nn * 32 + (nb + 1) / 2;
    }
    if (st == FMDH_W_ST_INVALID) return 0;
    if (st != 0) return r->len != len0 ? 2 : 0;
    {
        const uint32_t nn = fmd_ovlp_row_nei(r, x->max_nei), len = (uint32_t)r->len, ext = (uint32_t)r->ext_len;
        uint32_t b = (r->len != len0 ? 2 : 0) + nei_block(nn), j;
        int seed_n = 0;
        if (ext > FMDH_W_EXT_INLINE) *inl = 0;
        if ((r->flags & FMD_OVLP_F_PACK4) && (ext || own_seed)) {              /* some base of the row is not A/C/G/T: which? */
            fmdh_row_bases(x, 0, len + ext, tmp);
            for (j = len; j < len + ext; ++j) if (tmp[j] < 1 || tmp[j] > 4) *inl = 0;
            for (j = 0; j < len; ++j) if (tmp[j] < 1 || tmp[j] > 4) seed_n = 1;
        }
        if (!*inl) b += (ext + 1) / 2;
        if (own_seed) b += seed_n ? (len + 1) / 2 : (len + 3) / 4;
        return b;
    }
}
/* writes w (all but nxt / far / the link bits) and the variable part at v; returns the bytes written.  host_link: what the host's link pass needs of a row
 * travels in the two fields that pass fills -- x[0] of the unique neighbour in w.nxt, rec.lfork in w.far (fmdh_slim_link_host) */
static uint32_t row_write(const fmdh_row_t *x, int st, int big, int own_seed, int host_link, int32_t len0, fmdh_wrec_t *w, uint8_t *v, char *tmp)
{
    const fmd_ovlp_rec_t *r = x->rec;
    const uint32_t nn = st == 0 ? fmd_ovlp_row_nei(r, x->max_nei) : 0;
    uint32_t k, b;
    memset(w, 0, sizeof(*w));
    w->nxt = w->far = 0xffffffffu;
    w->rbeg = 0xffff;
    w->bits = (uint8_t)st;
    if (st != FMDH_W_ST_INVALID && st != FMDH_W_ST_SHORT) { w->k0 = (uint32_t)r->k[0]; w->k2 = (uint8_t)(r->k[2] > 0xff ? 0xff : r->k[2]); if (!big) w->dr = (uint8_t)(r->rank - r->k[0]); }
    if (st == 0) {
        w->n_nei = (uint8_t)(r->n_nei > 0xff ? 0xff : r->n_nei < 0 ? 0 : r->n_nei);
        if (r->rbeg >= 0 && r->rbeg <= 0xfffe) w->rbeg = (uint16_t)r->rbeg;
        w->ext_len = (uint8_t)(r->ext_len > 0xff ? 0xff : r->ext_len < 0 ? 0 : r->ext_len);
        if (host_link) w->far = r->lfork;
    }
    if (big) {
        const uint32_t nb = st == 0 ? (uint32_t)r->len + (uint32_t)r->ext_len : 0;
        w->bits |= FMDH_W_BIG | FMDH_W_EXTVAR;
        memcpy(v, r, 64);
        memcpy(v + 64, x->nei, (size_t)nn * 32);
        if (nb) {
            uint8_t *q = v + 64 + nn * 32;
            fmdh_row_bases(x, 0, nb, tmp);
            memset(q, 0, (nb + 1) / 2);
            for (k = 0; k < nb; ++k) q[k >> 1] |= (uint8_t)((tmp[k] & 15) << (4 * (k & 1)));
        }
        return 64 + nn * 32 + (nb + 1) / 2;
    }
    if (st == FMDH_W_ST_INVALID) return 0;
    w->vfl = (uint8_t)((r->n_ovlp ? FMDH_V_HAS_OVLP : 0) | ((r->reserved > 2 ? 2 : r->reserved) << FMDH_V_RES_SHIFT));
    b = 0;
    if (r->len != len0) { w->vfl |= FMDH_V_LEN_VAR; st16(v, (uint16_t)(r->len < 0 ? 0 : r->len)); b = 2; }
    if (st != 0) return b;
    if (nn == 1) { w->ov = (uint16_t)x->nei[0].info; if (host_link) w->nxt = (uint32_t)x->nei[0].x[0]; }
    else for (k = 0; k < nn; ++k, b += 6) { st32(v + b, (uint32_t)x->nei[k].x[0]); st16(v + b + 4, (uint16_t)x->nei[k].info); }
    {
        const uint32_t len = (uint32_t)r->len, ext = (uint32_t)r->ext_len;
        int inl = ext <= FMDH_W_EXT_INLINE, seed_n = 0, have = 0;
        uint32_t j;
        if ((r->flags & FMD_OVLP_F_PACK4) && (ext || own_seed)) {
            fmdh_row_bases(x, 0, len + ext, tmp); have = 1;
            for (j = len; j < len + ext; ++j) if (tmp[j] < 1 || tmp[j] > 4) inl = 0;
            for (j = 0; j < len; ++j) if (tmp[j] < 1 || tmp[j] > 4) seed_n = 1;
        }
        if (ext) {
            if (!have) fmdh_row_bases(x, len, ext, tmp + len);
            if (inl) { for (j = 0; j < ext; ++j) w->ext[j >> 2] |= (uint8_t)(((tmp[len + j] - 1) & 3) << (2 * (j & 3))); }
            else {
                w->bits |= FMDH_W_EXTVAR;
                memset(v + b, 0, (ext + 1) / 2);
                for (j = 0; j < ext; ++j) v[b + (j >> 1)] |= (uint8_t)((tmp[len + j] & 15) << (4 * (j & 1)));
                b += (ext + 1) / 2;
            }
        }
        if (own_seed) {
            w->vfl |= FMDH_V_HAS_SEED;
            if (seed_n) {
                w->vfl |= FMDH_V_SEED_N;
                memset(v + b, 0, (len + 1) / 2);
                for (j = 0; j < len; ++j) v[b + (j >> 1)] |= (uint8_t)((tmp[j] & 15) << (4 * (j & 1)));
                b += (len + 1) / 2;
            } else if (!(r->flags & FMD_OVLP_F_PACK4)) {           /* the fat row holds the same 2-bit codes: copied as they are */
                memcpy(v + b, x->var + fmd_ovlp_row_nei(r, x->max_nei) * 32, (len + 3) / 4);
                b += (len + 3) / 4;
            } else {
                memset(v + b, 0, (len + 3) / 4);
                for (j = 0; j < len; ++j) v[b + (j >> 2)] |= (uint8_t)(((tmp[j] - 1) & 3) << (2 * (j & 3)));
                b += (len + 3) / 4;
            }
        }
    }
    return b;
}

/* ------------------------------------------------------------------------------------------------ rows of a chunk */
typedef struct { char *p; size_t m; } tmp_t;
static char *tmp_for(tmp_t *t, const fmd_ovlp_rec_t *r)
{
    const size_t need = (size_t)(r->len > 0 ? r->len : 0) + (size_t)(r->ext_len > 0 ? r->ext_len : 0) + 16;
    if (need > t->m) { char *q = (char *)realloc(t->p, 2 * need); if (!q) return 0; t->p = q; t->m = 2 * need; }
    return t->p;
}
typedef struct {
    fmdh_slim_t *s; fat_t f; uint64_t nr;
    int g; uint64_t chunk;              /* rows (chunk << shift) .. of shard g: id = g + n_shards * row; or */
    const uint64_t *ids;                /* explicit ids (rows replaced: every row carries its own bases; rows_of_a_peer: the even ones do, as in a chunk) */
    int even_seeds;
    uint64_t slice_bytes[64], slice_off[64]; uint32_t max_nei[64]; int rc[64];
    uint8_t *dst; uint64_t dst_unit;    /* voff = byte offset / dst_unit (rows start on multiples of it) */
    int phase;
} add_t;
static inline uint64_t add_id(const add_t *a, uint64_t j) { return a->ids ? a->ids[j] : (uint64_t)a->g + (uint64_t)a->s->n_shards * ((a->chunk << a->s->chunk_shift) + j); }
static void add_main(void *ctx, int tid, int nt)
{
    add_t *a = (add_t *)ctx;
    const uint64_t lo = a->nr * (uint64_t)tid / (uint64_t)nt, hi = a->nr * (uint64_t)(tid + 1) / (uint64_t)nt, unit = a->dst_unit;
    tmp_t t = {0, 0};
    uint64_t j, at = a->phase ? a->slice_off[tid] : 0;
    uint32_t mx = 0;
    for (j = lo; j < hi; ++j) {
        const fmdh_row_t x = fat_row(&a->f, j);
        const uint64_t id = add_id(a, j);
        const int st = row_status(x.rec), nn = st == 0 ? (int)fmd_ovlp_row_nei(x.rec, x.max_nei) : 0;
        const int big = row_is_big(x.rec, st, (uint32_t)nn, x.nei, a->s->big_k2), own_seed = st == 0 && ((a->ids && !a->even_seeds) || !(id & 1) || big);
        char *tmp = tmp_for(&t, x.rec);
        int inl;
        if (!tmp || id >= a->s->n) { a->rc[tid] = tmp ? -ERANGE : -ENOMEM; break; }
        if (st == 0 && x.rec->n_nei > 0 && (uint32_t)x.rec->n_nei > mx) mx = (uint32_t)x.rec->n_nei;
        if (!a->phase) at += (row_var_bytes(&x, st, big, own_seed, a->s->len0, &inl, tmp) + unit - 1) / unit * unit;
        else {
            fmdh_wrec_t *w = &a->s->w[id];
            const uint64_t start = at;
            if (start / unit > 0xffffffffull) { a->rc[tid] = -ERANGE; break; }
            at += (row_write(&x, st, big, own_seed, a->s->host_link && !a->s->linked, a->s->len0, w, a->dst + at, tmp) + unit - 1) / unit * unit;
            w->voff = (uint32_t)(start / unit);
            if (a->ids) w->bits |= FMDH_W_XVAR;
        }
    }
    free(t.p);
    if (!a->phase) { a->slice_bytes[tid] = at; a->max_nei[tid] = mx; }
}
/* sizes, then the rows; *bytes = the size of the variable parts, (*alloc)(ctx, bytes) = where they go */
static double wall_now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec; }
static int add_run(add_t *a, int nt, uint8_t *(*alloc)(add_t *a, uint64_t bytes))
{
    uint64_t tot = 0;
    int k;
    double t0, t1, t2, t3;
    if ((uint64_t)nt > a->nr / 1024 + 1) nt = (int)(a->nr / 1024 + 1);
    if (nt > 64) nt = 64;
    memset(a->rc, 0, sizeof(a->rc));
    pthread_mutex_lock(&a->s->mu);          /* the common length: that of the first row that arrives with one (any value is correct: it decides bytes only) */
    if (!a->s->len0_set) {
        uint64_t j;
        for (j = 0; j < a->nr; ++j) if (row_status(&a->f.rec[j]) != FMDH_W_ST_INVALID && a->f.rec[j].len > 0 && a->f.rec[j].len <= 0xffff) { a->s->len0 = a->f.rec[j].len; a->s->len0_set = 1; break; }
    }
    pthread_mutex_unlock(&a->s->mu);
    a->phase = 0;
    t0 = wall_now();
    fmdh_par_for(nt, add_main, a);
    for (k = 0; k < nt; ++k) { if (a->rc[k]) return a->rc[k]; a->slice_off[k] = tot; tot += a->slice_bytes[k]; }
    t1 = wall_now();
    a->dst = alloc(a, tot);
    if (!a->dst) return -ENOMEM;
    a->phase = 1;
    t2 = wall_now();
    fmdh_par_for(nt, add_main, a);
    t3 = wall_now();
    if (a->g == 0) { a->s->t_add[0] += t1 - t0; a->s->t_add[1] += t2 - t1; a->s->t_add[2] += t3 - t2; a->s->t_add[3] += 1; }   /* (shard 0's calls: one writer) */
    for (k = 0; k < nt; ++k) if (a->rc[k]) return a->rc[k];
    pthread_mutex_lock(&a->s->mu);
    for (k = 0; k < nt; ++k) if (a->max_nei[k] > a->s->max_nei) a->s->max_nei = a->max_nei[k];
    pthread_mutex_unlock(&a->s->mu);
    return 0;
}
static uint8_t *alloc_chunk(add_t *a, uint64_t bytes)
{
    fmdh_slim_t *s = a->s;
    const size_t c = (size_t)a->g * s->cps + a->chunk;
    fmdh_big_free(s->var[c]);
    s->var[c] = (uint8_t *)fmdh_big_alloc(bytes + 8);
    s->var_len[c] = bytes;
    return s->var[c];
}
static uint8_t *alloc_x(add_t *a, uint64_t bytes)   /* behind what xvar holds; rows address it by offset, so growing it moves nothing they hold */
{
    fmdh_slim_t *s = a->s;
    int k;
    if (s->x_len + bytes + 8 > s->x_cap) {
        const uint64_t m = (s->x_len + bytes + 8) * 3 / 2 + 4096;
        uint8_t *q = (uint8_t *)realloc(s->xvar, m);
        if (!q) return 0;
        s->xvar = q; s->x_cap = m;
    }
    for (k = 0; k < 64; ++k) a->slice_off[k] += s->x_len;
    s->x_len += bytes;
    return s->xvar;
}

int fmdh_slim_add(fmdh_slim_t *s, int g, uint64_t chunk, const fmd_ovlp_rec_t *rec, const uint64_t *off, const uint8_t *var, uint32_t max_nei, uint64_t nr, int n_threads)
{
    add_t a;
    if (!s || g < 0 || g >= s->n_shards || chunk >= s->cps || nr > ((uint64_t)1 << s->chunk_shift)) return -EINVAL;
    if (nr == 0) return 0;
    memset(&a, 0, sizeof(a));
    a.s = s; a.f.rec = rec; a.f.off = off; a.f.var = var; a.f.max_nei = max_nei; a.nr = nr; a.g = g; a.chunk = chunk; a.dst_unit = 1;
    return add_run(&a, n_threads, alloc_chunk);
}

/* rows that were computed again (ids ascending or not; each row keeps its own bases); the old variable parts stay where they are, unused */
int fmdh_slim_replace(fmdh_slim_t *s, const uint64_t *ids, const fmd_ovlp_rec_t *rec, const uint64_t *off, const uint8_t *var, uint32_t max_nei, uint64_t n, int n_threads)
{
    add_t a;
    if (!s || (n && !ids)) return -EINVAL;
    if (n == 0) return 0;
    memset(&a, 0, sizeof(a));
    a.s = s; a.f.rec = rec; a.f.off = off; a.f.var = var; a.f.max_nei = max_nei; a.nr = n; a.ids = ids; a.dst_unit = 8;
    s->x_len = (s->x_len + 7) & ~(uint64_t)7;
    return add_run(&a, n_threads, alloc_x);
}

/* rows in ANY order of ids, each once (a piece of a peer of an N-process job, fmd_ovlp_dist_cfg_t.row_sink: the rows of one piece are neighbours in the
 * sorted order of their keys, not in id): as fmdh_slim_add keeps them -- the bases with the even row of a read only -- in the growing area fmdh_slim_replace uses */
int fmdh_slim_add_ids(fmdh_slim_t *s, const uint64_t *ids, const fmd_ovlp_rec_t *rec, const uint64_t *off, const uint8_t *var, uint32_t max_nei, uint64_t n, int n_threads)
{
    add_t a;
    if (!s || (n && !ids)) return -EINVAL;
    if (n == 0) return 0;
    memset(&a, 0, sizeof(a));
    a.s = s; a.f.rec = rec; a.f.off = off; a.f.var = var; a.f.max_nei = max_nei; a.nr = n; a.ids = ids; a.dst_unit = 8; a.even_seeds = 1;
    s->x_len = (s->x_len + 7) & ~(uint64_t)7;
    return add_run(&a, n_threads, alloc_x);
}

/* ------------------------------------------------------------------------------------------------ links and check_left */
static int und_push(fmdh_slim_t *s, uint64_t id, uint32_t rev)
{
    if (s->n_und == s->m_und) {
        const uint64_t m = s->m_und ? 2 * s->m_und : 1024;
        uint64_t *a = (uint64_t *)realloc(s->und, m * 8);
        uint32_t *b;
        if (!a) return -ENOMEM;
        s->und = a;
        b = (uint32_t *)realloc(s->und_rev, m * 4);
        if (!b) return -ENOMEM;
        s->und_rev = b; s->m_und = m;
    }
    s->und[s->n_und] = id; s->und_rev[s->n_und++] = rev;
    return 0;
}
/* the verdict of check_left (unitig.c:206-225) for the edge of row id, whose check_left_simple said `res` (0 / 1): a potential backward
 * bifurcation stands if the reverse strand of the neighbour has more than one irreducible overlap itself */
static inline void set_verdict(fmdh_slim_t *s, uint64_t id, int res, uint32_t rev)
{
    fmdh_wrec_t *w = &s->w[id];
    uint8_t b = __atomic_load_n(&w->bits, __ATOMIC_RELAXED) & (uint8_t)~(FMDH_W_CL | FMDH_W_UNDEC);
    if (res == 2) b |= FMDH_W_UNDEC;
    else if (res && (rev == 0xffffffffu || s->w[rev].n_nei > 1)) b |= FMDH_W_CL;
    __atomic_store_n(&w->bits, b, __ATOMIC_RELAXED);     /* one store of the final byte: other threads read this row's BIG / status bits meanwhile */
}
static inline int has_edge(const fmdh_wrec_t *w) { return (w->bits & FMDH_W_ST_MASK) == 0 && w->n_nei == 1 && w->rbeg != 0xffff; }

/* the device's link pass (fmd_ovlp_link_dev) over rows first .. first + n: the neighbour's row, and check_left_simple where lfork decided it */
int fmdh_slim_link_fold(fmdh_slim_t *s, uint64_t first, uint64_t n, const fmdh_link_t *link, const uint8_t *reserved)
{
    uint64_t j;
    if (!s || first + n > s->n) return -EINVAL;
    for (j = 0; j < n; ++j) {
        const uint64_t id = first + j;
        fmdh_wrec_t *w = &s->w[id];
        if (!has_edge(w)) continue;
        w->nxt = link[j].nxt;
        if (w->nxt == 0xffffffffu) { w->bits |= FMDH_W_UNDEC; continue; }    /* its neighbour has no row: the table is incomplete (the walk fails if it gets there) */
        set_verdict(s, id, reserved[j] > 1 ? 2 : reserved[j], link[j].rev);
        if ((w->bits & FMDH_W_UNDEC) && und_push(s, id, link[j].rev)) return -ENOMEM;
    }
    return 0;
}

typedef struct { fmdh_slim_t *s; int phase, force_exact; uint64_t *und[64]; uint32_t *rev[64]; uint64_t n_und[64], m_und[64]; int rc[64]; } lk_t;
static void lk_main(void *ctx, int tid, int nt)
{
    lk_t *L = (lk_t *)ctx;
    fmdh_slim_t *s = L->s;
    const uint64_t lo = s->n * (uint64_t)tid / (uint64_t)nt, hi = s->n * (uint64_t)(tid + 1) / (uint64_t)nt;
    uint64_t i;
    if (L->phase == 0) {   /* row_of: the smallest id wins (identical reads share one interval) */
        for (i = lo; i < hi; ++i) {
            const fmdh_wrec_t *w = &s->w[i];
            uint64_t k0 = w->k0;
            if ((w->bits & FMDH_W_ST_MASK) != 0) continue;
            if (w->bits & FMDH_W_BIG) { fmd_ovlp_rec_t r; memcpy(&r, fmdh_slim_var(s, i), 64); k0 = r.k[0]; }
            if (k0 < s->n) {
                uint32_t *slot = &s->row_of[k0], cur = __atomic_load_n(slot, __ATOMIC_RELAXED);
                while ((uint32_t)i < cur && !__atomic_compare_exchange_n(slot, &cur, (uint32_t)i, 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
            }
        }
        return;
    }
    for (i = lo; i < hi; ++i) {
        fmdh_wrec_t *w = &s->w[i];
        uint64_t x0;
        uint32_t rev;
        int res, reserved, rbeg;
        if (!has_edge(w)) continue;
        /* x[0] of `$neighbour$`: where row_write left it (w.nxt), or in the record a W_BIG row keeps.  The neighbour's other strand is the row beside the
         * neighbour's: x[1] of `$neighbour$` is k[0] of that strand (fmd_host.h), and where several identical reads share the interval (row_of = the
         * smallest id) the smallest of the other strands' ids is its partner -- but for a read that is its own reverse complement, whose two rows
         * are one record twice. */
        if (w->bits & FMDH_W_BIG) {
            fmd_ovlp_rec_t r; fmd_intv_t e;
            const uint8_t *p = fmdh_slim_var(s, i);
            memcpy(&r, p, 64); memcpy(&e, p + 64, 32);
            x0 = e.x[0]; reserved = r.reserved > 2 ? 2 : r.reserved; rbeg = r.rbeg;
        } else { x0 = w->nxt; reserved = (w->vfl >> FMDH_V_RES_SHIFT) & 3; rbeg = w->rbeg; }
        w->nxt = x0 < s->n ? s->row_of[x0] : 0xffffffffu;
        if (w->nxt == 0xffffffffu) { __atomic_store_n(&w->bits, (uint8_t)(w->bits | FMDH_W_UNDEC), __ATOMIC_RELAXED); continue; }
        rev = ((uint64_t)w->nxt ^ 1) < s->n ? (w->nxt ^ 1u) : 0xffffffffu;
        res = reserved;
        if (res == 2 || L->force_exact) {     /* check_left_simple was not run on this row: the rounds of the neighbour's reverse strand decide it (include/fmd_hip.h) */
            int d = 1;
            if (rev != 0xffffffffu && !L->force_exact) {
                const fmdh_wrec_t *q = &s->w[rev];
                uint16_t lf = (uint16_t)q->far;
                const uint8_t qb = __atomic_load_n(&q->bits, __ATOMIC_RELAXED);   /* (its owner may be storing its verdict bits: BIG and the status never change here) */
                if (qb & FMDH_W_BIG) { fmd_ovlp_rec_t r; memcpy(&r, fmdh_slim_var(s, rev), 64); lf = r.lfork; }
                if ((qb & FMDH_W_ST_MASK) == 0) d = fmd_lfork_decide(lf, rbeg);
            }
            res = d == 1 ? 2 : d < 0;
            if (L->force_exact && reserved != 2) res = reserved;   /* (the exact answer is there already) */
        }
        set_verdict(s, i, res, rev);
        if (w->bits & FMDH_W_UNDEC) {
            if (L->n_und[tid] == L->m_und[tid]) {
                const uint64_t m = L->m_und[tid] ? 2 * L->m_und[tid] : 1024;
                uint64_t *a = (uint64_t *)realloc(L->und[tid], m * 8);
                uint32_t *b = a ? (uint32_t *)realloc(L->rev[tid], m * 4) : 0;
                if (a) L->und[tid] = a;
                if (!a || !b) { L->rc[tid] = -ENOMEM; return; }
                L->rev[tid] = b; L->m_und[tid] = m;
            }
            L->und[tid][L->n_und[tid]] = i; L->rev[tid][L->n_und[tid]++] = rev;
        }
    }
}
/* One parallel pass over complete slim rows: row map, links, and check_left from the records' own `reserved` where it is 0 / 1 and from the
 * lfork of the neighbour's reverse strand where it is 2 (needs the table made with host_link = 1).  The rows left undecided are listed
 * (fmdh_slim_undecided) for fmd_ovlp_check_left_dev; fmdh_slim_set_reserved takes the answers. */
int fmdh_slim_link_host(fmdh_slim_t *s, int n_threads)
{
    lk_t *L;
    int k, rc = 0, nt = n_threads;
    if (!s) return -EINVAL;
    if (nt < 1) nt = 1;
    if (nt > 64) nt = 64;
    if ((uint64_t)nt > s->n / 4096 + 1) nt = (int)(s->n / 4096 + 1);
    if (!s->host_link || s->linked) return -EINVAL;   /* (x[0] and lfork travel in w.nxt / w.far of a host_link table until this pass has run: once) */
    L = (lk_t *)calloc(1, sizeof(lk_t));
    fmdh_big_free(s->row_of);
    s->row_of = (uint32_t *)fmdh_big_alloc((s->n ? s->n : 1) * 4);
    if (!L || !s->row_of) { free(L); return -ENOMEM; }
    memset(s->row_of, 0xff, s->n * 4);
    L->s = s; L->force_exact = getenv("FMD_CHECK_LEFT_EXACT") != NULL;
    s->n_und = 0;
    for (L->phase = 0; L->phase < 2; ++L->phase) fmdh_par_for(nt, lk_main, L);
    for (k = 0; k < nt && !rc; ++k) {
        uint64_t j;
        if (L->rc[k]) rc = L->rc[k];
        for (j = 0; j < L->n_und[k] && !rc; ++j) rc = und_push(s, L->und[k][j], L->rev[k][j]);     /* slices are id ranges: ascending */
    }
    for (k = 0; k < 64; ++k) { free(L->und[k]); free(L->rev[k]); }
    free(L);
    s->linked = 1;
    return rc;
}
void fmdh_slim_undecided(const fmdh_slim_t *s, const uint64_t **ids, uint64_t *n) { *ids = s->und; *n = s->n_und; }

/* check_left_simple of rows the link pass left open (ids ascending, as fmdh_slim_undecided lists them; vals 0 / 1) */
int fmdh_slim_set_reserved(fmdh_slim_t *s, const uint64_t *ids, const uint16_t *vals, uint64_t n)
{
    uint64_t j, lo = 0;
    for (j = 0; j < n; ++j) {
        uint64_t a = lo, b = s->n_und;
        while (a < b) { const uint64_t m = (a + b) / 2; if (s->und[m] < ids[j]) a = m + 1; else b = m; }
        if (a == s->n_und || s->und[a] != ids[j]) return -EINVAL;
        set_verdict(s, ids[j], vals[j] > 1 ? 2 : (int)vals[j], s->und_rev[a]);
        lo = a;
    }
    return 0;
}

/* plain steps (what the walk takes from one line: unitig_walk.c), and the transients go */
static void fin_main(void *ctx, int tid, int nt)
{
    fmdh_slim_t *s = (fmdh_slim_t *)ctx;
    const uint64_t lo = s->n * (uint64_t)tid / (uint64_t)nt, hi = s->n * (uint64_t)(tid + 1) / (uint64_t)nt;
    uint64_t i;
    for (i = lo; i < hi; ++i) {
        fmdh_wrec_t *w = &s->w[i];
        /* (another thread may be reading this row's bits -- its BIG bit and status, which this pass never changes -- as the neighbour of one of ITS
         * rows: the byte is written once, whole, and read through relaxed atomics) */
        const uint8_t b0 = __atomic_load_n(&w->bits, __ATOMIC_RELAXED) & (uint8_t)~FMDH_W_PLAIN;
        uint8_t b1 = b0;
        w->far = 0xffffffffu;                                  /* (lfork of a host_link table until now; the walk builds the skip list here) */
        if (i + 8 < hi && s->w[i + 8].nxt != 0xffffffffu) __builtin_prefetch(&s->w[s->w[i + 8].nxt]);
        if (has_edge(w) && w->nxt != 0xffffffffu && !(b0 & (FMDH_W_EXTVAR | FMDH_W_UNDEC | FMDH_W_BIG))) {
            /* the plain step takes k[1] of the neighbour from k0 of the neighbour's OTHER strand (unitig_walk.c): that row must exist and must have
             * been written with its interval (a short or flagged row carries k0 = 0) */
            const uint64_t nx = w->nxt, ot = nx ^ 1;
            if (ot < s->n && !(__atomic_load_n(&s->w[nx].bits, __ATOMIC_RELAXED) & FMDH_W_BIG)) {
                const uint8_t bo = __atomic_load_n(&s->w[ot].bits, __ATOMIC_RELAXED);
                const unsigned sto = bo & FMDH_W_ST_MASK;
                if (!(bo & FMDH_W_BIG) && (sto == 0 || sto == FMDH_W_ST_CONTAINED)) b1 |= FMDH_W_PLAIN;
            }
        }
        __atomic_store_n(&w->bits, b1, __ATOMIC_RELAXED);
    }
}
int fmdh_slim_finalize(fmdh_slim_t *s, int n_threads)
{
    int nt = n_threads;
    if (!s) return -EINVAL;
    if ((uint64_t)nt > s->n / 4096 + 1) nt = (int)(s->n / 4096 + 1);
    fmdh_par_for(nt, fin_main, s);
    slim_free_transients(s);
    return 0;
}

/* ------------------------------------------------------------------------------------------------ from a table of packed rows */
/* The tests' tables and callers that hold packed rows already (fmd_ovlp_packed_batch): every shard in pieces of one slim chunk.  A table that
 * was linked (fmdh_ovlp_table_link: row_of, link, rec.reserved) keeps those links; otherwise the slim rows are linked here. */
int fmdh_slim_from_table(const fmdh_ovlp_table_t *t, int n_threads, fmdh_slim_t **out)
{
    fmdh_slim_t *s;
    int g, rc = 0;
    uint64_t i;
    *out = 0;
    if (!t || t->n >= 0xffffffffull) return -ERANGE;
    s = fmdh_slim_new(t->n, t->n_shards, t->link == 0, FMDH_SLIM_CHUNK_SHIFT);
    if (!s) return -ENOMEM;
    for (g = 0; g < t->n_shards && !rc; ++g) {
        const fmdh_ovlp_shard_t *sh = &t->shard[g];
        uint64_t c, r0;
        for (c = 0, r0 = 0; r0 < sh->n && !rc; ++c, r0 += (uint64_t)1 << s->chunk_shift) {
            const uint64_t nr = sh->n - r0 < ((uint64_t)1 << s->chunk_shift) ? sh->n - r0 : (uint64_t)1 << s->chunk_shift;
            /* rows r0 .. r0 + nr of a shard lie in ONE chunk of the packed table or the caller made the chunks smaller than ours: row by row then */
            if ((r0 >> sh->chunk_shift) == ((r0 + nr - 1) >> sh->chunk_shift))
                rc = fmdh_slim_add(s, g, c, sh->rec + r0, sh->off + r0, sh->chunk[r0 >> sh->chunk_shift], sh->max_nei, nr, n_threads);
            else rc = -EINVAL;
        }
    }
    if (!rc && t->side_of) {   /* the rows of the side table replace what the main pass flagged */
        uint64_t n_side = 0, k = 0, *ids;
        for (i = 0; i < t->n; ++i) n_side += t->side_of[i] != 0xffffffffu;
        ids = (uint64_t *)malloc((n_side ? n_side : 1) * 8);
        if (!ids) rc = -ENOMEM;
        else {
            for (i = 0; i < t->n; ++i) if (t->side_of[i] != 0xffffffffu) { if (t->side_of[i] >= n_side) rc = -EINVAL; else ids[t->side_of[i]] = i; }
            for (k = 0; k < n_side && !rc; k += (uint64_t)1 << t->side.chunk_shift) {   /* (piece by piece: offsets count inside a chunk of the packed table) */
                const uint64_t nr = n_side - k < ((uint64_t)1 << t->side.chunk_shift) ? n_side - k : (uint64_t)1 << t->side.chunk_shift;
                rc = fmdh_slim_replace(s, ids + k, t->side.rec + k, t->side.off + k, t->side.chunk[k >> t->side.chunk_shift], t->side.max_nei, nr, n_threads);
            }
            free(ids);
        }
    }
    if (!rc) {
        if (t->link) {
            uint8_t *res = (uint8_t *)malloc(t->n ? t->n : 1);
            if (!res) rc = -ENOMEM;
            else {
                for (i = 0; i < t->n; ++i) { fmdh_rowv_t v; fmdh_slim_row(s, i, &v); res[i] = (uint8_t)v.reserved; }
                rc = fmdh_slim_link_fold(s, 0, t->n, t->link, res);
                free(res);
            }
        } else rc = fmdh_slim_link_host(s, n_threads);
    }
    if (!rc) rc = fmdh_slim_finalize(s, n_threads);
    if (rc) { fmdh_slim_free(s); return rc; }
    *out = s;
    return 0;
}

/* what a table of packed rows becomes: out[0] = bytes held, [1] = W_BIG rows, [2] = rows whose appended bases are in the variable part, [3] = plain steps,
 * [4] = rows that keep their own sequence, [5] = edges left undecided (tests/test_host_formats.py) */
int fmdh_slim_stats(const fmdh_ovlp_table_t *t, uint64_t out[6])
{
    fmdh_slim_t *s = 0;
    uint64_t i;
    const int rc = fmdh_slim_from_table(t, 2, &s);
    if (rc) return rc;
    memset(out, 0, 6 * sizeof(uint64_t));
    out[0] = fmdh_slim_bytes(s);
    for (i = 0; i < s->n; ++i) {
        fmdh_rowv_t v;
        fmdh_slim_row(s, i, &v);
        out[1] += (s->w[i].bits & FMDH_W_BIG) != 0; out[2] += (s->w[i].bits & FMDH_W_EXTVAR) != 0 && !(s->w[i].bits & FMDH_W_BIG); out[3] += (s->w[i].bits & FMDH_W_PLAIN) != 0;
        out[4] += v.status == 0 && (v.vflags & FMDH_V_HAS_SEED) != 0; out[5] += (s->w[i].bits & FMDH_W_UNDEC) != 0;
    }
    fmdh_slim_free(s);
    return 0;
}
