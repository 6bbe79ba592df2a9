/* ovlp_table.c -- the per-sequence overlap table `fermi unitig` walks, computed on one or several GPUs.
 *
 * The reference's fm6_unitig starts n_threads workers, worker j seeding from ids i = j (mod n_threads)
 * (unitig.c:394-404, :333).  Here the index-bound part of every seed and every extension is a row of a table
 * (include/fmd_hip.h, fmd_ovlp_*), and GPU g -- its own replica of the index, its own host thread -- computes the
 * rows of the ids i = g (mod n_dev).  Rows arrive packed (fmd_ovlp_packed_batch) and stay in the shard they were
 * computed in: fmdh_table_row() addresses id i as row i / n_dev of shard i % n_dev. */
#define _GNU_SOURCE
#include <errno.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "fmd_host.h"

#define TABLE_CHUNK_SHIFT 22   /* rows per pipelined chunk: 4 M (26 GB of device work area at the default capacities) */

static double now_s(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec; }

static void shard_free(fmdh_ovlp_shard_t *s)
{
    if (s->chunk) {
        const size_t nc = (size_t)((s->n + ((uint64_t)1 << s->chunk_shift) - 1) >> s->chunk_shift);
        fmd_ovlp_packed_free(s->chunk, nc);
        free(s->chunk);
    }
    fmdh_big_free(s->rec); fmdh_big_free(s->off);
    memset(s, 0, sizeof(*s));
}

/* rows of the ids  ids[0..n)  or  first, first + step, ...  with capacities (max_len, max_nei) */
static int shard_fill(fmd_dev_t *d, fmdh_ovlp_shard_t *s, const uint64_t *ids, uint64_t first, uint64_t step, uint64_t n, int min_match,
                      uint32_t max_len, uint32_t max_nei, int with_cl)
{
    const size_t nc = (size_t)((n + ((uint64_t)1 << TABLE_CHUNK_SHIFT) - 1) >> TABLE_CHUNK_SHIFT);
    memset(s, 0, sizeof(*s));
    s->n = n; s->chunk_shift = TABLE_CHUNK_SHIFT; s->max_nei = max_nei; s->seq_stride = 2 * ((max_len + 3) / 4 * 4);
    s->rec = (fmd_ovlp_rec_t *)fmdh_big_alloc((n ? n : 1) * sizeof(fmd_ovlp_rec_t));
    s->off = (uint64_t *)fmdh_big_alloc((n ? n : 1) * 8);
    s->chunk = (uint8_t **)calloc(nc ? nc : 1, sizeof(uint8_t *));
    if (!s->rec || !s->off || !s->chunk) { shard_free(s); return FMD_E_NOMEM; }
    {
        const int rc = fmd_ovlp_packed_batch(d, ids, first, step, n, min_match, max_len, max_nei, with_cl, s->rec, s->off, s->chunk_shift, s->chunk);
        if (rc) { shard_free(s); return rc; }
    }
    return FMD_OK;
}

/* ------------------------------------------------------------------------------------------------ link pass */
typedef struct {
    fmdh_ovlp_table_t *t; uint64_t lo, hi; int phase;
    int force_exact;          /* A/B (FMD_CHECK_LEFT_EXACT, read once by the caller): every edge through fmd_ovlp_check_left_dev */
    uint64_t *und; uint64_t n_und, m_und; int rc;
} lk_t;
static inline fmd_ovlp_rec_t *row_rec_mut(fmdh_ovlp_table_t *t, uint64_t id)
{
    if (t->side_of && t->side_of[id] != 0xffffffffu) return &t->side.rec[t->side_of[id]];
    return &t->shard[id % (uint64_t)t->n_shards].rec[id / (uint64_t)t->n_shards];
}
/* link[i] and, where lfork decides it, rec[i].reserved of one row; 1 = the edge is left to the exact kernel */
static inline int link_row(fmdh_ovlp_table_t *t, uint64_t i, int force_exact)
{
    const fmdh_row_t x = fmdh_table_row(t, i);
    fmd_ovlp_rec_t *r = row_rec_mut(t, i);
    fmdh_link_t *l = &t->link[i];
    int d = 1;
    l->nxt = l->rev = 0xffffffffu;
    if (r->status != 0 || r->n_nei != 1 || r->rbeg < 0 || (r->flags & FMD_OVLP_F_OVERFLOW)) return 0;
    if (x.nei[0].x[0] < t->n) l->nxt = t->row_of[x.nei[0].x[0]];
    if (x.nei[0].x[1] < t->n) l->rev = t->row_of[x.nei[0].x[1]];
    if (r->reserved != 2) return 0;                          /* the exact answer is already there */
    if (l->rev != 0xffffffffu) d = fmd_lfork_decide(row_rec_mut(t, l->rev)->lfork, r->rbeg);
    if (force_exact) d = 1;
    if (d != 1) r->reserved = d < 0 ? 1 : 0;
    return d == 1;
}
static void *lk_main(void *p)
{
    lk_t *w = (lk_t *)p;
    fmdh_ovlp_table_t *t = w->t;
    uint64_t i;
    if (w->phase == 0) { /* row_of: the smallest id wins (identical reads share one interval) */
        for (i = w->lo; i < w->hi; ++i) {
            const fmd_ovlp_rec_t *r = row_rec_mut(t, i);
            if (r->status == 0 && !(r->flags & FMD_OVLP_F_OVERFLOW) && r->k[0] < t->n) {
                uint32_t *slot = &t->row_of[r->k[0]], cur = __atomic_load_n(slot, __ATOMIC_RELAXED);
                while ((uint32_t)i < cur && !__atomic_compare_exchange_n(slot, &cur, (uint32_t)i, 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
            }
        }
        return 0;
    }
    for (i = w->lo; i < w->hi; ++i)
        if (link_row(t, i, w->force_exact)) {
            if (w->n_und == w->m_und) {
                uint64_t m = w->m_und ? w->m_und << 1 : 1024, *q = (uint64_t *)realloc(w->und, m * 8);
                if (!q) { w->rc = -ENOMEM; return 0; }
                w->und = q; w->m_und = m;
            }
            w->und[w->n_und++] = i;
        }
    return 0;
}

int fmdh_ovlp_table_link(fmdh_ovlp_table_t *t, int n_threads, uint64_t **undecided, uint64_t *n_undecided)
{
    const uint64_t n = t->n;
    lk_t *w;
    pthread_t *tid;
    char *started;
    int k, phase, rc = 0;
    uint64_t tot = 0;
    if (undecided) *undecided = 0;
    const int force_exact = getenv("FMD_CHECK_LEFT_EXACT") != NULL;
    if (n_undecided) *n_undecided = 0;
    if (n >= 0xffffffffull) return -ERANGE;
    if (n_threads < 1) n_threads = 1;
    if ((uint64_t)n_threads > n / 4096 + 1) n_threads = (int)(n / 4096 + 1);
    fmdh_big_free(t->row_of); fmdh_big_free(t->link);
    t->row_of = (uint32_t *)fmdh_big_alloc((n ? n : 1) * 4);
    t->link = (fmdh_link_t *)fmdh_big_alloc((n ? n : 1) * sizeof(fmdh_link_t));
    w = (lk_t *)calloc((size_t)n_threads, sizeof(lk_t));
    tid = (pthread_t *)calloc((size_t)n_threads, sizeof(pthread_t));
    started = (char *)calloc((size_t)n_threads, 1);
    if (!t->row_of || !t->link || !w || !tid || !started) { rc = -ENOMEM; goto done; }
    memset(t->row_of, 0xff, n * 4);
    for (phase = 0; phase < 2; ++phase) {
        for (k = 0; k < n_threads; ++k) {
            w[k].t = t; w[k].lo = n * (uint64_t)k / (uint64_t)n_threads; w[k].hi = n * (uint64_t)(k + 1) / (uint64_t)n_threads; w[k].phase = phase; w[k].force_exact = force_exact;
            started[k] = k > 0 && pthread_create(&tid[k], 0, lk_main, &w[k]) == 0;
        }
        for (k = 0; k < n_threads; ++k) if (!started[k]) lk_main(&w[k]);   /* slice 0, and any slice whose thread could not be created */
        for (k = 1; k < n_threads; ++k) if (started[k]) pthread_join(tid[k], 0);
    }
    for (k = 0; k < n_threads; ++k) { if (w[k].rc) rc = w[k].rc; tot += w[k].n_und; }
    if (!rc && undecided && tot) {
        uint64_t *u = (uint64_t *)malloc(tot * 8), o = 0;
        if (!u) rc = -ENOMEM;
        else {
            for (k = 0; k < n_threads; ++k) { memcpy(u + o, w[k].und, w[k].n_und * 8); o += w[k].n_und; } /* slices are id ranges: already sorted */
            *undecided = u;
        }
    }
    if (!rc && n_undecided) *n_undecided = tot;
done:
    if (w) for (k = 0; k < n_threads; ++k) free(w[k].und);
    free(w); free(tid); free(started);
    return rc;
}

/* One GPU linked the whole table (fmd_ovlp_packed_table) while the rows that exceeded a capacity were still flagged; they are in the side
 * table now.  Nothing else changed, so nothing else is linked again: the side rows enter row_of (identical reads overflow together, so no
 * entry that is there loses to one of them), and the rows of `ids` -- the side rows themselves and what the device reported: edges it could
 * not decide and edges into a row that was not there -- get their links and verdicts the way fmdh_ovlp_table_link gives them to every row.
 * (0.6 s of host threads over 10^8 rows for 6*10^5 that changed.)  ids: ascending, may repeat. */
static int table_patch_links(fmdh_ovlp_table_t *t, const uint64_t *side_ids, uint64_t n_side, const uint64_t *dev_und, uint64_t n_dev_und, uint64_t **und_out, uint64_t *n_und_out)
{
    const int force_exact = getenv("FMD_CHECK_LEFT_EXACT") != NULL;
    uint64_t a = 0, b = 0, last = ~0ull, n_und = 0, m_und = 0, *und = 0;
    *und_out = 0; *n_und_out = 0;
    for (a = 0; a < n_side; ++a) {
        const fmd_ovlp_rec_t *r = &t->side.rec[a];
        if (r->status == 0 && !(r->flags & FMD_OVLP_F_OVERFLOW) && r->k[0] < t->n && (uint32_t)side_ids[a] < t->row_of[r->k[0]]) t->row_of[r->k[0]] = (uint32_t)side_ids[a];
    }
    for (a = 0, b = 0; a < n_side || b < n_dev_und;) {   /* the two lists merged */
        const uint64_t i = b >= n_dev_und || (a < n_side && side_ids[a] <= dev_und[b]) ? side_ids[a++] : dev_und[b++];
        if (i == last) continue;
        last = i;
        if (link_row(t, i, force_exact)) {
            if (n_und == m_und) { uint64_t m = m_und ? m_und << 1 : 1024, *q = (uint64_t *)realloc(und, m * 8); if (!q) { free(und); return -ENOMEM; } und = q; m_und = m; }
            und[n_und++] = i;
        }
    }
    *und_out = und; *n_und_out = n_und;
    return 0;
}

/* ------------------------------------------------------------------------------------------------ build */
typedef struct {
    const char *fmd_path; int device, g, n_dev, min_match; uint32_t max_len, max_nei;
    fmdh_ovlp_shard_t *shard; fmd_dev_t *dev; uint64_t n_seq; int rc; double t_load, t_rows;
    fmdh_ovlp_table_t *whole;          /* one GPU: the table itself, linked on the device (fmd_ovlp_packed_table) */
    uint64_t *und, n_und;
} job_t;

/* one GPU holds every row: the link pass runs there too */
static int table_fill_linked(fmd_dev_t *d, fmdh_ovlp_table_t *t, uint64_t n, int min_match, uint32_t max_len, uint32_t max_nei, uint64_t **und, uint64_t *n_und)
{
    fmdh_ovlp_shard_t *s = &t->shard[0];
    const size_t nc = (size_t)((n + ((uint64_t)1 << TABLE_CHUNK_SHIFT) - 1) >> TABLE_CHUNK_SHIFT);
    memset(s, 0, sizeof(*s));
    s->n = n; s->chunk_shift = TABLE_CHUNK_SHIFT; s->max_nei = max_nei; s->seq_stride = 2 * ((max_len + 3) / 4 * 4);
    s->rec = (fmd_ovlp_rec_t *)fmdh_big_alloc((n ? n : 1) * sizeof(fmd_ovlp_rec_t));
    s->off = (uint64_t *)fmdh_big_alloc((n ? n : 1) * 8);
    s->chunk = (uint8_t **)calloc(nc ? nc : 1, sizeof(uint8_t *));
    t->row_of = (uint32_t *)fmdh_big_alloc((n ? n : 1) * 4);
    t->link = (fmdh_link_t *)fmdh_big_alloc((n ? n : 1) * sizeof(fmdh_link_t));
    if (!s->rec || !s->off || !s->chunk || !t->row_of || !t->link) { shard_free(s); fmdh_big_free(t->row_of); fmdh_big_free(t->link); t->row_of = 0; t->link = 0; return FMD_E_NOMEM; }
    {
        const int rc = fmd_ovlp_packed_table(d, n, min_match, max_len, max_nei, s->rec, s->off, s->chunk_shift, s->chunk, t->row_of, (fmd_ovlp_link_t *)t->link, und, n_und);
        if (rc) { shard_free(s); fmdh_big_free(t->row_of); fmdh_big_free(t->link); t->row_of = 0; t->link = 0; return rc; }
    }
    return FMD_OK;
}

static void *job_main(void *p)
{
    job_t *j = (job_t *)p;
    fmd_info_t info;
    double t0 = now_s();
    if (!j->dev) j->rc = fmd_dev_open_file(j->device, j->fmd_path, &j->dev);   /* (fmdh_ovlp_table_build_dev: the caller's handle) */
    if (j->rc) return 0;
    fmd_dev_info(j->dev, &info);
    j->n_seq = info.mcnt[1];
    j->t_load = now_s() - t0; t0 = now_s();
    {   /* The capacity of the main pass follows the reads: the lengths of 8192 evenly spaced sequences (fm6_retrieve in bulk, a few ms), the
         * longest of them rounded up to 32 -- 128 for reads of up to 128 bases.  A sequence longer than that is flagged by the main pass and
         * computed in the overflow pass below, which costs a second look at the index: fine for stragglers, not for a third of the read set
         * (reads of 70-150 bases under a fixed 128: 28 % of the rows went round again, and the ladder doubled their capacities out of memory). */
        const uint64_t ns = j->n_seq < 8192 ? j->n_seq : 8192;
        uint64_t *sid = (uint64_t *)malloc((ns ? ns : 1) * 8), k;
        fmd_ovlp_rec_t *sr = (fmd_ovlp_rec_t *)malloc((ns ? ns : 1) * sizeof(fmd_ovlp_rec_t));
        if (sid && sr && ns) {
            int32_t mx = 0;
            for (k = 0; k < ns; ++k) sid[k] = (uint64_t)((unsigned __int128)j->n_seq * k / ns);
            if (fmd_seqinfo_batch(j->dev, (size_t)ns, sid, 1024, sr) == FMD_OK) {
                for (k = 0; k < ns; ++k) if (sr[k].len > mx) mx = sr[k].len;
                if (mx > 3000) mx = 3000;
                if ((uint32_t)mx > j->max_len) j->max_len = ((uint32_t)mx + 31) / 32 * 32;
            }
        }
        free(sid); free(sr);
    }
    if (j->whole && j->n_seq < 0xffffffffull && !getenv("FMD_HOST_LINK")) j->rc = table_fill_linked(j->dev, j->whole, j->n_seq, j->min_match, j->max_len, j->max_nei, &j->und, &j->n_und);
    else {
        const uint64_t n = j->n_seq > (uint64_t)j->g ? (j->n_seq - (uint64_t)j->g + (uint64_t)j->n_dev - 1) / (uint64_t)j->n_dev : 0;
        j->whole = 0;
        j->rc = shard_fill(j->dev, j->shard, 0, (uint64_t)j->g, (uint64_t)j->n_dev, n, j->min_match, j->max_len, j->max_nei, 0);
    }
    j->t_rows = now_s() - t0;
    if (j->g != 0) { fmd_dev_close(j->dev); j->dev = 0; }   /* replica 0 stays open for the overflow pass */
    return 0;
}

void fmdh_ovlp_table_free(fmdh_ovlp_table_t *t)
{
    int g;
    if (!t) return;
    for (g = 0; g < t->n_shards; ++g) shard_free(&t->shard[g]);
    free(t->shard);
    shard_free(&t->side);
    free(t->side_of); fmdh_big_free(t->row_of); fmdh_big_free(t->link);
    memset(t, 0, sizeof(*t));
}

static int cmp_u64(const void *a, const void *b) { const uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b; return x < y ? -1 : x > y; }
static int table_build_core(const char *fmd_path, fmd_dev_t *preopened, int n_dev, const int *devices, int min_match, fmdh_ovlp_table_t *t, uint64_t *n_seq_out);
int fmdh_ovlp_table_build(const char *fmd_path, int n_dev, const int *devices, int min_match, fmdh_ovlp_table_t *t, uint64_t *n_seq_out)
{
    return table_build_core(fmd_path, 0, n_dev, devices, min_match, t, n_seq_out);
}
/* the same table from an index that is already in a GPU's HBM (the in-memory API: no .fmd in between); the handle stays the caller's */
int fmdh_ovlp_table_build_dev(fmd_dev_t *dev, int min_match, fmdh_ovlp_table_t *t, uint64_t *n_seq_out)
{
    fmd_info_t info;
    int device;
    if (!dev || fmd_dev_info(dev, &info)) return 1;
    device = info.device;
    return table_build_core("(memory)", dev, 1, &device, min_match, t, n_seq_out);
}
static int table_build_core(const char *fmd_path, fmd_dev_t *preopened, int n_dev, const int *devices, int min_match, fmdh_ovlp_table_t *t, uint64_t *n_seq_out)
{
    const int timing = getenv("FMD_TIMING") != 0;
    uint32_t max_len = 128, longest = 0;   /* (max_len: raised by the jobs' probe of the read lengths) */
    const uint32_t max_nei = 4;
    job_t *jobs;
    pthread_t *tid;
    char *started;
    uint64_t *ids = 0, n_side = 0, n_seq, i;
    int g, rc = 0, too_long_hint = 0;
    double t0 = now_s();
    if (n_dev < 1 || !devices || !t) return 1;
    memset(t, 0, sizeof(*t));
    jobs = (job_t *)calloc((size_t)n_dev, sizeof(job_t));
    tid = (pthread_t *)calloc((size_t)n_dev, sizeof(pthread_t));
    started = (char *)calloc((size_t)n_dev, 1);
    t->shard = (fmdh_ovlp_shard_t *)calloc((size_t)n_dev, sizeof(fmdh_ovlp_shard_t));
    if (!jobs || !tid || !started || !t->shard) { free(jobs); free(tid); free(started); free(t->shard); t->shard = 0; return 1; }
    t->n_shards = n_dev;
    for (g = 0; g < n_dev; ++g) {
        job_t x = {fmd_path, devices[g], g, n_dev, min_match, max_len, max_nei, &t->shard[g], 0, 0, 0, 0, 0, n_dev == 1 ? t : 0, 0, 0};
        jobs[g] = x;
        if (g == 0 && preopened) jobs[g].dev = preopened;
        if (g > 0) started[g] = pthread_create(&tid[g], 0, job_main, &jobs[g]) == 0;
    }
    job_main(&jobs[0]);                                           /* shard 0 on the calling thread */
    for (g = 1; g < n_dev; ++g) { if (started[g]) pthread_join(tid[g], 0); else job_main(&jobs[g]); } /* no thread: do it here, afterwards */
    for (g = 0; g < n_dev; ++g) {
        if (jobs[g].rc) { fprintf(stderr, "[E::%s] GPU %d: %s\n", __func__, devices[g], fmd_strerror(jobs[g].rc)); rc = 1; }
        if (timing) fprintf(stderr, "[M::%s] GPU %d: index load + transcode %.3f s, %llu rows (GPU + copies) %.3f s\n", __func__, devices[g], jobs[g].t_load,
                            (unsigned long long)t->shard[g].n, jobs[g].t_rows);
    }
    if (rc) goto done;
    n_seq = jobs[0].n_seq;
    max_len = jobs[0].max_len;
    for (g = 1; g < n_dev; ++g) if (jobs[g].max_len > max_len) max_len = jobs[g].max_len;
    for (g = 1; g < n_dev; ++g) if (jobs[g].n_seq != n_seq) { fprintf(stderr, "[E::%s] the replicas disagree\n", __func__); rc = 1; goto done; }
    t->n = n_seq;
    if (n_seq_out) *n_seq_out = n_seq;
    if (n_seq >= 0xffffffffull) { fprintf(stderr, "[E::%s] %llu sequences: the walk's row map holds 32-bit ids\n", __func__, (unsigned long long)n_seq); rc = 1; goto done; }
    /* the rows that did not fit (longer sequences, more neighbours, longer lists): again, alone, with the capacities
     * doubled until they do -- on the GPU; nothing falls back to the CPU */
    {   /* one pass over the records: the flagged ids, and whether any of them is a sequence longer than max_len */
        uint64_t cap_ids = 0;
        int too_long = 0;
        for (g = 0; g < n_dev; ++g) {
            const fmd_ovlp_rec_t *r = t->shard[g].rec;
            const uint64_t m = t->shard[g].n;
            for (i = 0; i < m; ++i) if (r[i].flags & FMD_OVLP_F_OVERFLOW) {
                if (n_side == cap_ids) { cap_ids = cap_ids ? 2 * cap_ids : 1 << 16; ids = (uint64_t *)realloc(ids, cap_ids * 8); if (!ids) { rc = 1; goto done; } }
                ids[n_side++] = i * (uint64_t)n_dev + (uint64_t)g;
                too_long |= (uint32_t)r[i].len > max_len;
                if (r[i].len > 0 && (uint32_t)r[i].len > longest) longest = (uint32_t)r[i].len;
            }
        }
        too_long_hint = too_long;
    }
    if (n_side) {
        uint32_t s_len = max_len, s_nei = max_nei;
        int attempt;
        double t1 = now_s();
        t->side_of = (uint32_t *)malloc(n_seq * 4);
        if (!ids || !t->side_of) { rc = 1; goto done; }
        if (n_dev > 1) qsort(ids, n_side, 8, cmp_u64);   /* (one shard: ascending already) */
        for (attempt = 0;; ++attempt) {
            uint64_t n_over = 0;
            if (attempt == 12) { fprintf(stderr, "[E::%s] %llu rows still overflow at max_len %u, max_nei %u\n", __func__, (unsigned long long)n_side, s_len, s_nei); rc = 1; goto done; }
            /* what overflows in practice is the neighbour list of a strand in a fork-rich corner (more than max_nei irreducible overlaps): room for
             * four times as many at once, longer sequences / candidate lists only where a flagged record says so or the first attempt was not enough */
            if (attempt == 0) { s_nei *= 4; if (too_long_hint && longest > s_len) s_len = (longest + 31) / 32 * 32; }
            else { s_nei *= 2; s_len = (s_len + s_len / 2 + 31) / 32 * 32; }     /* (the candidate lists' capacity follows max_len: fmd_ovlp_list_cap) */
            if (s_len > 4000 + (uint32_t)min_match) s_len = 4000 + (uint32_t)min_match;   /* (lists of 4096 entries and more: not supported) */
            shard_free(&t->side);
            rc = shard_fill(jobs[0].dev, &t->side, ids, 0, 0, n_side, min_match, s_len, s_nei, 0);
            if (rc) { fprintf(stderr, "[E::%s] overflow pass: %s\n", __func__, fmd_strerror(rc)); rc = 1; goto done; }
            for (i = 0; i < n_side; ++i) n_over += (t->side.rec[i].flags & FMD_OVLP_F_OVERFLOW) != 0;
            if (n_over == 0) break;
        }
        memset(t->side_of, 0xff, n_seq * 4);
        for (i = 0; i < n_side; ++i) t->side_of[ids[i]] = (uint32_t)i;
        if (timing) fprintf(stderr, "[M::%s] %llu rows again with capacities %u / %u: %.3f s\n", __func__, (unsigned long long)n_side, s_len, s_nei, now_s() - t1);
    }
    /* check_left_simple (unitig.c:186-204) of every edge: decided on the host from the lfork of the neighbour's reverse
     * strand; the edges that field does not decide go through the exact kernel (fmd_ovlp_check_left_dev), alone */
    {
        uint64_t *und = 0, n_und = 0, k;
        double t1 = now_s();
        int nt = 16;
        { const char *e = getenv("FMD_HOST_THREADS"); if (e && atoi(e) > 0) nt = atoi(e); }
        if (jobs[0].whole && n_side == 0) { /* linked on the device already */
            und = jobs[0].und; n_und = jobs[0].n_und; jobs[0].und = 0;
            if (timing) fprintf(stderr, "[M::%s] link pass on the GPU (inside the table pass), %llu edges left to the exact kernel\n", __func__, (unsigned long long)n_und);
        } else if (jobs[0].whole && !getenv("FMD_HOST_RELINK")) {   /* linked on the device, and a few rows were replaced since: those, and what pointed at them */
            rc = table_patch_links(t, ids, n_side, jobs[0].und, jobs[0].n_und, &und, &n_und);
            fmd_host_free(jobs[0].und); jobs[0].und = 0;
            if (rc) { fprintf(stderr, "[E::%s] link patch: %s\n", __func__, strerror(-rc)); rc = 1; goto done; }
            if (timing) fprintf(stderr, "[M::%s] link pass on the GPU (inside the table pass) + %llu rows linked again here: %.3f s, %llu edges left to the exact kernel\n", __func__,
                                (unsigned long long)n_side, now_s() - t1, (unsigned long long)n_und);
        } else {
            fmd_host_free(jobs[0].und); jobs[0].und = 0;    /* rows were replaced by the overflow pass: link again, here */
            rc = fmdh_ovlp_table_link(t, nt, &und, &n_und);
            if (rc) { fprintf(stderr, "[E::%s] link pass: %s\n", __func__, strerror(-rc)); rc = 1; free(und); goto done; }
            if (timing) fprintf(stderr, "[M::%s] link pass (%d threads): %.3f s, %llu edges left to the exact kernel\n", __func__, nt, now_s() - t1, (unsigned long long)n_und);
        }
        if (n_und) {
            fmdh_ovlp_shard_t ex;
            uint32_t s_len = max_len > longest ? max_len : (longest + 31) / 32 * 32, s_nei = max_nei;
            int attempt;
            t1 = now_s();
            memset(&ex, 0, sizeof(ex));
            for (attempt = 0;; ++attempt) {   /* same capacity ladder as above: a row of the side table needs its capacities here too */
                uint64_t n_over = 0;
                rc = shard_fill(jobs[0].dev, &ex, und, 0, 0, n_und, min_match, s_len, s_nei, 1);
                if (rc) { fprintf(stderr, "[E::%s] exact check_left pass: %s\n", __func__, fmd_strerror(rc)); rc = 1; free(und); goto done; }
                for (k = 0; k < n_und; ++k) n_over += (ex.rec[k].flags & FMD_OVLP_F_OVERFLOW) != 0;
                if (n_over == 0) break;
                if (attempt == 12) {   /* records that still overflow are invalid: their verdicts must not reach the table */
                    fprintf(stderr, "[E::%s] exact check_left pass: %llu rows still overflow at max_len %u, max_nei %u\n", __func__, (unsigned long long)n_over, s_len, s_nei);
                    rc = 1; shard_free(&ex); free(und); goto done;
                }
                s_nei *= 2; s_len = (s_len + s_len / 2 + 31) / 32 * 32;
                if (s_len > 4000 + (uint32_t)min_match) s_len = 4000 + (uint32_t)min_match;
                shard_free(&ex);
            }
            for (k = 0; k < n_und; ++k) row_rec_mut(t, und[k])->reserved = ex.rec[k].reserved;
            shard_free(&ex);
            if (timing) fprintf(stderr, "[M::%s] exact check_left of %llu rows: %.3f s\n", __func__, (unsigned long long)n_und, now_s() - t1);
        }
        free(und);
    }
    if (timing) fprintf(stderr, "[M::%s] table of %llu sequences on %d GPU(s): %.3f s\n", __func__, (unsigned long long)n_seq, n_dev, now_s() - t0);
done:
    for (g = 0; g < n_dev; ++g) if (jobs[g].dev && jobs[g].dev != preopened) fmd_dev_close(jobs[g].dev);
    free(ids); free(jobs); free(tid); free(started);
    if (rc) fmdh_ovlp_table_free(t);
    return rc;
}
