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

/* ------------------------------------------------------------------------------------------------ build */
/* The table `unitig` walks, built on n_dev GPUs and kept SLIM (slim_table.c): the packed chunks are staging buffers of the library
 * (fmd_ovlp_tabjob_rows / fmd_ovlp_packed_stream) and every chunk is folded into 32 bytes per row + a short variable part as it
 * arrives.  One GPU: the device links the table (fmd_ovlp_tabjob_link) after the rows that exceeded a capacity have been computed
 * again and patched into its copy; several GPUs: GPU g streams the rows of ids i = g (mod n_dev), host threads link the slim rows. */
#define STREAM_CHUNK_SHIFT 20   /* rows per streamed chunk: two pinned staging sets of ~340 MB at the default capacities */
typedef struct {
    const char *fmd_path; int device, g, n_dev, min_match; uint32_t max_len, max_nei;
    fmd_dev_t *dev; uint64_t n_seq; int rc; double t_load, t_rows;
    fmdh_slim_t *slim; int conv_threads, one_gpu_job;
    fmd_ovlp_tabjob_t *tabjob;
    uint64_t *flagged, n_flagged, m_flagged; uint32_t longest; int too_long;   /* what the sink saw: rows that exceeded a capacity */
    int sink_rc;
} job_t;

static int rows_sink(void *ctx, uint64_t first_row, size_t n_rows, const fmd_ovlp_rec_t *rec, const uint64_t *off, const uint8_t *var, uint64_t var_bytes)
{
    job_t *j = (job_t *)ctx;
    size_t k;
    int rc;
    (void)var_bytes;
    if (first_row & (((uint64_t)1 << STREAM_CHUNK_SHIFT) - 1)) return FMD_E_ARG;
    rc = fmdh_slim_add(j->slim, j->g, first_row >> STREAM_CHUNK_SHIFT, rec, off, var, j->max_nei, n_rows, j->conv_threads);
    if (j->g == 0 && getenv("FMD_TIMING") && ((first_row >> STREAM_CHUNK_SHIFT) & 31) == 0)    /* where the resident set stands as the rows arrive */
        fprintf(stderr, "[M::%s] rows %llu ..: resident set %.2f GB (peak so far %.2f)\n", __func__, (unsigned long long)first_row, fmdh_rss_gb(0), fmdh_rss_gb(1));
    if (rc) { j->sink_rc = rc; return FMD_E_NOMEM; }
    for (k = 0; k < n_rows; ++k) if (rec[k].flags & FMD_OVLP_F_OVERFLOW) {
        if (j->n_flagged == j->m_flagged) {
            const uint64_t m = j->m_flagged ? 2 * j->m_flagged : 1 << 16;
            uint64_t *q = (uint64_t *)realloc(j->flagged, m * 8);
            if (!q) { j->sink_rc = -ENOMEM; return FMD_E_NOMEM; }
            j->flagged = q; j->m_flagged = m;
        }
        j->flagged[j->n_flagged++] = (uint64_t)j->g + (uint64_t)j->n_dev * (first_row + k);
        j->too_long |= (uint32_t)rec[k].len > j->max_len;
        if (rec[k].len > 0 && (uint32_t)rec[k].len > j->longest) j->longest = (uint32_t)rec[k].len;
    }
    return 0;
}
static int links_sink(void *ctx, uint64_t first_row, size_t n_rows, const fmd_ovlp_link_t *link, const uint8_t *reserved)
{
    return fmdh_slim_link_fold((fmdh_slim_t *)ctx, first_row, n_rows, (const fmdh_link_t *)link, reserved) ? FMD_E_NOMEM : 0;
}

/* The capacity of the main pass follows the reads: the lengths of 8192 evenly spaced sequences (fm6_retrieve in bulk, a few ms), the
 * longest of them rounded up to 32 -- 128 for reads of up to 128 bases.  A sequence longer than that is flagged by the main pass and
 * computed in the overflow pass below, which costs a second look at the index: fine for stragglers, not for a third of the read set
 * (reads of 70-150 bases under a fixed 128: 28 % of the rows went round again, and the ladder doubled their capacities out of memory). */
static void probe_max_len(job_t *j)
{
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
static int job_open(job_t *j)
{
    fmd_info_t info;
    const double t0 = now_s();
    if (!j->dev) j->rc = fmd_dev_open_file(j->device, j->fmd_path, &j->dev);   /* (fmdh_slim_build_dev: the caller's handle) */
    if (j->rc) return j->rc;
    fmd_dev_info(j->dev, &info);
    j->n_seq = info.mcnt[1];
    j->t_load = now_s() - t0;
    probe_max_len(j);
    return 0;
}
static void *job_main(void *p)
{
    job_t *j = (job_t *)p;
    double t0;
    if (!j->dev && job_open(j)) return 0;
    t0 = now_s();
    if (j->one_gpu_job) j->rc = fmd_ovlp_tabjob_rows(j->dev, j->n_seq, j->min_match, j->max_len, j->max_nei, STREAM_CHUNK_SHIFT, rows_sink, j, &j->tabjob);
    else {
        const uint64_t n = j->n_seq > (uint64_t)j->g ? (j->n_seq - (uint64_t)j->g + (uint64_t)j->n_dev - 1) / (uint64_t)j->n_dev : 0;
        j->rc = fmd_ovlp_packed_stream(j->dev, 0, (uint64_t)j->g, (uint64_t)j->n_dev, n, j->min_match, j->max_len, j->max_nei, 0, STREAM_CHUNK_SHIFT, rows_sink, j);
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
static int slim_build_core(const char *fmd_path, fmd_dev_t *preopened, int n_dev, const int *devices, int min_match, fmdh_slim_t **out, uint64_t *n_seq_out);
/* seconds of the last table build of this process: [0] the slowest replica's index load + transcode, [1] the slowest replica's rows (GPU + copies over PCIe +
 * folding into the slim table), [2] the whole build (rows computed again, link pass, plain steps included), [3] bytes of the table.  What bench.py --gpus N
 * reports beside the RCCL step as the rate of the path `fermi-amd unitig -g 0,1,..` takes. */
static double g_last_build[4];
void fmdh_slim_last_build(double out[4]) { memcpy(out, g_last_build, sizeof(g_last_build)); }
int fmdh_slim_build(const char *fmd_path, int n_dev, const int *devices, int min_match, fmdh_slim_t **out, uint64_t *n_seq_out)
{
    return slim_build_core(fmd_path, 0, n_dev, devices, min_match, out, n_seq_out);
}
/* the same table from an index that is already in a GPU's HBM (the in-memory API: no .fmd in between); the handle stays the caller's */
int fmdh_slim_build_dev(fmd_dev_t *dev, int min_match, fmdh_slim_t **out, uint64_t *n_seq_out)
{
    fmd_info_t info;
    int device;
    if (!dev || fmd_dev_info(dev, &info)) return 1;
    device = info.device;
    return slim_build_core("(memory)", dev, 1, &device, min_match, out, n_seq_out);
}
/* a shard of packed rows computed for `ids` (the side pass, the exact check_left pass) into the slim rows, chunk by chunk */
static int replace_from_shard(fmdh_slim_t *s, const uint64_t *ids, const fmdh_ovlp_shard_t *sh, int nt)
{
    uint64_t k;
    int rc = 0;
    for (k = 0; k < sh->n && !rc; k += (uint64_t)1 << sh->chunk_shift) {
        const uint64_t nr = sh->n - k < ((uint64_t)1 << sh->chunk_shift) ? sh->n - k : (uint64_t)1 << sh->chunk_shift;
        rc = fmdh_slim_replace(s, ids + k, sh->rec + k, sh->off + k, sh->chunk[k >> sh->chunk_shift], sh->max_nei, nr, nt);
    }
    return rc;
}
/* What follows the rows, on ONE GPU (`dev`: an open replica of the index): the rows that exceeded a capacity again (ids[n_side], ascending), links and
 * check_left (by the device's table job if there is one, by host threads otherwise), the edges the lfork does not decide through the exact kernel, the
 * plain steps.  0, or 1 with the reason on stderr; the table stays the caller's either way. */
static int slim_finish(fmdh_slim_t *s, fmd_dev_t *dev, fmd_ovlp_tabjob_t **tabjob, const uint64_t *ids, uint64_t n_side, int too_long_hint, uint32_t longest,
                       int min_match, uint32_t max_len, uint32_t max_nei, int nt, int timing)
{
    uint64_t i;
    int rc = 0, had_tabjob = 0;
    if (n_side && !dev) { fprintf(stderr, "[E::%s] %llu rows exceeded a capacity and there is no GPU to compute them again\n", __func__, (unsigned long long)n_side); return 1; }
    /* the rows that did not fit (longer sequences, more neighbours, longer lists): again, alone, with the capacities
     * raised until they do -- on the GPU; nothing falls back to the CPU */
    if (n_side) {
        fmdh_ovlp_shard_t side;
        uint32_t s_len = max_len, s_nei = max_nei;
        int attempt;
        double t1 = now_s();
        memset(&side, 0, sizeof(side));
        uint64_t over_at_cap = 0;
        int at_cap = 0;
        for (attempt = 0;; ++attempt) {
            uint64_t n_over = 0;
            /* the ladder ends at sequences of 4000 + min_match bases (candidate lists of 4095 entries): two attempts there that leave the same rows flagged
             * are rows it cannot hold -- said so, with status 1 and nothing printed, rather than a list cut short (tests/test_gpu_parity.py pins this edge,
             * which the reference, whose vectors grow, does not have: kvec.h:76-82) */
            if (attempt == 12 || at_cap >= 2) {
                fprintf(stderr, "[E::%s] %llu rows still overflow at max_len %u, max_nei %u: sequences longer than %u bases and candidate lists of 4096 entries and more are not supported\n",
                        __func__, (unsigned long long)(over_at_cap ? over_at_cap : n_side), s_len, s_nei, 4000u + (uint32_t)min_match);
                rc = 1; shard_free(&side); return rc;
            }
            /* what overflows in practice is the neighbour list of a strand in a fork-rich corner (more than max_nei irreducible overlaps): room for
             * four times as many at once, longer sequences / candidate lists only where a flagged record says so or the first attempt was not enough */
            if (attempt == 0) { s_nei *= 4; if (too_long_hint && longest > s_len) s_len = (longest + 31) / 32 * 32; }
            else { s_nei *= 2; s_len = (s_len + s_len / 2 + 31) / 32 * 32; }     /* (the candidate lists' capacity follows max_len: fmd_ovlp_list_cap) */
            if (s_len > 4000 + (uint32_t)min_match) s_len = 4000 + (uint32_t)min_match;   /* (lists of 4096 entries and more: not supported) */
            shard_free(&side);
            rc = shard_fill(dev, &side, ids, 0, 0, n_side, min_match, s_len, s_nei, 0);
            if (rc) { fprintf(stderr, "[E::%s] overflow pass: %s\n", __func__, fmd_strerror(rc)); rc = 1; return rc; }
            for (i = 0; i < n_side; ++i) n_over += (side.rec[i].flags & FMD_OVLP_F_OVERFLOW) != 0;
            if (n_over == 0) break;
            if (s_len == 4000 + (uint32_t)min_match) { at_cap = n_over == over_at_cap ? at_cap + 1 : 1; over_at_cap = n_over; }
        }
        rc = replace_from_shard(s, ids, &side, nt);
        if (!rc && (*tabjob)) {   /* the device's copy of those rows, for its link pass */
            uint64_t *n01 = (uint64_t *)malloc(n_side * 16);
            if (!n01) rc = -ENOMEM;
            else {
                for (i = 0; i < n_side; ++i) {
                    const fmd_ovlp_rec_t *r = &side.rec[i];
                    const fmd_intv_t *ne = (const fmd_intv_t *)(side.chunk[i >> side.chunk_shift] + side.off[i]);
                    const int has = fmd_ovlp_row_nei(r, side.max_nei) > 0;
                    n01[2 * i] = has ? ne[0].x[0] : ~0ull; n01[2 * i + 1] = has ? ne[0].x[1] : ~0ull;
                }
                if (fmd_ovlp_tabjob_patch((*tabjob), n_side, ids, side.rec, n01)) rc = -EIO;
                free(n01);
            }
        }
        shard_free(&side);
        if (rc) { fprintf(stderr, "[E::%s] overflow pass: cannot replace the rows (%s)\n", __func__, strerror(-rc)); rc = 1; return rc; }
        if (timing) fprintf(stderr, "[M::%s] %llu rows again with capacities %u / %u: %.3f s\n", __func__, (unsigned long long)n_side, s_len, s_nei, now_s() - t1);
    }
    /* check_left_simple (unitig.c:186-204) of every edge: decided from the lfork of the neighbour's reverse strand, on the device that
     * holds every record or by host threads; the edges that field does not decide go through the exact kernel (fmd_ovlp_check_left_dev), alone */
    {
        const uint64_t *und = 0;
        uint64_t n_und = 0, k;
        double t1 = now_s();
        if ((*tabjob)) {
            uint64_t *dev_und = 0, n_dev_und = 0;
            had_tabjob = 1;
            rc = fmd_ovlp_tabjob_link((*tabjob), links_sink, s, &dev_und, &n_dev_und);
            fmd_host_free(dev_und);                      /* (the folded rows list the same edges, with the rows of their reverse strands) */
            fmd_ovlp_tabjob_free((*tabjob)); (*tabjob) = 0;
            if (rc) { fprintf(stderr, "[E::%s] link pass on the GPU: %s\n", __func__, fmd_strerror(rc)); rc = 1; return rc; }
        } else {
            rc = fmdh_slim_link_host(s, nt);
            if (rc) { fprintf(stderr, "[E::%s] link pass: %s\n", __func__, strerror(-rc)); rc = 1; return rc; }
        }
        fmdh_slim_undecided(s, &und, &n_und);
        if (timing) fprintf(stderr, "[M::%s] link pass (%s): %.3f s, %llu edges left to the exact kernel\n", __func__, had_tabjob ? "on the GPU, folded into the rows" : "host threads", now_s() - t1,
                            (unsigned long long)n_und);
        if (n_und && !dev) { fprintf(stderr, "[E::%s] %llu edges need the exact check_left kernel and there is no GPU\n", __func__, (unsigned long long)n_und); return 1; }
        if (n_und) {
            fmdh_ovlp_shard_t ex;
            uint32_t s_len = max_len > longest ? max_len : (longest + 31) / 32 * 32, s_nei = max_nei;
            uint16_t *vals;
            int attempt;
            t1 = now_s();
            memset(&ex, 0, sizeof(ex));
            for (attempt = 0;; ++attempt) {   /* same capacity ladder as above: a row that was computed again needs its capacities here too */
                uint64_t n_over = 0;
                rc = shard_fill(dev, &ex, und, 0, 0, n_und, min_match, s_len, s_nei, 1);
                if (rc) { fprintf(stderr, "[E::%s] exact check_left pass: %s\n", __func__, fmd_strerror(rc)); rc = 1; return rc; }
                for (k = 0; k < n_und; ++k) n_over += (ex.rec[k].flags & FMD_OVLP_F_OVERFLOW) != 0;
                if (n_over == 0) break;
                if (attempt == 12) {   /* records that still overflow are invalid: their verdicts must not reach the table */
                    fprintf(stderr, "[E::%s] exact check_left pass: %llu rows still overflow at max_len %u, max_nei %u\n", __func__, (unsigned long long)n_over, s_len, s_nei);
                    rc = 1; shard_free(&ex); return rc;
                }
                s_nei *= 2; s_len = (s_len + s_len / 2 + 31) / 32 * 32;
                if (s_len > 4000 + (uint32_t)min_match) s_len = 4000 + (uint32_t)min_match;
                shard_free(&ex);
            }
            vals = (uint16_t *)malloc(n_und * 2);
            if (!vals) { rc = 1; shard_free(&ex); return rc; }
            for (k = 0; k < n_und; ++k) vals[k] = ex.rec[k].reserved;
            rc = fmdh_slim_set_reserved(s, und, vals, n_und);
            free(vals);
            shard_free(&ex);
            if (rc) { fprintf(stderr, "[E::%s] exact check_left pass: %s\n", __func__, strerror(-rc)); rc = 1; return rc; }
            if (timing) fprintf(stderr, "[M::%s] exact check_left of %llu rows: %.3f s\n", __func__, (unsigned long long)n_und, now_s() - t1);
        }
    }
    {
        const double t1 = now_s();
        if (fmdh_slim_finalize(s, nt)) { rc = 1; return rc; }
        if (timing) fprintf(stderr, "[M::%s] plain steps marked: %.3f s\n", __func__, now_s() - t1);
    }
    return 0;
}

static int slim_build_core(const char *fmd_path, fmd_dev_t *preopened, int n_dev, const int *devices, int min_match, fmdh_slim_t **out, uint64_t *n_seq_out)
{
    const int timing = getenv("FMD_TIMING") != 0;
    const int nt = fmdh_host_threads();
    uint32_t max_len = 128, longest = 0;   /* (max_len: raised by the probe of the read lengths) */
    const uint32_t max_nei = 4;
    job_t *jobs;
    pthread_t *tid;
    char *started;
    fmdh_slim_t *s = 0;
    uint64_t *ids = 0, n_side = 0, n_seq;
    int g, rc = 0, too_long_hint = 0, one_gpu;
    double t0 = now_s();
    if (n_dev < 1 || !devices || !out) return 1;
    *out = 0;
    jobs = (job_t *)calloc((size_t)n_dev, sizeof(job_t));
    tid = (pthread_t *)calloc((size_t)n_dev, sizeof(pthread_t));
    started = (char *)calloc((size_t)n_dev, 1);
    if (!jobs || !tid || !started) { free(jobs); free(tid); free(started); return 1; }
    one_gpu = n_dev == 1 && !getenv("FMD_HOST_LINK");
    for (g = 0; g < n_dev; ++g) {
        job_t x;
        memset(&x, 0, sizeof(x));
        x.fmd_path = fmd_path; x.device = devices[g]; x.g = g; x.n_dev = n_dev; x.min_match = min_match; x.max_len = max_len; x.max_nei = max_nei;
        x.conv_threads = nt / n_dev > 0 ? nt / n_dev : 1; x.one_gpu_job = one_gpu;
        jobs[g] = x;
    }
    /* replica 0 first: the number of sequences sizes the table every job writes into */
    if (preopened) jobs[0].dev = preopened;
    if (job_open(&jobs[0])) { fprintf(stderr, "[E::%s] GPU %d: %s\n", __func__, devices[0], fmd_strerror(jobs[0].rc)); rc = 1; goto done; }
    n_seq = jobs[0].n_seq;
    if (n_seq_out) *n_seq_out = n_seq;
    if (timing) fprintf(stderr, "[M::%s] index in HBM: resident set %.2f GB, peak so far %.2f GB\n", __func__, fmdh_rss_gb(0), fmdh_rss_gb(1));
    if (n_seq >= 0xffffffffull) { fprintf(stderr, "[E::%s] %llu sequences: the walk's rows hold 32-bit ids\n", __func__, (unsigned long long)n_seq); rc = 1; goto done; }
    s = fmdh_slim_new(n_seq, n_dev, !one_gpu, STREAM_CHUNK_SHIFT);
    if (!s) { fprintf(stderr, "[E::%s] out of memory (%llu rows)\n", __func__, (unsigned long long)n_seq); rc = 1; goto done; }
    for (g = 0; g < n_dev; ++g) { jobs[g].slim = s; jobs[g].max_len = jobs[0].max_len; }
    for (g = 1; g < n_dev; ++g) started[g] = pthread_create(&tid[g], 0, job_main, &jobs[g]) == 0;
    job_main(&jobs[0]);                                           /* shard 0 on the calling thread */
    for (g = 1; g < n_dev; ++g) { if (started[g]) pthread_join(tid[g], 0); else job_main(&jobs[g]); } /* no thread: do it here, afterwards */
    for (g = 0; g < n_dev; ++g) {
        if (jobs[g].rc) { fprintf(stderr, "[E::%s] GPU %d: %s%s\n", __func__, devices[g], fmd_strerror(jobs[g].rc), jobs[g].sink_rc ? " (folding a chunk into the table)" : ""); rc = 1; }
        if (timing) fprintf(stderr, "[M::%s] GPU %d: index load + transcode %.3f s, rows (GPU + copies + folding) %.3f s\n", __func__, devices[g], jobs[g].t_load, jobs[g].t_rows);
        if (timing && g == 0) fprintf(stderr, "[M::%s] after the rows: resident set %.2f GB, peak so far %.2f GB; folding shard 0 (%d threads, %.0f pieces): sizes %.3f s, allocation %.3f s, rows %.3f s\n", __func__,
                                      fmdh_rss_gb(0), fmdh_rss_gb(1), jobs[0].conv_threads, s->t_add[3], s->t_add[0], s->t_add[1], s->t_add[2]);
    }
    if (rc) goto done;
    max_len = jobs[0].max_len;
    for (g = 1; g < n_dev; ++g) if (jobs[g].n_seq != n_seq) { fprintf(stderr, "[E::%s] the replicas disagree\n", __func__); rc = 1; goto done; }
    /* the rows that did not fit: gathered from the replicas, ascending; then everything that follows the rows (slim_finish) on replica 0 */
    for (g = 0; g < n_dev; ++g) { n_side += jobs[g].n_flagged; too_long_hint |= jobs[g].too_long; if (jobs[g].longest > longest) longest = jobs[g].longest; }
    if (n_side) {
        uint64_t o = 0;
        ids = (uint64_t *)malloc(n_side * 8);
        if (!ids) { rc = 1; goto done; }
        for (g = 0; g < n_dev; ++g) { memcpy(ids + o, jobs[g].flagged, jobs[g].n_flagged * 8); o += jobs[g].n_flagged; }
        if (n_dev > 1) qsort(ids, n_side, 8, cmp_u64);   /* (one shard: ascending already) */
    }
    rc = slim_finish(s, jobs[0].dev, &jobs[0].tabjob, ids, n_side, too_long_hint, longest, min_match, max_len, max_nei, nt, timing);
    if (rc) goto done;
    g_last_build[0] = g_last_build[1] = 0;
    for (g = 0; g < n_dev; ++g) { if (jobs[g].t_load > g_last_build[0]) g_last_build[0] = jobs[g].t_load; if (jobs[g].t_rows > g_last_build[1]) g_last_build[1] = jobs[g].t_rows; }
    g_last_build[2] = now_s() - t0; g_last_build[3] = (double)fmdh_slim_bytes(s);
    if (timing) fprintf(stderr, "[M::%s] table of %llu sequences on %d GPU(s): %.3f s, %.1f bytes per row in host memory (%.2f GB); resident set now %.2f GB, peak so far %.2f GB\n", __func__,
                        (unsigned long long)n_seq, n_dev, now_s() - t0, n_seq ? (double)fmdh_slim_bytes(s) / (double)n_seq : 0.0, (double)fmdh_slim_bytes(s) / 1e9, fmdh_rss_gb(0), fmdh_rss_gb(1));
done:
    for (g = 0; g < n_dev; ++g) {
        if (jobs[g].tabjob) fmd_ovlp_tabjob_free(jobs[g].tabjob);
        if (jobs[g].dev && jobs[g].dev != preopened) fmd_dev_close(jobs[g].dev);
        free(jobs[g].flagged);
    }
    free(ids); free(jobs); free(tid); free(started);
    if (rc) fmdh_slim_free(s); else *out = s;
    return rc;
}

/* ------------------------------------------------------------------------------------------------ the root of an N-process job (fmd_host.h) */
struct fmdh_dist_root {
    fmdh_slim_t *s; uint64_t n_seq; uint32_t max_len, max_nei; int nt;
    uint64_t *flagged, n_flagged, m_flagged; uint32_t longest; int too_long;    /* rows that exceeded a capacity, as they arrived */
    uint64_t *ids64, m_ids64, rows;
};
fmdh_dist_root_t *fmdh_dist_root_new(uint64_t n_seq, uint32_t max_len)
{
    fmdh_dist_root_t *r;
    if (n_seq == 0 || n_seq >= 0xffffffffull) return 0;
    r = (fmdh_dist_root_t *)calloc(1, sizeof(*r));
    if (!r) return 0;
    r->n_seq = n_seq; r->max_len = max_len; r->nt = fmdh_host_threads();
    r->s = fmdh_slim_new(n_seq, 1, 1, STREAM_CHUNK_SHIFT);     /* (one shard, unused: the rows arrive by id, fmdh_slim_add_ids; host threads link them) */
    if (!r->s) { free(r); return 0; }
    return r;
}
void fmdh_dist_root_free(fmdh_dist_root_t *r)
{
    if (!r) return;
    fmdh_slim_free(r->s);
    free(r->flagged); free(r->ids64); free(r);
}
uint64_t fmdh_dist_root_rows(const fmdh_dist_root_t *r) { return r ? r->rows : 0; }
int fmdh_dist_root_sink(void *ctx, uint64_t n_rows, const uint32_t *ids, const fmd_ovlp_rec_t *prec, const uint64_t *off, const uint8_t *var, uint32_t max_nei)
{
    fmdh_dist_root_t *r = (fmdh_dist_root_t *)ctx;
    uint64_t k;
    if (!r || !r->s) return -EINVAL;
    if (n_rows > r->m_ids64) {
        uint64_t *q = (uint64_t *)realloc(r->ids64, n_rows * 8);
        if (!q) return -ENOMEM;
        r->ids64 = q; r->m_ids64 = n_rows;
    }
    for (k = 0; k < n_rows; ++k) {
        if (ids[k] >= r->n_seq) return -ERANGE;
        r->ids64[k] = ids[k];
    }
    r->max_nei = max_nei;
    if (fmdh_slim_add_ids(r->s, r->ids64, prec, off, var, max_nei, n_rows, r->nt)) return -ENOMEM;
    for (k = 0; k < n_rows; ++k) if (prec[k].flags & FMD_OVLP_F_OVERFLOW) {
        if (r->n_flagged == r->m_flagged) {
            const uint64_t m = r->m_flagged ? 2 * r->m_flagged : 1 << 16;
            uint64_t *q = (uint64_t *)realloc(r->flagged, m * 8);
            if (!q) return -ENOMEM;
            r->flagged = q; r->m_flagged = m;
        }
        r->flagged[r->n_flagged++] = ids[k];
        r->too_long |= (uint32_t)prec[k].len > r->max_len;
        if (prec[k].len > 0 && (uint32_t)prec[k].len > r->longest) r->longest = (uint32_t)prec[k].len;
    }
    r->rows += n_rows;
    return 0;
}
int fmdh_dist_root_finish(fmdh_dist_root_t *r, fmd_dev_t *dev, int min_match, fmdh_slim_t **out)
{
    fmd_ovlp_tabjob_t *none = 0;
    int rc;
    if (out) *out = 0;
    if (!r || !out) { fmdh_dist_root_free(r); return 1; }      /* (dev may be NULL where no row needs the GPU again: the tests' tables) */
    if (r->rows != r->n_seq) {
        fprintf(stderr, "[E::%s] %llu of %llu rows have arrived\n", __func__, (unsigned long long)r->rows, (unsigned long long)r->n_seq);
        fmdh_dist_root_free(r);
        return 1;
    }
    if (r->n_flagged > 1) qsort(r->flagged, r->n_flagged, 8, cmp_u64);
    rc = slim_finish(r->s, dev, &none, r->flagged, r->n_flagged, r->too_long, r->longest, min_match, r->max_len, r->max_nei ? r->max_nei : 4, r->nt, getenv("FMD_TIMING") != 0);
    if (!rc) { *out = r->s; r->s = 0; }
    fmdh_dist_root_free(r);
    return rc;
}
