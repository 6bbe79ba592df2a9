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
    free(s->rec); free(s->off);
    memset(s, 0, sizeof(*s));
}

/* rows of the ids  ids[0..n)  or  first, first + step, ...  with capacities (max_len, max_nei) */
static int shard_fill(fmd_dev_t *d, fmdh_ovlp_shard_t *s, const uint64_t *ids, uint64_t first, uint64_t step, uint64_t n, int min_match,
                      uint32_t max_len, uint32_t max_nei)
{
    const size_t nc = (size_t)((n + ((uint64_t)1 << TABLE_CHUNK_SHIFT) - 1) >> TABLE_CHUNK_SHIFT);
    memset(s, 0, sizeof(*s));
    s->n = n; s->chunk_shift = TABLE_CHUNK_SHIFT; s->max_nei = max_nei; s->seq_stride = 2 * ((max_len + 3) / 4 * 4);
    s->rec = (fmd_ovlp_rec_t *)fmdh_big_alloc((n ? n : 1) * sizeof(fmd_ovlp_rec_t));
    s->off = (uint64_t *)fmdh_big_alloc((n ? n : 1) * 8);
    s->chunk = (uint8_t **)calloc(nc ? nc : 1, sizeof(uint8_t *));
    if (!s->rec || !s->off || !s->chunk) { shard_free(s); return FMD_E_NOMEM; }
    {
        const int rc = fmd_ovlp_packed_batch(d, ids, first, step, n, min_match, max_len, max_nei, /*check_left*/1, s->rec, s->off, s->chunk_shift, s->chunk);
        if (rc) { shard_free(s); return rc; }
    }
    return FMD_OK;
}

typedef struct {
    const char *fmd_path; int device, g, n_dev, min_match; uint32_t max_len, max_nei;
    fmdh_ovlp_shard_t *shard; fmd_dev_t *dev; uint64_t n_seq; int rc; double t_load, t_rows;
} job_t;

static void *job_main(void *p)
{
    job_t *j = (job_t *)p;
    fmd_info_t info;
    double t0 = now_s();
    j->rc = fmd_dev_open_file(j->device, j->fmd_path, &j->dev);
    if (j->rc) return 0;
    fmd_dev_info(j->dev, &info);
    j->n_seq = info.mcnt[1];
    j->t_load = now_s() - t0; t0 = now_s();
    {
        const uint64_t n = j->n_seq > (uint64_t)j->g ? (j->n_seq - (uint64_t)j->g + (uint64_t)j->n_dev - 1) / (uint64_t)j->n_dev : 0;
        j->rc = shard_fill(j->dev, j->shard, 0, (uint64_t)j->g, (uint64_t)j->n_dev, n, j->min_match, j->max_len, j->max_nei);
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
    free(t->side_of);
    memset(t, 0, sizeof(*t));
}

int fmdh_ovlp_table_build(const char *fmd_path, int n_dev, const int *devices, int min_match, fmdh_ovlp_table_t *t, uint64_t *n_seq_out)
{
    const int timing = getenv("FMD_TIMING") != 0;
    const uint32_t max_len = 128, max_nei = 4;
    job_t *jobs;
    pthread_t *tid;
    char *started;
    uint64_t *ids = 0, n_side = 0, n_seq, i;
    int g, rc = 0;
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
        job_t x = {fmd_path, devices[g], g, n_dev, min_match, max_len, max_nei, &t->shard[g], 0, 0, 0, 0, 0};
        jobs[g] = x;
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
    for (g = 1; g < n_dev; ++g) if (jobs[g].n_seq != n_seq) { fprintf(stderr, "[E::%s] the replicas disagree\n", __func__); rc = 1; goto done; }
    t->n = n_seq;
    if (n_seq_out) *n_seq_out = n_seq;
    if (n_seq >= 0xffffffffull) { fprintf(stderr, "[E::%s] %llu sequences: the walk's row map holds 32-bit ids\n", __func__, (unsigned long long)n_seq); rc = 1; goto done; }
    /* the rows that did not fit (longer sequences, more neighbours, longer lists): again, alone, with the capacities
     * doubled until they do -- on the GPU; nothing falls back to the CPU */
    for (i = 0; i < n_seq; ++i) n_side += (t->shard[i % (uint64_t)n_dev].rec[i / (uint64_t)n_dev].flags & FMD_OVLP_F_OVERFLOW) != 0;
    if (n_side) {
        uint32_t s_len = max_len, s_nei = max_nei;
        uint64_t k = 0;
        int attempt;
        double t1 = now_s();
        ids = (uint64_t *)malloc(n_side * 8);
        t->side_of = (uint32_t *)malloc(n_seq * 4);
        if (!ids || !t->side_of) { rc = 1; goto done; }
        for (i = 0; i < n_seq; ++i) if (t->shard[i % (uint64_t)n_dev].rec[i / (uint64_t)n_dev].flags & FMD_OVLP_F_OVERFLOW) ids[k++] = i;
        for (attempt = 0;; ++attempt) {
            uint64_t n_over = 0;
            if (attempt == 12) { fprintf(stderr, "[E::%s] %llu rows still overflow at max_len %u, max_nei %u\n", __func__, (unsigned long long)n_side, s_len, s_nei); rc = 1; goto done; }
            s_len *= 2; s_nei *= 2;
            shard_free(&t->side);
            rc = shard_fill(jobs[0].dev, &t->side, ids, 0, 0, n_side, min_match, s_len, s_nei);
            if (rc) { fprintf(stderr, "[E::%s] overflow pass: %s\n", __func__, fmd_strerror(rc)); rc = 1; goto done; }
            for (i = 0; i < n_side; ++i) n_over += (t->side.rec[i].flags & FMD_OVLP_F_OVERFLOW) != 0;
            if (n_over == 0) break;
        }
        memset(t->side_of, 0xff, n_seq * 4);
        for (i = 0; i < n_side; ++i) t->side_of[ids[i]] = (uint32_t)i;
        if (timing) fprintf(stderr, "[M::%s] %llu rows again with capacities %u / %u: %.3f s\n", __func__, (unsigned long long)n_side, s_len, s_nei, now_s() - t1);
    }
    if (timing) fprintf(stderr, "[M::%s] table of %llu sequences on %d GPU(s): %.3f s\n", __func__, (unsigned long long)n_seq, n_dev, now_s() - t0);
done:
    for (g = 0; g < n_dev; ++g) if (jobs[g].dev) fmd_dev_close(jobs[g].dev);
    free(ids); free(jobs); free(tid); free(started);
    if (rc) fmdh_ovlp_table_free(t);
    return rc;
}
