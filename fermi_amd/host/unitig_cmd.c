/* unitig_cmd.c -- `fermi unitig` (cmd.c:184-216 -> fm6_unitig, unitig.c:378) with the index work on
 * the GPU: one fmd_ovlp_batch over all sequence ids, then the host walk (unitig_walk.c). */
#include <errno.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "fmd_host.h"

static double now_s(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec; }

int fmdh_unitig(const char *fmd_path, int device, int min_match, const char *rank_file, FILE *out)
{
    fmd_dev_t *d = 0;
    fmd_info_t info;
    const int timing = getenv("FMD_TIMING") != 0; /* phase times on stderr */
    double t0 = now_s(), t1;
    int rc = fmd_dev_open_file(device, fmd_path, &d);
    if (rc) { fprintf(stderr, "[E::%s] cannot load `%s': %s\n", __func__, fmd_path, fmd_strerror(rc)); return 1; }
    fmd_dev_info(d, &info);
    if (timing) { t1 = now_s(); fprintf(stderr, "[M::%s] index load + transcode: %.3f s\n", __func__, t1 - t0); t0 = t1; }
    const uint64_t n = info.mcnt[1];
    uint32_t max_len = 128, max_nei = 4;
    uint64_t *ids = (uint64_t *)malloc(n * 8);
    fmd_ovlp_rec_t *rec = (fmd_ovlp_rec_t *)fmdh_big_alloc(n * sizeof(*rec));
    fmd_intv_t *nei = 0;
    uint8_t *seq = 0;
    uint64_t *sorted = 0;
    fmd_ovlp_rec_t *side_rec = 0; fmd_intv_t *side_nei = 0; uint8_t *side_seq = 0; uint32_t *side_of = 0;
    if (!ids || !rec) { rc = 1; goto done; }
    if (rank_file) { /* load_sorted, cmd.c:173-182 */
        FILE *fp = fopen(rank_file, "rb");
        sorted = (uint64_t *)malloc(n * 8);
        if (!fp || !sorted || fread(sorted, 8, n, fp) != n) { fprintf(stderr, "[E::%s] cannot read the rank file `%s'\n", __func__, rank_file); if (fp) fclose(fp); rc = 1; goto done; }
        fclose(fp);
    }
    for (uint64_t i = 0; i < n; ++i) ids[i] = i;
    {
        const uint32_t stride = 2 * ((max_len + 3) / 4 * 4);
        nei = (fmd_intv_t *)fmdh_big_alloc(n * max_nei * sizeof(*nei));
        seq = (uint8_t *)fmdh_big_alloc(n * (size_t)stride);
        if (!nei || !seq) { rc = 1; goto done; }
        rc = fmd_ovlp_batch(d, n, ids, min_match, max_len, max_nei, rec, nei, seq, stride, /*check_left*/1);
        if (rc) { fprintf(stderr, "[E::%s] overlap discovery failed: %s\n", __func__, fmd_strerror(rc)); rc = 1; goto done; }
        if (timing) { t1 = now_s(); fprintf(stderr, "[M::%s] overlap table of %llu sequences (GPU + copies): %.3f s\n", __func__, (unsigned long long)n, t1 - t0); t0 = t1; }
        /* the rows that did not fit (longer sequences, more neighbours): again, alone, with capacities
         * doubled until they do; they go to the side arrays of the table */
        uint64_t n_side = 0;
        for (uint64_t i = 0; i < n; ++i) if (rec[i].flags & FMD_OVLP_F_OVERFLOW) ids[n_side++] = i;
        fmdh_ovlp_table_t t = {n, max_nei, stride, rec, nei, seq, 0, 0, 0, 0, 0};
        if (n_side) {
            uint32_t s_len = max_len, s_nei = max_nei, s_stride = stride;
            side_rec = (fmd_ovlp_rec_t *)malloc(n_side * sizeof(*side_rec));
            side_of = (uint32_t *)malloc(n * 4);
            if (!side_rec || !side_of) { rc = 1; goto done; }
            for (int attempt = 0;; ++attempt) {
                uint64_t n_over = 0;
                if (attempt == 12) { fprintf(stderr, "[E::%s] capacities exhausted\n", __func__); rc = 1; goto done; }
                s_len *= 2; s_nei *= 2; s_stride = 2 * ((s_len + 3) / 4 * 4);
                free(side_nei); free(side_seq);
                side_nei = (fmd_intv_t *)calloc(n_side * s_nei, sizeof(*side_nei));
                side_seq = (uint8_t *)calloc(n_side, s_stride);
                if (!side_nei || !side_seq) { rc = 1; goto done; }
                rc = fmd_ovlp_batch(d, n_side, ids, min_match, s_len, s_nei, side_rec, side_nei, side_seq, s_stride, 1);
                if (rc) { fprintf(stderr, "[E::%s] overlap discovery failed: %s\n", __func__, fmd_strerror(rc)); rc = 1; goto done; }
                for (uint64_t i = 0; i < n_side; ++i) n_over += (side_rec[i].flags & FMD_OVLP_F_OVERFLOW) != 0;
                if (n_over == 0) break;
            }
            memset(side_of, 0xff, n * 4);
            for (uint64_t i = 0; i < n_side; ++i) { side_of[ids[i]] = (uint32_t)i; rec[ids[i]] = side_rec[i]; }
            t.side_of = side_of; t.side_max_nei = s_nei; t.side_stride = s_stride; t.side_nei = side_nei; t.side_seq = side_seq;
            if (timing) { t1 = now_s(); fprintf(stderr, "[M::%s] %llu rows again with capacities %u / %u: %.3f s\n", __func__, (unsigned long long)n_side, s_len, s_nei, t1 - t0); t0 = t1; }
        }
        rc = fmdh_unitig_walk(&t, n, min_match, sorted, out);
        if (rc) { fprintf(stderr, "[E::%s] walk failed: %s\n", __func__, strerror(-rc)); rc = 1; }
        if (timing) { t1 = now_s(); fprintf(stderr, "[M::%s] walk + output: %.3f s\n", __func__, t1 - t0); t0 = t1; }
    }
done:
    free(ids); free(rec); free(nei); free(seq); free(sorted); free(side_rec); free(side_nei); free(side_seq); free(side_of);
    fmd_dev_close(d);
    return rc;
}
