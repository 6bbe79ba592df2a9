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
    fmd_ovlp_rec_t *rec = (fmd_ovlp_rec_t *)malloc(n * sizeof(*rec));
    fmd_intv_t *nei = 0;
    uint8_t *seq = 0;
    uint64_t *sorted = 0;
    if (!ids || !rec) { rc = 1; goto done; }
    if (rank_file) { /* load_sorted, cmd.c:173-182 */
        FILE *fp = fopen(rank_file, "rb");
        sorted = (uint64_t *)malloc(n * 8);
        if (!fp || !sorted || fread(sorted, 8, n, fp) != n) { fprintf(stderr, "[E::%s] cannot read the rank file `%s'\n", __func__, rank_file); if (fp) fclose(fp); rc = 1; goto done; }
        fclose(fp);
    }
    for (uint64_t i = 0; i < n; ++i) ids[i] = i;
    for (int attempt = 0; attempt < 8; ++attempt) { /* grow the capacities until no record overflows */
        const uint32_t stride = 2 * ((max_len + 3) / 4 * 4);
        uint64_t n_over = 0;
        free(nei); free(seq);
        nei = (fmd_intv_t *)calloc(n * max_nei, sizeof(*nei));
        seq = (uint8_t *)calloc(n, stride);
        if (!nei || !seq) { rc = 1; goto done; }
        rc = fmd_ovlp_batch(d, n, ids, min_match, max_len, max_nei, rec, nei, seq, stride, /*check_left*/1);
        if (rc) { fprintf(stderr, "[E::%s] overlap discovery failed: %s\n", __func__, fmd_strerror(rc)); rc = 1; goto done; }
        for (uint64_t i = 0; i < n; ++i) n_over += (rec[i].flags & FMD_OVLP_F_OVERFLOW) != 0;
        if (timing) { t1 = now_s(); fprintf(stderr, "[M::%s] overlap table of %llu sequences (GPU + copies): %.3f s\n", __func__, (unsigned long long)n, t1 - t0); t0 = t1; }
        if (n_over == 0) {
            fmdh_ovlp_table_t t = {n, max_nei, stride, rec, nei, seq};
            rc = fmdh_unitig_walk(&t, n, min_match, sorted, out);
            if (timing) { t1 = now_s(); fprintf(stderr, "[M::%s] walk + output: %.3f s\n", __func__, t1 - t0); t0 = t1; }
            if (rc) { fprintf(stderr, "[E::%s] walk failed: %s\n", __func__, strerror(-rc)); rc = 1; }
            goto done;
        }
        max_len *= 2; max_nei *= 2; /* whole-table retry keeps the code simple; overflows are rare */
    }
    fprintf(stderr, "[E::%s] capacities exhausted\n", __func__);
    rc = 1;
done:
    free(ids); free(rec); free(nei); free(seq); free(sorted);
    fmd_dev_close(d);
    return rc;
}
