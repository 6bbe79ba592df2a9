/* unitig_cmd.c -- `fermi unitig` (cmd.c:184-216 -> fm6_unitig, unitig.c:378) with the index work on
 * the GPU(s): the packed overlap table of all sequence ids (ovlp_table.c), then the host walk (unitig_walk.c). */
#include <errno.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "fmd_host.h"

static double now_s(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec; }

int fmdh_unitig(const char *fmd_path, int n_dev, const int *devices, int min_match, const char *rank_file, FILE *out)
{
    const int timing = getenv("FMD_TIMING") != 0; /* phase times on stderr */
    fmdh_slim_t *t = 0;
    uint64_t n = 0, *sorted = 0;
    double t0 = now_s();
    int rc = fmdh_slim_build(fmd_path, n_dev, devices, min_match, &t, &n);
    if (rc) { fprintf(stderr, "[E::%s] cannot build the overlap table of `%s'\n", __func__, fmd_path); return 1; }
    if (rank_file) { /* load_sorted, cmd.c:173-182 */
        FILE *fp = fopen(rank_file, "rb");
        sorted = (uint64_t *)malloc((n ? n : 1) * 8);
        if (!fp || !sorted || fread(sorted, 8, n, fp) != n) {
            fprintf(stderr, "[E::%s] cannot read the rank file `%s'\n", __func__, rank_file);
            if (fp) fclose(fp);
            rc = 1; goto done;
        }
        fclose(fp);
    }
    t0 = now_s();
    rc = fmdh_unitig_walk_slim(t, n, min_match, sorted, out, 0);
    if (rc) { fprintf(stderr, "[E::%s] walk failed: %s\n", __func__, strerror(-rc)); rc = 1; }
    if (timing) fprintf(stderr, "[M::%s] walk + output: %.3f s\n", __func__, now_s() - t0);
done:
    free(sorted);
    t0 = now_s();
    fmdh_slim_free(t);
    if (timing) fprintf(stderr, "[M::%s] table released: %.3f s\n", __func__, now_s() - t0);
    return rc;
}
