/* seqsort_cmd.c -- `fermi seqsort <reads.fmd>` (cmd.c:486-505 -> fm6_seqsort, seqsort.c:37-70): the
 * rank -> read-index map the paired-end pipeline feeds to `unitig -r`.  The reference walks every
 * forward strand with fm6_retrieve (exact.c:100-127); here that walk runs on the GPU
 * (fmd_seqinfo_batch) and the table is assembled on the host exactly as seqsort.c:12-35 does. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "fmd_host.h"

int fmdh_seqsort(const char *fmd_path, int device, uint64_t **sorted_out, uint64_t *n_out)
{
    fmd_dev_t *d = 0;
    fmd_info_t info;
    int rc = fmd_dev_open_file(device, fmd_path, &d);
    if (rc) { fprintf(stderr, "[E::%s] cannot load `%s': %s\n", __func__, fmd_path, fmd_strerror(rc)); return 1; }
    fmd_dev_info(d, &info);
    const uint64_t n_seq = info.mcnt[1], n = (n_seq + 1) / 2;
    uint64_t *ids = (uint64_t *)malloc(n * 8), *sorted = (uint64_t *)calloc(n_seq, 8);
    fmd_ovlp_rec_t *rec = (fmd_ovlp_rec_t *)malloc(n * sizeof(*rec));
    uint32_t max_len = 256;
    if (!ids || !sorted || !rec) { rc = 1; goto done; }
    for (uint64_t i = 0; i < n; ++i) ids[i] = 2 * i; /* forward strands only (seqsort.c:18) */
    for (;;) {
        uint64_t n_over = 0;
        rc = fmd_seqinfo_batch(d, n, ids, max_len, rec);
        if (rc) { fprintf(stderr, "[E::%s] %s\n", __func__, fmd_strerror(rc)); rc = 1; goto done; }
        for (uint64_t i = 0; i < n; ++i) n_over += (rec[i].flags & FMD_OVLP_F_OVERFLOW) != 0;
        if (!n_over) break;
        max_len *= 4;
        if (max_len > (1u << 20)) { rc = 1; goto done; }
    }
    for (uint64_t j = 0; j < n; ++j) { /* seqsort.c:18-32 */
        const uint64_t i = 2 * j, k = rec[j].rank, x0 = rec[j].k[0], x1 = rec[j].k[1], sz = rec[j].k[2];
        const uint64_t flag = (uint64_t)(rec[j].status == FMD_OVLP_CONTAINED) << 1 | (uint64_t)(sz > 1 && k != x0);
        sorted[k] = i << 2 | flag;
        if (x0 != x1) sorted[x1 + (k - x0)] = (i | 1) << 2 | flag;
        else sorted[k + 1] = (i | 1) << 2 | flag;
    }
    {
        uint64_t cnt0 = 0, n_contained = 0, n_dups = 0;
        for (uint64_t i = 0; i < n_seq; ++i)
            if (sorted[i] == 0) ++cnt0; else if (sorted[i] & 2) ++n_contained; else if (sorted[i] & 1) ++n_dups;
        fprintf(stderr, "[M::%s] #zeros=%ld, #contained=%ld, #duplicates=%ld\n", __func__, (long)cnt0, (long)n_contained, (long)n_dups);
    }
    *sorted_out = sorted; *n_out = n_seq; sorted = 0;
done:
    free(ids); free(rec); free(sorted);
    fmd_dev_close(d);
    return rc;
}
