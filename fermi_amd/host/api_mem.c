/* api_mem.c -- the reference's in-memory API (fermi.h:119-123) on the GPU path: reads in one buffer, unitigs or corrected reads out,
 * no file in between.  fm6_api_unitig (unitig.c:413-434) = fm6_build2 + unitig_core with one thread; fm6_api_correct (correct.c:464-511)
 * = fm6_build2 + ec_collect over every suffix bucket + ec_fix over every read.  Every piece is the one the CLI uses: fmd_build_bwt,
 * the overlap table (ovlp_table.c) and the walk (unitig_walk.c); fmd_kmer_collect, the device hash table and fmd_ecfix_batch. */
#include <ctype.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "fmd_host.h"

static int cmp_int(const void *a, const void *b) { const int x = *(const int *)a, y = *(const int *)b; return x < y ? -1 : x > y; }

int fmdh_api_seqlen(int64_t l, const char *seq, double quantile)
{
    int64_t i, beg, cnt = 0, j = 0;
    int *len, ret;
    for (i = 0; i < l; ++i) cnt += seq[i] == 0;
    if (cnt == 0) return 0;
    len = (int *)malloc((size_t)cnt * sizeof(int));
    if (!len) return -1;
    for (beg = i = 0; i < l; ++i) if (seq[i] == 0) { len[j++] = (int)(i - beg); beg = i + 1; }
    qsort(len, (size_t)cnt, sizeof(int), cmp_int);
    { size_t k = (size_t)((double)cnt * quantile); if (k >= (size_t)cnt) k = (size_t)cnt - 1; ret = len[k]; }   /* ks_ksmall(int, cnt, len, cnt * quantile) */
    free(len);
    return ret;
}

/* the reads of the buffer back to back (nt6, no NULs) + offsets; *n = number of reads (every NUL ends one, as fm6_build2 counts them) */
static int gather(int64_t l, const char *seq, uint8_t **bases, uint64_t **off, size_t *n)
{
    int64_t i;
    size_t cnt = 0, k = 0, t = 0;
    for (i = 0; i < l; ++i) cnt += seq[i] == 0;
    *bases = (uint8_t *)malloc((size_t)(l > 0 ? l : 1) + 8);
    *off = (uint64_t *)malloc((cnt + 1) * 8);
    if (!*bases || !*off) { free(*bases); free(*off); return -1; }
    (*off)[0] = 0;
    for (i = 0; i < l; ++i) {
        const unsigned char c = (unsigned char)seq[i];
        if (c == 0) (*off)[++k] = t;
        else (*bases)[t++] = c < 6 ? c : fmdh_nt6[c];
    }
    *n = cnt;
    return 0;
}

static int open_index(int device, size_t n, const uint8_t *bases, const uint64_t *off, fmd_dev_t **dev)
{
    uint64_t n_sym = 0;
    uint8_t *bwt = (uint8_t *)malloc(2 * ((size_t)off[n] + n) + 64);
    int rc;
    if (!bwt) return FMD_E_NOMEM;
    rc = fmd_build_bwt(device, n, bases, off, bwt, &n_sym);
    if (rc == FMD_OK) rc = fmd_dev_open_bwt(device, bwt, n_sym, dev);
    free(bwt);
    return rc;
}

int fmdh_api_unitig(int device, int min_match, int64_t l, char *seq, FILE *out)
{
    uint8_t *bases = 0;
    uint64_t *off = 0, n_seq = 0;
    size_t n = 0;
    int64_t i;
    fmd_dev_t *dev = 0;
    fmdh_slim_t *t = 0;
    int rc;
    if (l <= 0 || !seq || !out || seq[l - 1] != 0) return 1;
    if (min_match < 0) min_match = (int)(fmdh_api_seqlen(l, seq, .25) * .33 + .499);
    for (i = 0; i < l; ++i) if ((unsigned char)seq[i] > 5) seq[i] = (char)fmdh_nt6[(unsigned char)seq[i]];   /* in place, as unitig.c:422-423 */
    if (gather(l, seq, &bases, &off, &n)) return 1;
    rc = open_index(device, n, bases, off, &dev);
    free(bases); free(off);
    if (rc) { fprintf(stderr, "[E::%s] index construction failed: %s\n", __func__, fmd_strerror(rc)); return 1; }
    rc = fmdh_slim_build_dev(dev, min_match, &t, &n_seq);
    fmd_dev_close(dev);
    if (rc) { fprintf(stderr, "[E::%s] cannot build the overlap table\n", __func__); return 1; }
    rc = fmdh_unitig_walk_slim(t, n_seq, min_match, 0, out, FMDH_WALK_FULL_RECORDS);   /* what mag_g_print(g) prints (example.c:42) */
    fmdh_slim_free(t);
    return rc ? 1 : 0;
}

int fmdh_api_correct(int device, int kmer, int step, int64_t l, char *seq, char *qual)
{
    const int w = kmer > 0 ? kmer : 19, min_occ = 3, suf_len = w > 15 ? w - 15 : 1;   /* correct.c:477-481 */
    const double max_corr = 0.3;
    uint8_t *bases = 0, *q = 0;
    uint64_t *off = 0, n_trip = 0;
    uint32_t *bucket = 0, *key = 0;
    uint8_t *val = 0;
    int32_t *info = 0;
    int64_t cnt[2], i;
    size_t n = 0, r;
    fmd_dev_t *dev = 0;
    fmd_ectab_t *tab = 0;
    int rc;
    (void)max_corr;
    if (l <= 0 || !seq || seq[l - 1] != 0 || w > 27) return 1;
    if (gather(l, seq, &bases, &off, &n)) return 1;
    q = (uint8_t *)malloc((size_t)off[n] + 8);
    info = (int32_t *)calloc(n ? n : 1, sizeof(int32_t));
    if (!q || !info) { rc = FMD_E_NOMEM; goto done; }
    for (i = 0, r = 0; i < l; ++i) if (seq[i] != 0) q[r++] = qual ? (uint8_t)qual[i] : (uint8_t)(20 + 33);   /* DEFAULT_QUAL, correct.c:462 */
    if ((rc = open_index(device, n, bases, off, &dev)) != 0) goto done;
    if ((rc = fmd_kmer_collect(dev, w, min_occ, suf_len, &bucket, &key, &val, &n_trip, cnt)) != 0) goto done;
    fmd_dev_close(dev); dev = 0;
    if ((rc = fmd_ectab_build(device, w, suf_len, n_trip, bucket, key, val, &tab)) != 0) goto done;
    if ((rc = fmd_ecfix_batch(tab, n, bases, q, off, step, info)) != 0) goto done;
    /* correct.c:247-249: a base that changed becomes lower case with quality 36, the others upper case */
    for (i = 0, r = 0; i < l; ++i) {
        const unsigned char c = (unsigned char)seq[i];
        if (c == 0) continue;
        {
            const int same = (c < 6 ? c : fmdh_nt6[c]) == bases[r];
            seq[i] = same ? (char)(c < 6 ? c : toupper(c)) : "$acgtn"[bases[r] < 6 ? bases[r] : 5];
            if (qual) qual[i] = same ? (char)q[r] : 36;          /* (ec_fix1 edits the qualities of the bases it keeps, too) */
        }
        ++r;
    }
done:
    if (rc) fprintf(stderr, "[E::%s] %s\n", __func__, fmd_strerror(rc));
    if (dev) fmd_dev_close(dev);
    if (tab) fmd_ectab_free(tab);
    fmd_host_free(bucket); fmd_host_free(key); fmd_host_free(val);
    free(bases); free(off); free(q); free(info);
    return rc ? 1 : 0;
}
