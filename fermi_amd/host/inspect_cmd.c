/* inspect_cmd.c -- `fermi chkbwt [-p] [-r] <idx>` (cmd.c:47-130) and `fermi unpack [-i INT]... <idx>`
 * (cmd.c:132-171).  chkbwt -p prints the BWT the device layout holds (decoded on the GPU from the
 * planes, so it checks the upload/transcode path end to end); -r runs the rank self-check on the
 * GPU; unpack prints every sequence with its rank (fm_retrieve, exact.c:59) from fmd_retrieve_batch. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "fmd_host.h"

int fmdh_chkbwt(const char *fmd_path, int device, int plain, int check_rank, FILE *out)
{
    fmd_dev_t *d = 0;
    fmd_info_t info;
    int rc = fmd_dev_open_file(device, fmd_path, &d);
    if (rc) { fprintf(stderr, "[E::%s] Fail to read the index file.\n", __func__); return 1; }
    fmd_dev_info(d, &info);
    if (check_rank) {
        uint64_t n_bad = 0, first = 0;
        rc = fmd_dev_check_rank(d, &n_bad, &first);
        if (rc) { fprintf(stderr, "[E::%s] %s\n", __func__, fmd_strerror(rc)); fmd_dev_close(d); return 1; }
        if (n_bad) { fprintf(stderr, "[E::%s] rank differs from the symbol counts at %llu positions, first at %llu\n", __func__, (unsigned long long)n_bad, (unsigned long long)first); fmd_dev_close(d); return 1; }
        fprintf(stderr, "[M::%s] Checked the rank function at %llu positions.\n", __func__, (unsigned long long)info.mcnt[0]);
    }
    if (plain) {
        const uint64_t chunk = 1ull << 26;
        uint8_t *buf = (uint8_t *)malloc(chunk);
        for (uint64_t o = 0; o < info.mcnt[0] && rc == 0; o += chunk) {
            const uint64_t n = info.mcnt[0] - o < chunk ? info.mcnt[0] - o : chunk;
            rc = fmd_dev_export_bwt(d, o, n, buf);
            if (rc) { fprintf(stderr, "[E::%s] %s\n", __func__, fmd_strerror(rc)); rc = 1; break; }
            for (uint64_t i = 0; i < n; ++i) buf[i] = (uint8_t)"$ACGTN"[buf[i] < 6 ? buf[i] : 5];
            fwrite(buf, 1, n, out);
        }
        free(buf);
        if (rc == 0) fputc('\n', out);
    }
    fmd_dev_close(d);
    return rc;
}

int fmdh_unpack(const char *fmd_path, int device, int n_list, const uint64_t *list, FILE *out)
{
    fmd_dev_t *d = 0;
    fmd_info_t info;
    int rc = fmd_dev_open_file(device, fmd_path, &d);
    if (rc) { fprintf(stderr, "[E::%s] cannot load `%s': %s\n", __func__, fmd_path, fmd_strerror(rc)); return 1; }
    fmd_dev_info(d, &info);
    const uint64_t n_seq = info.mcnt[1], batch = 1u << 20;
    uint64_t total = 0, *ids;
    if (n_list) { /* only indices below the sequence count are printed (cmd.c:161-163) */
        ids = (uint64_t *)malloc((size_t)n_list * 8);
        for (int i = 0; i < n_list; ++i) if (list[i] < n_seq) ids[total++] = list[i];
    } else {
        total = n_seq;
        ids = (uint64_t *)malloc((size_t)(total ? total : 1) * 8);
        for (uint64_t i = 0; i < total; ++i) ids[i] = i;
    }
    uint32_t stride = 256;
    uint8_t *seq = 0; uint32_t *len = (uint32_t *)malloc(batch * 4); uint64_t *rank = (uint64_t *)malloc(batch * 8);
    char *line = 0; size_t line_m = 0;
    for (uint64_t o = 0; o < total && rc == 0; o += batch) {
        const uint64_t n = total - o < batch ? total - o : batch;
        for (;;) { /* a sequence longer than the stride: run the batch again with room for it */
            uint32_t mx = 0;
            free(seq); seq = (uint8_t *)malloc(n * (size_t)stride);
            rc = fmd_retrieve_batch(d, n, ids + o, seq, stride, len, rank);
            if (rc) { fprintf(stderr, "[E::%s] %s\n", __func__, fmd_strerror(rc)); rc = 1; break; }
            for (uint64_t i = 0; i < n; ++i) if (len[i] > mx) mx = len[i];
            if (mx <= stride) break;
            while (stride < mx) stride *= 2;
        }
        if (rc) break;
        for (uint64_t i = 0; i < n; ++i) { /* fm_retrieve gives the sequence reversed; print_i reverses it back (cmd.c:132-140) */
            const uint8_t *s = seq + i * (size_t)stride;
            if ((size_t)len[i] + 32 > line_m) { line_m = (size_t)len[i] * 2 + 64; line = (char *)realloc(line, line_m); }
            for (uint32_t j = 0; j < len[i]; ++j) line[j] = "$ACGTN"[s[len[i] - 1 - j] < 6 ? s[len[i] - 1 - j] : 5];
            fwrite(line, 1, len[i], out);
            fprintf(out, "\t%ld\n", (long)rank[i]);
        }
    }
    free(seq); free(len); free(rank); free(ids); free(line);
    fmd_dev_close(d);
    return rc;
}
