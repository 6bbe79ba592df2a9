/* exact_cmd.c -- `fermi exact [-s] <idx> <src.fa>` (cmd.c:292-331): SMEMs of every query against
 * the index, found on the GPU (fmd_smem_batch), printed as the reference prints them
 * (cmd.c:320-327, fm6_write_smem smem.c:412-418). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "fmd_host.h"

#define EXACT_BATCH 262144

static int flush_batch(fmd_dev_t *d, const fmd_info_t *info, int self_match, size_t n, char **names, uint8_t *bases, uint64_t *off,
                       uint32_t max_len, FILE *out)
{
    uint32_t max_mem = 64;
    fmd_intv_t *mem = 0;
    uint32_t *n_mem = (uint32_t *)malloc(n * 4);
    int rc = 0;
    for (;;) { /* grow the per-read capacity until nothing overflows */
        size_t i;
        int over = 0;
        free(mem);
        mem = (fmd_intv_t *)malloc(n * (size_t)max_mem * sizeof(*mem));
        if (!mem || !n_mem) { rc = 1; break; }
        rc = fmd_smem_batch(d, n, bases, off, self_match, max_len, max_mem, mem, n_mem);
        if (rc) { fprintf(stderr, "[E::%s] %s\n", __func__, fmd_strerror(rc)); rc = 1; break; }
        for (i = 0; i < n; ++i) over |= (int)(n_mem[i] >> 31);
        if (!over) {
            for (i = 0; i < n; ++i) {
                uint32_t k;
                fprintf(out, "SQ\t%s\t%d\t%d\n", names[i], (int)(off[i + 1] - off[i]), (int)n_mem[i]);
                for (k = 0; k < n_mem[i]; ++k) {
                    const fmd_intv_t *a = &mem[i * (size_t)max_mem + k];
                    fprintf(out, "EM\t%u\t%u\t%u\t%c%c\n", (unsigned)(a->info >> 32 & 0x3fffffff), (unsigned)(a->info & 0x3fffffff),
                            (unsigned)(a->x[2] > 0xffffffffu ? 0xffffffffu : a->x[2]), "OT"[a->info >> 63], "OT"[a->x[1] < info->mcnt[1]]);
                }
                fputs("//\n", out);
            }
            break;
        }
        max_mem *= 4;
        if (max_mem > 65536) { rc = 1; break; }
    }
    free(mem); free(n_mem);
    return rc;
}

int fmdh_exact(const char *fmd_path, const char *fa_path, int device, int self_match, FILE *out)
{
    fmd_dev_t *d = 0;
    fmd_info_t info;
    int rc = fmd_dev_open_file(device, fmd_path, &d), l;
    if (rc) { fprintf(stderr, "[E::%s] cannot load `%s': %s\n", __func__, fmd_path, fmd_strerror(rc)); return 1; }
    fmd_dev_info(d, &info);
    fmdh_seqio_t *io = fmdh_seq_open(fa_path);
    if (!io) { fprintf(stderr, "[E::%s] cannot open `%s'\n", __func__, fa_path); fmd_dev_close(d); return 1; }
    char **names = (char **)calloc(EXACT_BATCH, sizeof(char *));
    uint64_t *off = (uint64_t *)malloc((EXACT_BATCH + 1) * 8);
    size_t n = 0, cap = 1 << 20, tot = 0;
    uint8_t *bases = (uint8_t *)malloc(cap);
    uint32_t max_len = 1;
    off[0] = 0;
    for (;;) {
        l = fmdh_seq_read(io);
        if (l < 0 || n == EXACT_BATCH) {
            if (n && (rc = flush_batch(d, &info, self_match, n, names, bases, off, max_len, out)) != 0) break;
            for (size_t i = 0; i < n; ++i) free(names[i]);
            n = 0; tot = 0; max_len = 1;
            if (l < 0) break;
        }
        if (tot + (size_t)l + 8 > cap) { while (tot + (size_t)l + 8 > cap) cap <<= 1; bases = (uint8_t *)realloc(bases, cap); }
        const char *s = fmdh_seq_bases(io);
        for (int i = 0; i < l; ++i) bases[tot + i] = fmdh_nt6[(unsigned char)s[i]]; /* seq_char2nt6 */
        names[n] = strdup(fmdh_seq_name(io));
        tot += (size_t)l; off[++n] = tot;
        if ((uint32_t)l > max_len) max_len = (uint32_t)l;
    }
    free(names); free(off); free(bases);
    fmdh_seq_close(io);
    fmd_dev_close(d);
    return rc;
}
