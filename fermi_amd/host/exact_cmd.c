/* exact_cmd.c -- `fermi exact [-s] <idx> <src.fa>` (cmd.c:292-331): SMEMs of every query against
 * the index, found on the GPU (fmd_smem_batch), printed as the reference prints them
 * (cmd.c:320-327, fm6_write_smem smem.c:412-418). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "fmd_host.h"

#define EXACT_BATCH 262144
#define EXACT_LONG 4096          /* queries longer than this go through the chain path below */
#define EXACT_CALLS (1 << 18)

/* SMEMs of one LONG query (a contig against a read index): the candidate lists of the per-read kernel
 * grow with the query, not with the match, so a long query is handled the way remap handles
 * contigs -- forward reach of every position (fmd_reach_batch), the chain x -> x + reach[x] of start
 * positions fm6_smem visits (smem.c:404-409), one fm6_smem1_core work item per chain position.  With
 * -s (self_match) the chain's step is not the plain reach, so the whole query is one item that walks
 * the chain itself.  *p_max_len = bound on the match length, grown on overflow. */
static int smem_long(fmd_dev_t *d, const uint8_t *q, uint32_t len, int self_match, uint32_t *p_max_len, fmd_intv_t **o_mem, size_t *o_n)
{
    uint8_t *buf = (uint8_t *)calloc((size_t)len + 8, 1);
    uint32_t *reach = 0, *n_mem = 0;
    fmd_smem_win_t *calls = 0;
    fmd_intv_t *mem = 0, *all = 0;
    size_t n_call = 0, all_n = 0, all_m = 0, done = 0;
    uint32_t max_mem = 256;
    int rc = 0;
    memcpy(buf, q, len);
    if (self_match) {
        calls = (fmd_smem_win_t *)calloc(1, sizeof(*calls));
        calls[0].seq_len = len; calls[0].start = 0; calls[0].stop = len; n_call = 1;
        max_mem = 4096;
    } else {
        reach = (uint32_t *)malloc(((size_t)len + 1) * 4);
        rc = fmd_reach_batch(d, (size_t)len + 1, buf, reach);
        if (rc) { fprintf(stderr, "[E::%s] %s\n", __func__, fmd_strerror(rc)); rc = 1; goto done; }
        size_t m_call = 1024;
        calls = (fmd_smem_win_t *)malloc(m_call * sizeof(*calls));
        for (uint32_t x = 0; x < len;) {
            if (n_call == m_call) { m_call <<= 1; calls = (fmd_smem_win_t *)realloc(calls, m_call * sizeof(*calls)); }
            const uint32_t r = reach[x];
            if (r) { fmd_smem_win_t *c = &calls[n_call++]; c->seq_off = 0; c->seq_len = len; c->start = x; c->stop = x + 1; c->reserved = 0; }
            x += r ? r : 1;
        }
    }
    n_mem = (uint32_t *)malloc((n_call ? n_call : 1) * 4);
    while (done < n_call && rc == 0) {
        const size_t nb = n_call - done < EXACT_CALLS ? n_call - done : EXACT_CALLS;
        for (;;) {
            int over = 0;
            free(mem);
            mem = (fmd_intv_t *)malloc(nb * (size_t)max_mem * sizeof(*mem));
            if (!mem) { rc = 1; break; }
            rc = fmd_smem_win_batch(d, nb, buf, (uint64_t)len + 1, calls + done, self_match, *p_max_len, max_mem, mem, n_mem + done);
            if (rc) { fprintf(stderr, "[E::%s] %s\n", __func__, fmd_strerror(rc)); rc = 1; break; }
            for (size_t w = 0; w < nb; ++w) over |= (int)(n_mem[done + w] >> 31);
            if (!over) break;
            if (max_mem >= (1u << 24) || *p_max_len >= (1u << 24)) { rc = 1; break; }
            max_mem *= 4; *p_max_len *= 2;
        }
        if (rc) break;
        for (size_t w = 0; w < nb; ++w) {
            const uint32_t k = n_mem[done + w];
            if (all_n + k > all_m) { all_m = (all_n + k) * 2 + 1024; all = (fmd_intv_t *)realloc(all, all_m * sizeof(*all)); }
            memcpy(all + all_n, mem + w * (size_t)max_mem, k * sizeof(*all)); all_n += k;
        }
        done += nb;
    }
done:
    free(buf); free(reach); free(n_mem); free(calls); free(mem);
    if (rc) { free(all); all = 0; all_n = 0; }
    *o_mem = all; *o_n = all_n;
    return rc;
}

static void print_query(const fmd_info_t *info, const char *name, int len, const fmd_intv_t *a, size_t n, FILE *out)
{
    fprintf(out, "SQ\t%s\t%d\t%d\n", name, len, (int)n);
    for (size_t k = 0; k < n; ++k, ++a)
        fprintf(out, "EM\t%u\t%u\t%u\t%c%c\n", (unsigned)(a->info >> 32 & 0x3fffffff), (unsigned)(a->info & 0x3fffffff),
                (unsigned)(a->x[2] > 0xffffffffu ? 0xffffffffu : a->x[2]), "OT"[a->info >> 63], "OT"[a->x[1] < info->mcnt[1]]);
    fputs("//\n", out);
}

static int flush_batch(fmd_dev_t *d, const fmd_info_t *info, int self_match, size_t n, char **names, uint8_t *bases, uint64_t *off,
                       uint32_t max_len_all, FILE *out)
{
    /* short queries: one lane each (fmd_smem_batch); long ones: smem_long, one at a time */
    size_t ns = 0, i;
    uint32_t max_len = 1, max_mem = 64, long_max_len = 256;
    uint64_t *soff = (uint64_t *)malloc((n + 1) * 8), tot = 0;
    size_t *sidx = (size_t *)malloc((n + 1) * sizeof(size_t));
    uint8_t *sb = (uint8_t *)malloc((size_t)off[n] + 8);
    fmd_intv_t *mem = 0;
    uint32_t *n_mem = (uint32_t *)malloc((n + 1) * 4);
    int rc = 0;
    (void)max_len_all;
    soff[0] = 0;
    for (i = 0; i < n; ++i) {
        const uint64_t l = off[i + 1] - off[i];
        if (l > EXACT_LONG) continue;
        memcpy(sb + tot, bases + off[i], l); tot += l;
        sidx[ns] = i; soff[++ns] = tot;
        if ((uint32_t)l > max_len) max_len = (uint32_t)l;
    }
    for (; ns;) { /* grow the per-read capacity until nothing overflows */
        int over = 0;
        free(mem);
        mem = (fmd_intv_t *)malloc(ns * (size_t)max_mem * sizeof(*mem));
        if (!mem || !n_mem) { rc = 1; break; }
        rc = fmd_smem_batch(d, ns, sb, soff, self_match, max_len, max_mem, mem, n_mem);
        if (rc) { fprintf(stderr, "[E::%s] %s\n", __func__, fmd_strerror(rc)); rc = 1; break; }
        for (i = 0; i < ns; ++i) over |= (int)(n_mem[i] >> 31);
        if (!over) break;
        max_mem *= 4;
        if (max_mem > 65536) { rc = 1; break; }
    }
    if (rc == 0) {
        size_t k = 0;
        for (i = 0; i < n && rc == 0; ++i) {
            const int l = (int)(off[i + 1] - off[i]);
            if (k < ns && sidx[k] == i) { print_query(info, names[i], l, mem + k * (size_t)max_mem, n_mem[k], out); ++k; }
            else {
                fmd_intv_t *lm = 0; size_t ln = 0;
                rc = smem_long(d, bases + off[i], (uint32_t)l, self_match, &long_max_len, &lm, &ln);
                if (rc == 0) print_query(info, names[i], l, lm, ln, out);
                free(lm);
            }
        }
    }
    free(mem); free(n_mem); free(soff); free(sidx); free(sb);
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
            if (n) rc = flush_batch(d, &info, self_match, n, names, bases, off, max_len, out);
            for (size_t i = 0; i < n; ++i) free(names[i]);     /* also when the batch failed */
            n = 0; tot = 0; max_len = 1;
            if (rc || l < 0) break;
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
