/* exact_cmd.c -- `fermi exact [-s] <idx> <src.fa>` (cmd.c:292-331): SMEMs of every query against
 * the index, found on the GPU (fmd_smem_batch), printed as the reference prints them
 * (cmd.c:320-327, fm6_write_smem smem.c:412-418). */
#include <pthread.h>
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

/* one GPU's share [lo, hi) of the short queries of a batch (`exact -g a,b,..`: the reference hands reads to its threads the same way,
 * smem.c:379-380; the index is replicated, results land in the batch's arrays, so the output order is the input order) */
typedef struct { fmd_dev_t *d; const uint8_t *sb; const uint64_t *soff; size_t lo, hi; int self_match; uint32_t max_len, max_mem; fmd_intv_t *mem; uint32_t *n_mem; int rc; } xs_t;
static void *xs_main(void *p)
{
    xs_t *x = (xs_t *)p;
    const size_t m = x->hi - x->lo;
    x->rc = 0;
    if (m == 0) return 0;
    uint64_t *o = (uint64_t *)malloc((m + 1) * 8);   /* offsets relative to the share's first base */
    if (!o) { x->rc = FMD_E_NOMEM; return 0; }
    for (size_t i = 0; i <= m; ++i) o[i] = x->soff[x->lo + i] - x->soff[x->lo];
    x->rc = fmd_smem_batch(x->d, m, x->sb + x->soff[x->lo], o, x->self_match, x->max_len, x->max_mem, x->mem + x->lo * (size_t)x->max_mem, x->n_mem + x->lo);
    free(o);
    return 0;
}
static int smem_shares(fmd_dev_t **devs, int n_dev, size_t ns, const uint8_t *sb, const uint64_t *soff, int self_match, uint32_t max_len, uint32_t max_mem,
                       fmd_intv_t *mem, uint32_t *n_mem)
{
    xs_t x[FMDH_MAX_GPUS];
    pthread_t tid[FMDH_MAX_GPUS];
    int started[FMDH_MAX_GPUS], g, rc = 0;
    if (n_dev == 1) return fmd_smem_batch(devs[0], ns, sb, soff, self_match, max_len, max_mem, mem, n_mem);
    for (g = 0; g < n_dev; ++g) {
        x[g].d = devs[g]; x[g].sb = sb; x[g].soff = soff; x[g].lo = ns * (size_t)g / (size_t)n_dev; x[g].hi = ns * (size_t)(g + 1) / (size_t)n_dev;
        x[g].self_match = self_match; x[g].max_len = max_len; x[g].max_mem = max_mem; x[g].mem = mem; x[g].n_mem = n_mem; x[g].rc = 0;
    }
    for (g = 1; g < n_dev; ++g) started[g] = pthread_create(&tid[g], 0, xs_main, &x[g]) == 0;
    xs_main(&x[0]);
    for (g = 1; g < n_dev; ++g) { if (started[g]) pthread_join(tid[g], 0); else xs_main(&x[g]); }
    for (g = 0; g < n_dev; ++g) if (x[g].rc && !rc) rc = x[g].rc;
    return rc;
}

static int flush_batch(fmd_dev_t **devs, int n_dev, const fmd_info_t *info, int self_match, size_t n, char **names, uint8_t *bases, uint64_t *off,
                       uint32_t max_len_all, FILE *out)
{
    fmd_dev_t *d = devs[0];   /* long queries (contigs) stay on the first GPU */
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
        rc = smem_shares(devs, n_dev, ns, sb, soff, self_match, max_len, max_mem, mem, n_mem);
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
    return fmdh_exact_multi(fmd_path, fa_path, 1, &device, self_match, out);
}

int fmdh_exact_multi(const char *fmd_path, const char *fa_path, int n_dev, const int *devices, int self_match, FILE *out)
{
    fmd_dev_t *devs[FMDH_MAX_GPUS];
    fmd_info_t info;
    int rc = 0, l, g;
    if (n_dev < 1 || n_dev > FMDH_MAX_GPUS) return 1;
    memset(devs, 0, sizeof(devs));
    for (g = 0; g < n_dev && rc == 0; ++g) rc = fmd_dev_open_file(devices[g], fmd_path, &devs[g]);   /* the full index on every GPU */
    if (rc) {
        fprintf(stderr, "[E::%s] cannot load `%s': %s\n", __func__, fmd_path, fmd_strerror(rc));
        for (g = 0; g < n_dev; ++g) if (devs[g]) fmd_dev_close(devs[g]);
        return 1;
    }
    fmd_dev_info(devs[0], &info);
    fmdh_seqio_t *io = fmdh_seq_open(fa_path);
    if (!io) { fprintf(stderr, "[E::%s] cannot open `%s'\n", __func__, fa_path); for (g = 0; g < n_dev; ++g) fmd_dev_close(devs[g]); return 1; }
    char **names = (char **)calloc(EXACT_BATCH, sizeof(char *));
    uint64_t *off = (uint64_t *)malloc((EXACT_BATCH + 1) * 8);
    size_t n = 0, cap = 1 << 20, tot = 0;
    uint8_t *bases = (uint8_t *)malloc(cap);
    uint32_t max_len = 1;
    off[0] = 0;
    for (;;) {
        l = fmdh_seq_read(io);
        if (l < 0 || n == EXACT_BATCH) {
            if (n) rc = flush_batch(devs, n_dev, &info, self_match, n, names, bases, off, max_len, out);
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
    for (g = 0; g < n_dev; ++g) fmd_dev_close(devs[g]);
    return rc;
}
