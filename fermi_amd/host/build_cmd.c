/* build_cmd.c -- `fermi build [-f] [-o out.fmd] [-l maxlen] [-O] <in.fa>` (cmd.c:378-484): read the
 * sequences, convert to nt6, trim even-length self-reverse-complement reads by one base unless -O
 * (cmd.c:457-463), build the BWT of  read $ revcomp $ ...  on the GPU (fmd_build_bwt) and write the
 * RLD\2 container the reference writes (rld_writer.c).  The output file is byte-identical to
 * `fermi build`'s; -b (block size) other than 3 and -i (append to an index) are not supported. */
#define _GNU_SOURCE
#include <limits.h>
#include <pthread.h>
#include <time.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "fmd_host.h"

static double now_s(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec; }

/* a piece of a span parsed by seqpar.c: its reads clamped to max_len, converted to nt6 in place, palindromes trimmed, empty records dropped
 * (what the loop of fmdh_build does to one read at a time); on return p->len[0 .. p->n) / p->bytes describe the reads that are left, packed */
typedef struct { fmdh_ppart_t *p; int max_len, no_fr; } prep_t;
static void *prep_main(void *d)
{
    prep_t *j = (prep_t *)d;
    fmdh_ppart_t *p = j->p;
    size_t i, src = 0, dst = 0, n = 0;
    for (i = 0; i < p->n; ++i) {
        const uint32_t l0 = p->len[i];
        uint32_t l = l0 > (uint32_t)j->max_len ? (uint32_t)j->max_len : l0, k;
        uint8_t *o = (uint8_t *)p->seq + dst;
        const unsigned char *s = (const unsigned char *)p->seq + src;
        src += l0;
        if (l == 0) continue;
        for (k = 0; k < l; ++k) o[k] = fmdh_nt6[s[k]];        /* (dst <= src: in place) */
        if (j->no_fr) l = fmdh_trim_palindrome(o, l);
        dst += l; p->len[n++] = l;
    }
    p->n = n; p->bytes = dst;
    return 0;
}
typedef struct { uint8_t *dst; const fmdh_ppart_t *p; } cpy_t;
static void *cpy_main(void *d) { cpy_t *c = (cpy_t *)d; memcpy(c->dst, c->p->seq, c->p->bytes); return 0; }

int fmdh_build(const char *fa_path, const char *out_path, int device, int max_len, int no_fr)
{
    const int timing = getenv("FMD_TIMING") != 0;
    const double t0 = now_s();
    fmdh_seqio_t *io = fmdh_seq_open(fa_path);
    if (!io) { fprintf(stderr, "[E::%s] Fail to open the input file.\n", __func__); return 1; }
    size_t cap = 1 << 20, tot = 0, n = 0, ncap = 1 << 16;
    uint8_t *bases = (uint8_t *)malloc(cap);
    uint64_t *off = (uint64_t *)malloc((ncap + 1) * 8);
    int l, rc = 0;
    off[0] = 0;
    /* a plain file: spans of it parsed, encoded and trimmed by several threads (seqpar.c); gzip and stdin: the one reader below */
    int pt = 16;
    { const char *e = getenv("FMD_HOST_THREADS"); if (e && atoi(e) > 0) pt = atoi(e); }
    fmdh_pseq_t *pr = fmdh_pseq_open(fa_path, pt, (size_t)512 << 20);
    while (pr) {
        fmdh_ppart_t *parts = 0;
        int np = 0, k;
        prep_t pj[64]; cpy_t cj[64]; pthread_t tid[64]; int st[64];
        const int got = fmdh_pseq_next(pr, &parts, &np);
        if (got <= 0) { if (got < 0) rc = 1; break; }
        for (k = 0; k < np; ++k) { pj[k].p = &parts[k]; pj[k].max_len = max_len; pj[k].no_fr = no_fr; }
        for (k = 1; k < np; ++k) st[k] = pthread_create(&tid[k], 0, prep_main, &pj[k]) == 0;
        prep_main(&pj[0]);
        for (k = 1; k < np; ++k) { if (st[k]) pthread_join(tid[k], 0); else prep_main(&pj[k]); }
        size_t add_b = 0, add_n = 0;
        for (k = 0; k < np; ++k) { add_b += parts[k].bytes; add_n += parts[k].n; }
        if (tot + add_b + 8 > cap) { while (tot + add_b + 8 > cap) cap <<= 1; bases = (uint8_t *)realloc(bases, cap); }
        if (n + add_n > ncap) { while (n + add_n > ncap) ncap <<= 1; off = (uint64_t *)realloc(off, (ncap + 1) * 8); }
        if (!bases || !off) { rc = 1; break; }
        { size_t o = tot; for (k = 0; k < np; ++k) { cj[k].dst = bases + o; cj[k].p = &parts[k]; o += parts[k].bytes; } }
        for (k = 1; k < np; ++k) st[k] = pthread_create(&tid[k], 0, cpy_main, &cj[k]) == 0;
        cpy_main(&cj[0]);
        for (k = 1; k < np; ++k) { if (st[k]) pthread_join(tid[k], 0); else cpy_main(&cj[k]); }
        for (k = 0; k < np; ++k) for (size_t i = 0; i < parts[k].n; ++i) { tot += parts[k].len[i]; off[++n] = tot; }
    }
    if (pr) fmdh_pseq_close(pr);
    if (rc) { fprintf(stderr, "[E::%s] out of memory\n", __func__); fmdh_seq_close(io); free(bases); free(off); return 1; }
    while (!pr && (l = fmdh_seq_read(io)) >= 0) {
        const char *s = fmdh_seq_bases(io);
        if (l > max_len) l = max_len;
        if (l == 0) continue; /* an empty record contributes nothing to the reference's index either */
        if (tot + (size_t)l + 8 > cap) { while (tot + (size_t)l + 8 > cap) cap <<= 1; bases = (uint8_t *)realloc(bases, cap); }
        for (int i = 0; i < l; ++i) bases[tot + i] = fmdh_nt6[(unsigned char)s[i]];
        if (no_fr) l = (int)fmdh_trim_palindrome(bases + tot, (uint32_t)l);
        if (n == ncap) { ncap <<= 1; off = (uint64_t *)realloc(off, (ncap + 1) * 8); }
        tot += (size_t)l; off[++n] = tot;
    }
    fmdh_seq_close(io);
    if (n == 0) { fprintf(stderr, "[E::%s] no sequences\n", __func__); free(bases); free(off); return 1; }
    uint64_t n_sym = 0;
    const double t1 = now_s();
    double t2;
    if (getenv("FMD_BUILD_RUNS")) {
        /* Opt-in (FMD_BUILD_RUNS=1): the BWT stays on the device and leaves it as runs (`len << 3 | sym` bytes, fmd_bwt_to_rle6) instead of a byte per
         * symbol; the container written from them is the same file (neighbouring runs of one symbol are merged by every reader of that stream,
         * rld_writer.c).  Measured (tools/ab_build.py, profiles/r4_e2e): 10^7 reads 3.5 s against 4.2 s, but at 5*10^7 the forty chunks of run-length
         * encoding cost the BWT phase 1.6-5.8 s for 0.9-2.6 s saved in the writer -- not the default. */
        void *d_reads = 0, *d_off = 0;
        uint8_t *d_bwt = 0, *rle6 = 0;
        uint64_t n_rle6 = 0;
        uint32_t mx = 0; int uniform = 1;
        size_t i;
        for (i = 0; i < n; ++i) { const uint64_t ll = off[i + 1] - off[i]; if (ll > mx) mx = (uint32_t)ll; if (ll != off[1] - off[0]) uniform = 0; }
        rc = fmd_dev_malloc(device, tot + 64, &d_reads);
        if (!rc) rc = fmd_dev_malloc(device, (n + 1) * 8, &d_off);
        if (!rc) rc = fmd_memcpy_h2d(d_reads, bases, tot, 0);
        if (!rc) rc = fmd_memcpy_h2d(d_off, off, (n + 1) * 8, 0);
        if (!rc) rc = fmd_build_bwt_dev(device, 0, n, (const uint8_t *)d_reads, (const uint64_t *)d_off, tot, mx, uniform, &d_bwt, &n_sym);
        fmd_dev_free(d_reads); fmd_dev_free(d_off);
        free(bases); bases = 0;                                    /* (5 GB the encoder does not need) */
        if (!rc) rc = fmd_bwt_to_rle6(device, d_bwt, n_sym, &rle6, &n_rle6);
        fmd_dev_free(d_bwt);
        t2 = now_s();
        if (rc) fprintf(stderr, "[E::%s] BWT construction failed: %s\n", __func__, fmd_strerror(rc));
        else {
            rc = fmdh_write_rld_from_rle6(rle6, n_rle6, out_path);
            if (rc) fprintf(stderr, "[E::%s] cannot write `%s'\n", __func__, out_path);
        }
        fmd_host_free(rle6);
    } else {   /* the byte BWT on the host */
        uint8_t *bwt = (uint8_t *)malloc(2 * (tot + n) + 64);
        rc = bwt ? fmd_build_bwt(device, n, bases, off, bwt, &n_sym) : FMD_E_NOMEM;
        t2 = now_s();
        if (rc) fprintf(stderr, "[E::%s] BWT construction failed: %s\n", __func__, fmd_strerror(rc));
        else {
            rc = fmdh_write_rld_from_bwt(bwt, n_sym, out_path);
            if (rc) fprintf(stderr, "[E::%s] cannot write `%s'\n", __func__, out_path);
        }
        free(bwt);
    }
    if (timing) fprintf(stderr, "[M::%s] %zu sequences, %llu symbols: read + encode %.3f s, BWT on the GPU (incl. copies) %.3f s, .fmd %.3f s\n", __func__, n,
                        (unsigned long long)n_sym, t1 - t0, t2 - t1, now_s() - t2);
    free(bases); free(off);
    return rc ? 1 : 0;
}
