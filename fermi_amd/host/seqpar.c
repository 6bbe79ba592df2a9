/* seqpar.c -- a plain FASTA/FASTQ file parsed by several threads (fmd_host.h: fmdh_pseq_*).  One thread parses ~2 GB/s of FASTQ; `correct`
 * and `build` of 5*10^7 reads spent 3.9 and 4.8 s there, more than in their GPU stages.  The reader's semantics (seqio.c = kseq.h:171-210)
 * are stateful -- '@' may start a quality line --, so a record start cannot be recognised locally; it can be GUESSED (a line that starts
 * with '@', followed two lines later by one that starts with '+') and the guess VERIFIED: the piece before it, parsed from a position that
 * is known to be right, must end exactly there, between two records.  Pieces that verify are what one reader would have produced. */
#define _GNU_SOURCE
#include <fcntl.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include "fmd_host.h"

struct fmdh_pseq {
    const unsigned char *map; size_t size, pos, span;
    int fd, nt, serial;         /* serial: the input is not four-line FASTQ (a guess failed to verify): one reader from here on */
    fmdh_ppart_t *part;        /* nt pieces */
    size_t *cut;               /* nt + 1 piece boundaries of the current span */
    size_t *end_pos; int *clean;
};

static int part_room(fmdh_ppart_t *p, size_t more)
{
    if (p->bytes + more + 8 > p->m_bytes) {
        size_t m = p->m_bytes ? p->m_bytes : (size_t)1 << 20;
        char *a, *q;
        while (p->bytes + more + 8 > m) m <<= 1;
        a = (char *)realloc(p->seq, m); if (a) p->seq = a;
        q = (char *)realloc(p->qual, m); if (q) p->qual = q;
        if (!a || !q) return -1;
        p->m_bytes = m;
    }
    if (p->n + 1 > p->m_n) {
        const size_t m = p->m_n ? 2 * p->m_n : 1 << 16;
        uint32_t *l = (uint32_t *)realloc(p->len, m * 4);
        if (!l) return -1;
        p->len = l; p->m_n = m;
    }
    return 0;
}

/* parse [beg, stop_at): records that START before stop_at (the last one runs to its end); *end = where the reader stands afterwards */
static void parse_range(const fmdh_pseq_t *r, size_t beg, size_t stop_at, fmdh_ppart_t *dst, size_t *end, int *clean_out)
{
    /* the piece's counters change with every record and the pieces of a span sit side by side in one array (72 bytes each): the thread works on a
     * copy of its own and stores it back once -- and likewise `clean`, whose neighbours belong to the other threads */
    fmdh_ppart_t loc = *dst, *p = &loc;
    int clean_v = 0, *clean = &clean_v;
    fmdh_seqio_t *io = fmdh_seq_open_mem(r->map + beg, r->size - beg);
    p->n = p->bytes = 0; p->has_qual = 0; p->bad = 0;
    *clean = 0; *end = beg;
    if (!io) { p->bad = 1; *dst = loc; *clean_out = 0; return; }
    for (;;) {
        int len;
        if (fmdh_seq_between_records(io) && beg + fmdh_seq_mem_pos(io) >= stop_at) { *clean = 1; break; }
        len = fmdh_seq_read(io);
        if (len == -1) { *clean = 1; break; }                       /* end of file */
        if (len < 0) { p->bad = 2; break; }                         /* truncated quality */
        if (part_room(p, (size_t)len)) { p->bad = 1; break; }
        memcpy(p->seq + p->bytes, fmdh_seq_bases(io), (size_t)len);
        { const char *q = fmdh_seq_qual(io); if (q) { memcpy(p->qual + p->bytes, q, (size_t)len); p->has_qual = 1; } else memset(p->qual + p->bytes, 0, (size_t)len); }
        p->len[p->n++] = (uint32_t)len; p->bytes += (size_t)len;
        if (!fmdh_seq_between_records(io)) { /* a FASTA-style end: the next header's first character is consumed already; such input is read by one reader */
            if (beg + fmdh_seq_mem_pos(io) >= stop_at) { *clean = 0; break; }
        }
    }
    *end = beg + fmdh_seq_mem_pos(io);
    fmdh_seq_close(io);
    *dst = loc; *clean_out = clean_v;
}

/* first guessed record start at or after x: a line starting with '@' whose line after next starts with '+' */
static size_t guess_start(const fmdh_pseq_t *r, size_t x, size_t limit)
{
    const unsigned char *m = r->map;
    size_t p = x;
    int tries = 0;
    if (p == 0) return 0;
    while (p < limit && tries < 64) {
        const unsigned char *nl = (const unsigned char *)memchr(m + p, '\n', limit - p);
        size_t a, b;
        const unsigned char *n1, *n2;
        if (!nl) break;
        a = (size_t)(nl - m) + 1;                     /* a line starts here */
        if (a >= limit) break;
        p = a; ++tries;
        if (m[a] != '@') continue;
        n1 = (const unsigned char *)memchr(m + a, '\n', r->size - a); if (!n1) break;
        b = (size_t)(n1 - m) + 1; if (b >= r->size) break;
        n2 = (const unsigned char *)memchr(m + b, '\n', r->size - b); if (!n2) break;
        if ((size_t)(n2 - m) + 1 < r->size && n2[1] == '+') return a;
    }
    return limit;
}

typedef struct { fmdh_pseq_t *r; int k; } job_t;
static void *job_main(void *d)
{
    job_t *j = (job_t *)d;
    fmdh_pseq_t *r = j->r;
    parse_range(r, r->cut[j->k], r->cut[j->k + 1], &r->part[j->k], &r->end_pos[j->k], &r->clean[j->k]);
    return 0;
}

fmdh_pseq_t *fmdh_pseq_open(const char *fn, int n_threads, size_t span_bytes)
{
    struct stat st;
    fmdh_pseq_t *r;
    int fd;
    if (!fn || strcmp(fn, "-") == 0 || n_threads < 2 || getenv("FMD_SEQ_SERIAL")) return 0;
    fd = open(fn, O_RDONLY);
    if (fd < 0) return 0;
    if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode) || st.st_size < 4) { close(fd); return 0; }
    r = (fmdh_pseq_t *)calloc(1, sizeof(*r));
    if (!r) { close(fd); return 0; }
    r->fd = fd; r->size = (size_t)st.st_size; r->nt = n_threads > 64 ? 64 : n_threads; r->span = span_bytes ? span_bytes : (size_t)256 << 20;
    r->map = (const unsigned char *)mmap(0, r->size, PROT_READ, MAP_PRIVATE, fd, 0);
    if (r->map == (const unsigned char *)MAP_FAILED) { close(fd); free(r); return 0; }
    if (r->map[0] == 0x1f && r->map[1] == 0x8b) { munmap((void *)r->map, r->size); close(fd); free(r); return 0; }   /* gzip: one zlib stream, one reader */
#ifdef MADV_SEQUENTIAL
    madvise((void *)r->map, r->size, MADV_SEQUENTIAL);
#endif
    r->part = (fmdh_ppart_t *)calloc((size_t)r->nt, sizeof(fmdh_ppart_t));
    r->cut = (size_t *)calloc((size_t)r->nt + 1, sizeof(size_t));
    r->end_pos = (size_t *)calloc((size_t)r->nt, sizeof(size_t));
    r->clean = (int *)calloc((size_t)r->nt, sizeof(int));
    if (!r->part || !r->cut || !r->end_pos || !r->clean) { fmdh_pseq_close(r); return 0; }
    return r;
}

int fmdh_pseq_next(fmdh_pseq_t *r, fmdh_ppart_t **parts, int *n_parts)
{
    pthread_t tid[64];
    job_t job[64];
    int started[64], k, np, ok = 1;
    size_t stop;
    if (r->pos >= r->size) return 0;
    stop = r->pos + r->span < r->size ? r->pos + r->span : r->size;
    np = r->serial ? 1 : r->nt;
    if (stop - r->pos < ((size_t)1 << 16)) np = 1;   /* (not worth the threads) */
    r->cut[0] = r->pos;
    for (k = 1; k < np; ++k) { r->cut[k] = guess_start(r, r->pos + (stop - r->pos) / (size_t)np * (size_t)k, stop); if (r->cut[k] < r->cut[k - 1]) r->cut[k] = r->cut[k - 1]; }
    r->cut[np] = stop;
    for (k = 0; k < np; ++k) { job[k].r = r; job[k].k = k; }
    for (k = 1; k < np; ++k) started[k] = pthread_create(&tid[k], 0, job_main, &job[k]) == 0;
    job_main(&job[0]);
    for (k = 1; k < np; ++k) { if (started[k]) pthread_join(tid[k], 0); else job_main(&job[k]); }
    /* the chain: piece k is right if piece k - 1 is and ended, between two records, exactly where k starts */
    for (k = 0; k < np && ok; ++k) {
        if (r->part[k].bad == 1) return -1;
        if (k + 1 < np && r->cut[k + 1] < stop) ok = r->clean[k] && r->end_pos[k] == r->cut[k + 1] && r->part[k].bad == 0;
        else if (k + 1 < np) ok = r->clean[k] && r->part[k].bad == 0;   /* (the pieces behind it are empty: their guess ran into the end of the span) */
    }
    if (ok) {
        int last = np - 1;
        while (last > 0 && r->cut[last] >= stop) --last;            /* pieces whose start was not found hold nothing */
        for (k = last + 1; k < np; ++k) if (r->part[k].n) ok = 0;
        if (ok) {
            if (r->part[last].bad == 2) { r->pos = r->size; *parts = r->part; *n_parts = last + 1; return 1; }   /* a truncated record ends the input, as it ends kseq_read's loop (correct.c:372) */
            r->pos = r->end_pos[last];
            if (!r->clean[last] && r->pos < r->size) ok = 0;           /* a FASTA-style end inside the file: one reader for the rest */
            else { *parts = r->part; *n_parts = last + 1; return 1; }
        }
    }
    /* a guess was wrong (or the input is not four-line FASTQ): this span again, one reader from where the last span ended */
    if (np > 1) r->serial = 1;
    parse_range(r, r->cut[0], stop, &r->part[0], &r->end_pos[0], &r->clean[0]);
    if (r->part[0].bad == 1) return -1;
    if (r->part[0].bad == 2) { r->pos = r->size; *parts = r->part; *n_parts = 1; return 1; }
    if (!r->clean[0]) {   /* FASTA: the reader holds the next header's first character; step back onto it */
        r->pos = r->end_pos[0] > r->cut[0] && r->end_pos[0] < r->size ? r->end_pos[0] - 1 : r->end_pos[0];
        if (r->end_pos[0] >= r->size) r->pos = r->size;
    } else r->pos = r->end_pos[0];
    *parts = r->part; *n_parts = 1;
    return 1;
}

void fmdh_pseq_close(fmdh_pseq_t *r)
{
    int k;
    if (!r) return;
    if (r->part) for (k = 0; k < r->nt; ++k) { free(r->part[k].seq); free(r->part[k].qual); free(r->part[k].len); }
    free(r->part); free(r->cut); free(r->end_pos); free(r->clean);
    if (r->map && r->map != (const unsigned char *)MAP_FAILED) munmap((void *)r->map, r->size);
    if (r->fd >= 0) close(r->fd);
    free(r);
}
