/* seqio.c -- FASTA/FASTQ reader (plain or gzip) with the record semantics of the reader fermi uses
 * (kseq.h:171-210): a record starts at '>' or '@'; the name ends at the first white space (isspace); the
 * sequence may span lines and ends at the next '>', '@' or '+'; after '+' the quality has as many
 * characters as the sequence.  The corner cases follow that reader too, because `build`, `correct` and `remap`
 * must see the bytes fermi sees: the first character of every sequence line is taken as it is -- the '\n' of a
 * blank line included (it becomes an N and the next line continues the same "line") --, a CR before the line end is
 * dropped only when more than one character has been collected, and one quality line is read even for an empty
 * sequence. */
#include <ctype.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>
#include "fmd_host.h"

struct fmdh_seqio {
    gzFile fp;                  /* NULL: the bytes come from memory (fmdh_seq_open_mem) */
    unsigned char own[1 << 16];
    const unsigned char *buf;   /* own[], or the caller's memory */
    size_t beg, end;
    int eof, last_char;
    char *name, *seq, *qual, *comment;
    size_t name_l, name_m, seq_l, seq_m, qual_l, qual_m, comment_l, comment_m;
};

static int io_fill(fmdh_seqio_t *io) /* 0 at end of file */
{
    int k;
    if (io->eof) return 0;
    if (!io->fp) { io->eof = 1; return 0; }     /* memory: everything was there from the start */
    io->beg = 0;
    k = gzread(io->fp, io->own, sizeof(io->own));
    if (k <= 0) { io->eof = 1; io->end = 0; return 0; }
    io->end = (size_t)k;
    return 1;
}
static inline int io_getc(fmdh_seqio_t *io)
{
    if (io->beg >= io->end && !io_fill(io)) return -1;
    return io->buf[io->beg++];
}
static void put(char **s, size_t *l, size_t *m, int c)
{
    if (*l + 2 > *m) { *m = *m ? *m << 1 : 256; *s = (char *)realloc(*s, *m); }
    (*s)[(*l)++] = (char)c; (*s)[*l] = 0;
}
/* append the rest of the current line (without its '\n'; a trailing CR is dropped) in bulk;
 * returns -1 when the file ended before a newline and nothing was read */
static int io_append_line(fmdh_seqio_t *io, char **s, size_t *l, size_t *m)
{
    int got = 0;
    for (;;) {
        if (io->beg >= io->end && !io_fill(io)) break;
        const unsigned char *p = io->buf + io->beg, *nl = (const unsigned char *)memchr(p, '\n', (size_t)(io->end - io->beg));
        const size_t k = nl ? (size_t)(nl - p) : (size_t)(io->end - io->beg);
        if (*l + k + 2 > *m) { while (*l + k + 2 > *m) *m = *m ? *m << 1 : 256; *s = (char *)realloc(*s, *m); }
        memcpy(*s + *l, p, k); *l += k;
        got = 1;
        io->beg += k + (nl ? 1 : 0);
        if (nl) break;
    }
    if (*l > 1 && (*s)[*l - 1] == '\r') --*l;       /* kseq.h:135: only when more than one character is there */
    if (*s) (*s)[*l] = 0;
    return got ? 0 : -1;
}

fmdh_seqio_t *fmdh_seq_open(const char *fn)
{
    fmdh_seqio_t *io = (fmdh_seqio_t *)calloc(1, sizeof(*io));
    io->fp = strcmp(fn, "-") ? gzopen(fn, "r") : gzdopen(0, "r");
    if (!io->fp) { free(io); return 0; }
    gzbuffer(io->fp, 1u << 20);   /* zlib's default is 8 KiB: 270 000 read() calls for a 2.2 GB FASTQ */
    io->buf = io->own;
    put(&io->name, &io->name_l, &io->name_m, 0); io->name_l = 0;
    put(&io->seq, &io->seq_l, &io->seq_m, 0); io->seq_l = 0;
    put(&io->qual, &io->qual_l, &io->qual_m, 0); io->qual_l = 0;
    put(&io->comment, &io->comment_l, &io->comment_m, 0); io->comment_l = 0;
    return io;
}
/* the same reader over n bytes of memory (a slice of an mmap'ed file: seqpar.c parses slices concurrently) */
fmdh_seqio_t *fmdh_seq_open_mem(const void *p, size_t n)
{
    fmdh_seqio_t *io = (fmdh_seqio_t *)calloc(1, sizeof(*io));
    if (!io) return 0;
    io->buf = (const unsigned char *)p; io->beg = 0; io->end = n;
    put(&io->name, &io->name_l, &io->name_m, 0); io->name_l = 0;
    put(&io->seq, &io->seq_l, &io->seq_m, 0); io->seq_l = 0;
    put(&io->qual, &io->qual_l, &io->qual_m, 0); io->qual_l = 0;
    put(&io->comment, &io->comment_l, &io->comment_m, 0); io->comment_l = 0;
    return io;
}
/* memory readers: bytes consumed so far, and whether the reader stands between two records (nothing of the next one read yet) */
size_t fmdh_seq_mem_pos(const fmdh_seqio_t *io) { return io->beg; }
int fmdh_seq_between_records(const fmdh_seqio_t *io) { return io->last_char == 0; }
void fmdh_seq_close(fmdh_seqio_t *io)
{
    if (!io) return;
    if (io->fp) gzclose(io->fp);
    free(io->name); free(io->seq); free(io->qual); free(io->comment); free(io);
}
const char *fmdh_seq_name(const fmdh_seqio_t *io) { return io->name; }
char *fmdh_seq_bases(fmdh_seqio_t *io) { return io->seq; }
char *fmdh_seq_qual(fmdh_seqio_t *io) { return io->qual_l ? io->qual : 0; }
const char *fmdh_seq_comment(const fmdh_seqio_t *io) { return io->comment_l ? io->comment : 0; } /* rest of the header line (kseq.h:183) */

int fmdh_seq_read(fmdh_seqio_t *io) /* sequence length, -1 at end of file, -2 on a truncated quality */
{
    int c;
    if (io->last_char == 0) {
        while ((c = io_getc(io)) != -1 && c != '>' && c != '@') {}
        if (c == -1) return -1;
        io->last_char = c;
    }
    io->name_l = io->seq_l = io->qual_l = io->comment_l = 0; io->name[0] = io->seq[0] = io->qual[0] = io->comment[0] = 0;
    while ((c = io_getc(io)) != -1 && !isspace(c)) put(&io->name, &io->name_l, &io->name_m, c);
    if (c == -1 && io->name_l == 0) return -1;
    if (c != '\n') { /* comment: the rest of the line, a trailing CR dropped (kseq.h:135) */
        while ((c = io_getc(io)) != -1 && c != '\n') put(&io->comment, &io->comment_l, &io->comment_m, c);
        if (io->comment_l > 1 && io->comment[io->comment_l - 1] == '\r') io->comment[--io->comment_l] = 0;
    }
    while ((c = io_getc(io)) != -1 && c != '>' && c != '+' && c != '@') { /* the first character of each line decides (kseq.h:186-191) */
        put(&io->seq, &io->seq_l, &io->seq_m, c);                              /* whatever it is */
        io_append_line(io, &io->seq, &io->seq_l, &io->seq_m);
    }
    if (c == '>' || c == '@') io->last_char = c;
    if (c != '+') { if (c == -1) io->last_char = 0; return (int)io->seq_l; }
    while ((c = io_getc(io)) != -1 && c != '\n') {}
    if (c == -1) return -2;
    while (io_append_line(io, &io->qual, &io->qual_l, &io->qual_m) == 0 && io->qual_l < io->seq_l) {} /* whole lines, at least one (kseq.h:206) */
    io->last_char = 0;
    if (io->qual_l != io->seq_l) return -2;
    return (int)io->seq_l;
}
