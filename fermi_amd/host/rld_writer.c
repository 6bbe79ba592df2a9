/* rld_writer.c -- writes fermi's on-disk index formats from a run stream produced on the GPU.
 *
 * RLD\2 (rld.c:242-263): 64-byte blocks (8 little-endian u64 words).  A block opens with the
 * symbol counts of the PREVIOUS block -- 7 x u16, or 7 x u32 with bit 31 of the first set when
 * that block held >= 0x8000 symbols (rld.c:120-129) -- followed by MSB-first run codes
 * Elias-delta(len) . 3-bit symbol (rld.c:47-53, 160-175).  A code never straddles a block; the
 * last block of every 2^23-word chunk leaves its final word unused (rld.h:66); a header-only
 * block ends the stream (rld.c:226-236).  Rank frames (rld.c:186-224) are appended even though
 * the GPU never reads them, so that the reference can load the file.
 */
/* The encoder runs on several host threads (FMD_HOST_THREADS, default 16).  Where a block ends depends on everything before it,
 * so the stream is greedy and sequential -- but two greedy packings of the same runs that start out of step fall into step
 * again within a few blocks, as soon as both happen to open a block at the same run.  Each thread therefore encodes its slice of
 * the runs SPECULATIVELY, as if a block began at its first run, and notes where its blocks start; one pass then stitches: it
 * carries on from the true state at the start of a slice, run by run, until it opens a block where the speculative encoding of
 * that slice opened one too (same first run, same header size), takes the rest of the slice's blocks as they are (only the
 * first header -- the counts of the true previous block -- is its own), and goes on to the next slice.  The last block of every
 * 2^23-word chunk has one word less; the slices cannot know which of their blocks those are, so the stitch stops in front of
 * each and re-encodes until it is in step again.  The bytes are those of the one-thread encoder (tests/test_host_formats.py). */
#include <errno.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "fmd_host.h"

#define WORDS_PER_BLOCK 8u
static uint64_t g_words_per_chunk = 1u << 23;   /* rld.h:66; set by encode() before the writers start (test hook: FMD_RLD_TEST_CHUNK_WORDS) */
#define WORDS_PER_CHUNK g_words_per_chunk

typedef struct {
    uint64_t *w;            /* payload words */
    uint64_t cap;
    uint64_t head;          /* first word of the open block */
    uint64_t cur;           /* word receiving bits */
    uint64_t last_usable;   /* last word of the open block a code may touch */
    unsigned room;          /* free bits in w[cur] */
    uint64_t total[7];      /* [0] all symbols so far, [1+c] symbol c */
    uint64_t at_open[7];    /* the same when the open block started */
    int held_sym;           /* run being accumulated */
    uint64_t held_len;
    unsigned cur_hw;        /* header words of the open block (2 or 4) */
    /* speculative slices note where their blocks start: symbols before the block, input index of its first run, header words */
    int recording;
    struct blkrec { uint64_t pos0, in0; unsigned hw; } *rec;
    uint64_t n_rec, m_rec, cur_in;
} writer_t;

static int rec_push(writer_t *s, uint64_t pos0, uint64_t in0, unsigned hw)
{
    if (s->n_rec == s->m_rec) {
        const uint64_t m = s->m_rec ? s->m_rec * 2 : 1024;
        struct blkrec *r = (struct blkrec *)realloc(s->rec, m * sizeof(*r));
        if (!r) return -ENOMEM;
        s->rec = r; s->m_rec = m;
    }
    s->rec[s->n_rec].pos0 = pos0; s->rec[s->n_rec].in0 = in0; s->rec[s->n_rec].hw = hw;
    ++s->n_rec;
    return 0;
}

static inline unsigned bit_length(uint64_t v) { return v ? 64u - (unsigned)__builtin_clzll(v) : 0u; }

static int grow(writer_t *s, uint64_t need)
{
    if (need <= s->cap) return 0;
    uint64_t ncap = s->cap ? s->cap : 4096;
    while (ncap < need) ncap *= 2;
    uint64_t *nw = (uint64_t *)realloc(s->w, ncap * 8);
    if (!nw) return -ENOMEM;
    memset(nw + s->cap, 0, (ncap - s->cap) * 8);
    s->w = nw; s->cap = ncap;
    return 0;
}

static uint64_t usable_tail(uint64_t head)
{
    const int chunk_end = ((head + WORDS_PER_BLOCK) % WORDS_PER_CHUNK) == 0;
    return head + WORDS_PER_BLOCK - (chunk_end ? 2 : 1);
}

static int open_next_block(writer_t *s)
{
    uint64_t d[7];
    int i, rc;
    for (i = 0; i < 7; ++i) d[i] = s->total[i] - s->at_open[i];
    s->head += WORDS_PER_BLOCK;
    if ((rc = grow(s, s->head + 2 * WORDS_PER_BLOCK)) != 0) return rc;
    if (d[0] >= 0x8000) {
        uint32_t *h = (uint32_t *)(s->w + s->head);
        for (i = 0; i < 7; ++i) h[i] = (uint32_t)d[i];
        h[0] |= 0x80000000u;
        s->cur = s->head + 4; s->cur_hw = 4;
    } else {
        uint16_t *h = (uint16_t *)(s->w + s->head);
        for (i = 0; i < 7; ++i) h[i] = (uint16_t)d[i];
        s->cur = s->head + 2; s->cur_hw = 2;
    }
    s->last_usable = s->recording ? s->head + WORDS_PER_BLOCK - 1 : usable_tail(s->head);   /* (a slice cannot know its chunk ends: the stitch does) */
    s->room = 64;
    memcpy(s->at_open, s->total, sizeof(s->total));
    if (s->recording && (rc = rec_push(s, s->total[0], s->cur_in, s->cur_hw)) != 0) return rc;
    return 0;
}

static unsigned code_width(uint64_t len) /* bits of the code of a run: gamma(nbits) . low bits of len . 3-bit symbol */
{
    const unsigned nbits = bit_length(len);
    return 2 * bit_length(nbits) - 1 + (nbits - 1) + 3;
}
static int must_open(const writer_t *s, uint64_t len) { return code_width(len) >= s->room && s->cur == s->last_usable; }

static int put_run(writer_t *s, uint64_t len, int sym)
{
    /* delta code: gamma(nbits) . low bits of len (without its leading 1) . 3-bit symbol */
    const unsigned nbits = bit_length(len);           /* = floor(log2 len) + 1 */
    const unsigned gamma_w = 2 * bit_length(nbits) - 1;
    unsigned width = gamma_w + (nbits - 1) + 3;
    const uint64_t low = len & ((1ull << (nbits - 1)) - 1);
    const uint64_t code = (((uint64_t)nbits << (nbits - 1)) | low) << 3 | (uint64_t)sym;
    int rc;
    if (width >= s->room && s->cur == s->last_usable && (rc = open_next_block(s)) != 0) return rc;
    if (width > s->room) {
        width -= s->room;
        s->w[s->cur++] |= code >> width;
        s->room = 64 - width;
        s->w[s->cur] = code << s->room;
    } else {
        s->room -= width;
        s->w[s->cur] |= code << s->room;
    }
    s->total[0] += len;
    s->total[1 + sym] += len;
    return 0;
}

static void read_header(const uint64_t *blk, uint64_t h[7])
{
    int i;
    if ((uint32_t)blk[0] >> 31) {
        const uint32_t *q = (const uint32_t *)blk;
        h[0] = q[0] & 0x7fffffffu;
        for (i = 1; i < 7; ++i) h[i] = q[i];
    } else {
        const uint16_t *q = (const uint16_t *)blk;
        for (i = 0; i < 7; ++i) h[i] = q[i];
    }
}

static int finish_and_dump(writer_t *s, const char *path)
{
    int rc, i;
    uint64_t n_words, n_frames, *frame, k, b, mcnt[7], run[6] = {0, 0, 0, 0, 0, 0};
    int ibits;
    FILE *fp;
    if (s->held_len && (rc = put_run(s, s->held_len, s->held_sym)) != 0) return rc;
    if ((rc = open_next_block(s)) != 0) return rc;   /* closing header-only block */
    n_words = s->cur;
    memcpy(mcnt, s->total, sizeof(mcnt));
    /* frames (rld.c:186-224) */
    {
        const uint64_t n_blks = n_words / WORDS_PER_BLOCK + 1, last = n_words / WORDS_PER_BLOCK * WORDS_PER_BLOCK;
        ibits = (int)bit_length((uint32_t)(mcnt[0] / n_blks)) - 1 + 4;
        n_frames = ((mcnt[0] + (1ull << ibits) - 1) >> ibits) + 1;
        frame = (uint64_t *)calloc(n_frames * 7, 8);
        if (!frame) return -ENOMEM;
        for (b = WORDS_PER_BLOCK, k = 1; b <= last; b += WORDS_PER_BLOCK) {
            uint64_t h[7], sum = 0;
            read_header(s->w + b, h);
            for (i = 0; i < 6; ++i) { run[i] += h[i + 1]; sum += run[i]; }
            while (sum >= k << ibits) ++k;
            if (k < n_frames) {
                frame[k * 7] = b;
                for (i = 0; i < 6; ++i) frame[k * 7 + 1 + i] = run[i];
            }
        }
        for (k = 1; k < n_frames; ++k)
            if (frame[k * 7] == 0) memcpy(frame + k * 7, frame + (k - 1) * 7, 56);
    }
    fp = strcmp(path, "-") ? fopen(path, "wb") : stdout; /* "-" = stdout, like rld_dump (rld.c:248) */
    if (!fp) { free(frame); return -errno; }
    {
        const uint32_t a = 6u << 16 | 3u;
        const uint64_t zero = 0, n_bytes = n_words * 8;
        fwrite("RLD\2", 1, 4, fp); fwrite(&a, 4, 1, fp); fwrite(&zero, 8, 1, fp);
        fwrite(&n_bytes, 8, 1, fp); fwrite(&n_frames, 8, 1, fp); fwrite(mcnt + 1, 8, 6, fp);
        fwrite(s->w, 8, n_words, fp);
        fwrite(frame, 56, n_frames, fp);
    }
    rc = ferror(fp) ? -EIO : 0;
    if (fp != stdout) fclose(fp); else fflush(fp);
    free(frame);
    return rc;
}

static int writer_init(writer_t *s)
{
    memset(s, 0, sizeof(*s));
    s->held_sym = -1;
    if (grow(s, 2 * WORDS_PER_BLOCK)) return -ENOMEM;
    s->head = 0; s->cur = 2; s->cur_hw = 2; s->room = 64; s->last_usable = usable_tail(0); /* block 0: zero header */
    return 0;
}

/* ---- the runs of a source (a byte BWT, or an RLE\6 stream: byte = len << 3 | sym), neighbours of one symbol merged ---------- */
typedef struct { const uint8_t *src; int is_bwt; uint64_t i, end; } runit_t;
static inline int src_sym(const runit_t *it, uint64_t i) { return it->is_bwt ? it->src[i] : (it->src[i] & 7); }
static int run_next(runit_t *it, uint64_t *len, int *sym, uint64_t *in)
{
    uint64_t i = it->i, l = 0;
    int c;
    if (!it->is_bwt) while (i < it->end && (it->src[i] >> 3) == 0) ++i;   /* empty runs */
    if (i >= it->end) { it->i = i; return 0; }
    *in = i; c = src_sym(it, i);
    if (it->is_bwt) { uint64_t j = i + 1; while (j < it->end && it->src[j] == (uint8_t)c) ++j; l = j - i; i = j; }
    else for (; i < it->end && ((it->src[i] >> 3) == 0 || (it->src[i] & 7) == c); ++i) l += it->src[i] >> 3;
    *len = l; *sym = c; it->i = i;
    return 1;
}

#include <time.h>
static double rw_now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec; }
static int encode_sequential(const uint8_t *src, int is_bwt, uint64_t n, const char *path)
{
    writer_t s;
    runit_t it = {src, is_bwt, 0, n};
    uint64_t len, in;
    int sym, rc = writer_init(&s);
    while (rc == 0 && run_next(&it, &len, &sym, &in)) rc = put_run(&s, len, sym);
    if (rc == 0) rc = finish_and_dump(&s, path);
    free(s.w);
    return rc;
}

/* ---- speculative slices + stitch ------------------------------------------------------------------------------------------- */
/* (aligned: the writers' counters change with every run, and two slices in one cache line made 16 threads no faster than one) */
typedef struct { writer_t w; const uint8_t *src; int is_bwt; uint64_t beg, end; int rc; } __attribute__((aligned(128))) slice_t;
static void *slice_main(void *p)
{
    slice_t *sl = (slice_t *)p;
    writer_t *s = &sl->w;
    runit_t it = {sl->src, sl->is_bwt, sl->beg, sl->end};
    uint64_t len, in;
    int sym;
    if ((sl->rc = writer_init(s)) != 0) return 0;          /* as if a block with a two-word header began at the first run */
    s->recording = 1;
    s->last_usable = WORDS_PER_BLOCK - 1;
    if ((sl->rc = rec_push(s, 0, sl->beg, 2)) != 0) return 0;
    while (sl->rc == 0 && run_next(&it, &len, &sym, &in)) { s->cur_in = in; sl->rc = put_run(s, len, sym); }
    return 0;
}
static int block_is_empty(const writer_t *s) { return s->cur == s->head + s->cur_hw && s->room == 64; }

/* W has just opened an empty block where slice sl opened its block j: take the slice's blocks from j on.  Returns 1 when the
 * rest of the slice was taken, 0 when the take stopped in front of a chunk-end block (*resume_in = where to go on), < 0 on error */
static int adopt(writer_t *W, const slice_t *sl, uint64_t j, const uint64_t off[7], uint64_t *resume_in, uint64_t *resume_j)
{
    const writer_t *S = &sl->w;
    uint64_t e, k, hdr[4];
    int i, rc;
    for (e = j; e < S->n_rec; ++e)                      /* the first block that is the last of a chunk in the true stream */
        if (((W->head + (e - j) * WORDS_PER_BLOCK + WORDS_PER_BLOCK) % WORDS_PER_CHUNK) == 0) break;
    if (e == j) return 0;                                 /* the open block itself: nothing to take, go on run by run */
    if ((rc = grow(W, W->head + (e - j + 2) * WORDS_PER_BLOCK)) != 0) return rc;
    memcpy(hdr, W->w + W->head, W->cur_hw * 8);          /* the true header of the first block taken */
    memcpy(W->w + W->head, S->w + j * WORDS_PER_BLOCK, (size_t)(e - j) * WORDS_PER_BLOCK * 8);
    memcpy(W->w + W->head, hdr, W->cur_hw * 8);
    if (e == S->n_rec) {                                  /* to the end of the slice: its open block becomes ours */
        const uint64_t shift = W->head - j * WORDS_PER_BLOCK;
        W->head = S->head + shift; W->cur = S->cur + shift; W->room = S->room; W->cur_hw = S->cur_hw;
        W->last_usable = usable_tail(W->head);
        for (i = 0; i < 7; ++i) { W->total[i] = off[i] + S->total[i]; W->at_open[i] = off[i] + S->at_open[i]; }
        memset(W->w + W->cur + 1, 0, (size_t)(W->head + 2 * WORDS_PER_BLOCK - W->cur - 1) * 8);   /* nothing stale behind the write position */
        return 1;
    }
    /* blocks j .. e-1 are complete; block e must be encoded again, one word shorter.  Symbol counts at its start = the slice's
     * own headers summed (header k = the counts of block k - 1) */
    {
        uint64_t tot[7] = {0, 0, 0, 0, 0, 0, 0}, prev[7] = {0, 0, 0, 0, 0, 0, 0}, h[7];
        for (k = 1; k <= e; ++k) {
            read_header(S->w + k * WORDS_PER_BLOCK, h);
            if (k == e) memcpy(prev, tot, sizeof(tot));
            for (i = 0; i < 7; ++i) tot[i] += h[i];
        }
        if (e == 1) memset(prev, 0, sizeof(prev));
        W->head += (e - 1 - j) * WORDS_PER_BLOCK;
        W->last_usable = usable_tail(W->head); W->cur = W->last_usable; W->room = 0;   /* full: the next run opens block e */
        W->cur_hw = S->rec[e - 1].hw;
        for (i = 0; i < 7; ++i) { W->total[i] = off[i] + tot[i]; W->at_open[i] = off[i] + prev[i]; }
        memset(W->w + W->head + WORDS_PER_BLOCK, 0, WORDS_PER_BLOCK * 8);
    }
    *resume_in = S->rec[e].in0; *resume_j = e;
    return 0;
}

static int encode_parallel(const uint8_t *src, int is_bwt, uint64_t n, const char *path, int T)
{
    slice_t *sl = 0;
    if (posix_memalign((void **)&sl, 128, (size_t)T * sizeof(*sl)) != 0) sl = 0; else memset(sl, 0, (size_t)T * sizeof(*sl));
    pthread_t *tid = (pthread_t *)calloc((size_t)T, sizeof(*tid));
    int *started = (int *)calloc((size_t)T, sizeof(int));
    writer_t W;
    int t, rc = 0;
    uint64_t cut = 0;
    if (!sl || !tid || !started) { free(sl); free(tid); free(started); return -ENOMEM; }
    memset(&W, 0, sizeof(W));
    for (t = 0; t < T; ++t) {   /* slices begin where the symbol changes, so that no merged run spans two of them */
        runit_t probe = {src, is_bwt, 0, n};
        uint64_t b = t ? n / (uint64_t)T * (uint64_t)t : 0;
        if (b < cut) b = cut;
        if (t) while (b < n && b > 0 && (is_bwt ? src[b] == src[b - 1] : ((src[b] >> 3) == 0 || (src[b - 1] >> 3) == 0 || src_sym(&probe, b) == src_sym(&probe, b - 1)))) ++b;
        sl[t].src = src; sl[t].is_bwt = is_bwt; sl[t].beg = b;
        if (t) sl[t - 1].end = b;
        cut = b;
    }
    sl[T - 1].end = n;
    const double t_beg = rw_now();
    for (t = 1; t < T; ++t) started[t] = pthread_create(&tid[t], 0, slice_main, &sl[t]) == 0;
    slice_main(&sl[0]);
    for (t = 1; t < T; ++t) { if (started[t]) pthread_join(tid[t], 0); else slice_main(&sl[t]); }
    const double t_sl = rw_now();
    for (t = 0; t < T && rc == 0; ++t) rc = sl[t].rc;
    if (rc == 0) rc = writer_init(&W);
    {   /* room for everything the slices wrote, so that the stitch does not realloc its way up */
        uint64_t words = 4 * WORDS_PER_BLOCK;
        for (t = 0; t < T; ++t) words += sl[t].w.head + 2 * WORDS_PER_BLOCK;
        if (rc == 0) rc = grow(&W, words + words / 64);
    }
    for (t = 0; t < T && rc == 0; ++t) {
        const slice_t *S = &sl[t];
        runit_t it = {src, is_bwt, S->beg, S->end};
        uint64_t off[7], jh = 0, len, in;
        int sym, done = 0;
        memcpy(off, W.total, sizeof(off));
        int at_start = block_is_empty(&W);          /* (the very first block of the stream; any other slice starts inside a block) */
        while (rc == 0 && !done) {
            int fresh = at_start;
            at_start = 0;
            if (!run_next(&it, &len, &sym, &in)) break;
            if (!fresh && must_open(&W, len)) { W.cur_in = in; if ((rc = open_next_block(&W)) != 0) break; fresh = 1; }
            if (fresh) {   /* an empty block whose first run is this one: in step with the slice?  (same first run, same header size) */
                const uint64_t P = W.total[0] - off[0];
                while (jh < S->w.n_rec && S->w.rec[jh].pos0 < P) ++jh;
                if (jh < S->w.n_rec && S->w.rec[jh].pos0 == P && S->w.rec[jh].hw == W.cur_hw) {
                    uint64_t rin = 0, rj = 0;
                    const int a = adopt(&W, S, jh, off, &rin, &rj);
                    if (a < 0) { rc = a; break; }
                    if (a == 1) { done = 1; break; }
                    if (rj > jh) { it.i = rin; jh = rj; continue; }   /* took the blocks up to a chunk end: this run is in them */
                }
            }
            rc = put_run(&W, len, sym);
        }
        free(sl[t].w.w); sl[t].w.w = 0; free(sl[t].w.rec); sl[t].w.rec = 0;
    }
    const double t_st = rw_now();
    if (rc == 0) rc = finish_and_dump(&W, path);
    if (getenv("FMD_TIMING")) fprintf(stderr, "[M::%s] %d slices encoded in %.3f s, stitched in %.3f s, frames + file %.3f s\n", __func__, T, t_sl - t_beg, t_st - t_sl, rw_now() - t_st);
    for (t = 0; t < T; ++t) { free(sl[t].w.w); free(sl[t].w.rec); }
    free(W.w); free(sl); free(tid); free(started);
    return rc;
}

static int encode(const uint8_t *src, int is_bwt, uint64_t n, const char *path)
{
    int T = 16;
    /* test hook, read ONCE before any writer thread starts: a shrunken chunk exercises the stitch at chunk ends.  The reference cannot
     * load such a file (RLD_LSIZE, rld.h:66), so the product honours it only together with FMD_RLD_TEST_HOOKS=1. */
    { const char *e = getenv("FMD_RLD_TEST_CHUNK_WORDS"), *on = getenv("FMD_RLD_TEST_HOOKS");
      g_words_per_chunk = (on && atoi(on) == 1 && e && atoll(e) >= 64) ? (uint64_t)atoll(e) / 8 * 8 : 1u << 23; }
    { const char *e = getenv("FMD_HOST_THREADS"); if (e && atoi(e) > 0) T = atoi(e); }
    { const char *e = getenv("FMD_RLD_THREADS"); if (e && atoi(e) > 0) T = atoi(e); }   /* (tests: 1 = the one-thread encoder) */
    if (T > 256) T = 256;
    if (T < 2 || n < (uint64_t)T * 4096) return encode_sequential(src, is_bwt, n, path);
    return encode_parallel(src, is_bwt, n, path, T);
}

int fmdh_write_rld_from_rle6(const uint8_t *runs, uint64_t n_bytes, const char *path) { return encode(runs, 0, n_bytes, path); }
int fmdh_write_rld_from_bwt(const uint8_t *bwt, uint64_t n, const char *path) { return encode(bwt, 1, n, path); }

int fmdh_write_rle6(const uint8_t *runs, uint64_t n_bytes, const char *path)
{
    FILE *fp = fopen(path, "wb");
    int rc;
    if (!fp) return -errno;
    fwrite("RLE\6", 1, 4, fp);
    fwrite(runs, 1, n_bytes, fp);
    rc = ferror(fp) ? -EIO : 0;
    fclose(fp);
    return rc;
}

const uint8_t fmdh_nt6[256] = {
    0, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5,
    5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5,
    5, 1, 5, 2, 5, 5, 5, 3, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 4, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5,
    5, 1, 5, 2, 5, 5, 5, 3, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 4, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5,
    5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5,
    5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5,
    5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5,
    5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5};

uint32_t fmdh_trim_palindrome(const uint8_t *s, uint32_t len)
{
    uint32_t i;
    if (len == 0 || (len & 1)) return len;
    for (i = 0; i < len / 2; ++i)
        if (s[i] + s[len - 1 - i] != 5) return len;
    return len - 1;
}
