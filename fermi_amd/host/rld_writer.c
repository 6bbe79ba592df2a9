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
#include <errno.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "fmd_host.h"

#define WORDS_PER_BLOCK 8u
#define WORDS_PER_CHUNK (1u << 23)

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
} writer_t;

static unsigned bit_length(uint64_t v) { unsigned n = 0; while (v) { ++n; v >>= 1; } return n; }

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
        s->cur = s->head + 4;
    } else {
        uint16_t *h = (uint16_t *)(s->w + s->head);
        for (i = 0; i < 7; ++i) h[i] = (uint16_t)d[i];
        s->cur = s->head + 2;
    }
    s->last_usable = usable_tail(s->head);
    s->room = 64;
    memcpy(s->at_open, s->total, sizeof(s->total));
    return 0;
}

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

static int feed(writer_t *s, uint64_t len, int sym)
{
    int rc = 0;
    if (len == 0) return 0;
    if (sym == s->held_sym) { s->held_len += len; return 0; }
    if (s->held_len) rc = put_run(s, s->held_len, s->held_sym);
    s->held_sym = sym; s->held_len = len;
    return rc;
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
    s->head = 0; s->cur = 2; s->room = 64; s->last_usable = usable_tail(0); /* block 0: zero header */
    return 0;
}

int fmdh_write_rld_from_rle6(const uint8_t *runs, uint64_t n_bytes, const char *path)
{
    writer_t s;
    uint64_t i;
    int rc = writer_init(&s);
    for (i = 0; i < n_bytes && rc == 0; ++i)
        if (runs[i] >> 3) rc = feed(&s, runs[i] >> 3, runs[i] & 7);
    if (rc == 0) rc = finish_and_dump(&s, path);
    free(s.w);
    return rc;
}

int fmdh_write_rld_from_bwt(const uint8_t *bwt, uint64_t n, const char *path)
{
    writer_t s;
    uint64_t i = 0;
    int rc = writer_init(&s);
    while (i < n && rc == 0) {
        uint64_t j = i + 1;
        while (j < n && bwt[j] == bwt[i]) ++j;
        rc = feed(&s, j - i, bwt[i]);
        i = j;
    }
    if (rc == 0) rc = finish_and_dump(&s, path);
    free(s.w);
    return rc;
}

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
