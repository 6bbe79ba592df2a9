/* oracle/rld_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of fermi's run-length-delta BWT container: Elias-delta run codec, block
 * headers, rank frames, .fmd I/O and the rank1a / rank2a queries.  Restates rld.c / rld.h of
 * the reference (file:line cited per function); held as one flat word array instead of the
 * reference's 2^23-word chunk table (the bytes are the same, rld.c:310-318).
 * Parity: pinned against oracle/_ref (the compiled reference) and tests/golden/ vectors.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "fmd_oracle.h"

#define BLK_WORDS 8            /* ssize = 1<<sbits, sbits = 3 (rld.c:69, cmd.c:380) */
#define CHUNK_WORDS (1u << 23) /* RLD_LSIZE (rld.h:9-10) */

static __thread orc_counters_t tl_counters;

static orc_counters_t g_counters; /* totals handed over by finished worker threads */

void orc_counters_flush(void) /* called by batch workers before they exit */
{
    __atomic_fetch_add(&g_counters.rank1a, tl_counters.rank1a, __ATOMIC_RELAXED);
    __atomic_fetch_add(&g_counters.rank2a, tl_counters.rank2a, __ATOMIC_RELAXED);
    __atomic_fetch_add(&g_counters.rank2a_spill, tl_counters.rank2a_spill, __ATOMIC_RELAXED);
    memset(&tl_counters, 0, sizeof(tl_counters));
}

orc_counters_t orc_counters_read(void) /* calling thread's + flushed workers'; clears both */
{
    orc_counters_t c;
    orc_counters_flush();
    c.rank1a = __atomic_exchange_n(&g_counters.rank1a, 0, __ATOMIC_RELAXED);
    c.rank2a = __atomic_exchange_n(&g_counters.rank2a, 0, __ATOMIC_RELAXED);
    c.rank2a_spill = __atomic_exchange_n(&g_counters.rank2a_spill, 0, __ATOMIC_RELAXED);
    return c;
}

static int floor_log2_u32(uint32_t v) /* ilog2() of rld.c:40-45; -1 for 0 */
{
    int r = -1;
    while (v) { ++r; v >>= 1; }
    return r;
}

/* ------------------------------------------------------------------------------------------
 * Block headers (rld.c:120-129, rld.h:68): 7 x u16, or 7 x u32 when bit 31 of the first u32
 * is set.  Counter 0 = symbols in the PREVIOUS block, 1..6 = its $ACGTN counts.
 * ---------------------------------------------------------------------------------------- */
static inline int hdr_is32(const uint64_t *blk) { return (uint32_t)blk[0] >> 31; }

static inline void hdr_read(const uint64_t *blk, uint64_t h[7])
{
    int j;
    if (hdr_is32(blk)) {
        const uint32_t *q = (const uint32_t *)blk;
        h[0] = q[0] & 0x7fffffffu;
        for (j = 1; j < 7; ++j) h[j] = q[j];
    } else {
        const uint16_t *q = (const uint16_t *)blk;
        for (j = 0; j < 7; ++j) h[j] = q[j];
    }
}
static inline int hdr_words(const uint64_t *blk) { return hdr_is32(blk) ? 4 : 2; } /* rld.c:76-77 */

/* ------------------------------------------------------------------------------------------
 * Run decoder (rld.h:77-94 / rld.c:395-415).  Bits are consumed MSB-first.  `bit` is an
 * absolute bit offset into e->w; reads never look past word `wend` (exclusive).
 * Returns the run length (0 = block exhausted) and the symbol.
 * ---------------------------------------------------------------------------------------- */
static inline uint64_t peek_bits(const uint64_t *w, uint64_t bit, uint64_t wend)
{
    uint64_t i = bit >> 6;
    unsigned off = bit & 63;
    uint64_t x;
    if (i >= wend) return 0;
    x = w[i] << off;
    if (off && i + 1 < wend) x |= w[i + 1] >> (64 - off);
    return x;
}

static inline uint64_t get_run(const uint64_t *w, uint64_t *bit, uint64_t wend, int *sym)
{
    uint64_t x = peek_bits(w, *bit, wend), len;
    if (x >> 63) { /* '1' = length 1, then 3 symbol bits */
        *sym = (x >> 60) & 7;
        *bit += 4;
        return 1;
    } else {
        int z = x ? __builtin_clzll(x) : 64;
        int gw, nlow;
        if (z >= 6) return 0;                /* rld.h:84: >=6 leading zeros = padding */
        gw = 2 * z + 1;                      /* gamma code of v = floor(log2 len) + 1 */
        nlow = (int)(x >> (64 - gw)) - 1;    /* v - 1 explicit low bits follow */
        len = ((x << gw) >> (64 - nlow)) | (1ull << nlow);
        *sym = (int)((x << (gw + nlow)) >> 61);
        *bit += gw + nlow + 3;
        return len;
    }
}

/* ------------------------------------------------------------------------------------------
 * Encoder (rld.c:47-53, 111-184, 226-236).
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    orc_rld_t *e;
    uint64_t cap;      /* allocated words */
    uint64_t shead;    /* word offset of the current block */
    uint64_t p;        /* word being written */
    uint64_t stail;    /* last writable word of the block (rld.h:66) */
    int r;             /* free bits left in word p */
    int pend_c;        /* pending run */
    int64_t pend_l;
    uint64_t run[7];   /* running counts: [0] total, [c+1] symbol c */
    uint64_t at_blk[7];/* the same at the start of the current block */
} enc_t;

static void enc_reserve(enc_t *s, uint64_t need_words)
{
    if (need_words > s->cap) {
        uint64_t ncap = s->cap ? s->cap : 1024;
        while (ncap < need_words) ncap <<= 1;
        s->e->w = (uint64_t *)realloc(s->e->w, ncap * 8);
        memset(s->e->w + s->cap, 0, (ncap - s->cap) * 8);
        s->cap = ncap;
    }
}

static inline uint64_t tail_of(uint64_t shead)
{   /* the last block of every 2^23-word chunk leaves its final word unused (rld.h:66) */
    return shead + BLK_WORDS - ((shead + BLK_WORDS) % CHUNK_WORDS == 0 ? 2 : 1);
}

static void enc_begin(enc_t *s, orc_rld_t *e)
{
    memset(s, 0, sizeof(*s));
    s->e = e;
    enc_reserve(s, 2 * BLK_WORDS);
    s->shead = 0; s->p = 2; s->stail = tail_of(0); s->r = 64; /* rld.c:96-106 */
    s->pend_c = -1; s->pend_l = 0;
}

static void enc_new_block(enc_t *s) /* rld.c:111-134 */
{
    int i;
    uint64_t d0 = s->run[0] - s->at_blk[0];
    s->shead += BLK_WORDS;
    enc_reserve(s, s->shead + 2 * BLK_WORDS);
    if (d0 >= 0x8000) {
        uint32_t *q = (uint32_t *)(s->e->w + s->shead);
        for (i = 0; i < 7; ++i) q[i] = (uint32_t)(s->run[i] - s->at_blk[i]);
        q[0] |= 1u << 31;
        s->p = s->shead + 4;
    } else {
        uint16_t *q = (uint16_t *)(s->e->w + s->shead);
        for (i = 0; i < 7; ++i) q[i] = (uint16_t)(s->run[i] - s->at_blk[i]);
        s->p = s->shead + 2;
    }
    s->stail = tail_of(s->shead);
    s->r = 64;
    memcpy(s->at_blk, s->run, sizeof(s->run));
}

static void enc_emit(enc_t *s, int64_t l, int c) /* rld.c:47-53, 160-175 */
{
    int y = floor_log2_u32((uint32_t)l);            /* run lengths < 2^32 */
    int z = floor_log2_u32((uint32_t)(y + 1));
    int w = 2 * z + 1 + y + 3;
    uint64_t code = (((uint64_t)l ^ (1ull << y)) | (uint64_t)(y + 1) << y) << 3 | (uint64_t)c;
    uint64_t *W;
    if (w >= s->r && s->p == s->stail) enc_new_block(s);
    W = s->e->w;
    if (w > s->r) {
        w -= s->r;
        W[s->p++] |= code >> w;
        s->r = 64 - w;
        W[s->p] = code << s->r;
    } else {
        s->r -= w;
        W[s->p] |= code << s->r;
    }
    s->run[0] += l;
    s->run[c + 1] += l;
}

static void enc_push(enc_t *s, int64_t l, int c) /* rld.c:177-184 */
{
    if (l == 0) return;
    if (s->pend_c != c) {
        if (s->pend_l) enc_emit(s, s->pend_l, s->pend_c);
        s->pend_l = l; s->pend_c = c;
    } else s->pend_l += l;
}

static void build_frames(orc_rld_t *e) /* rld.c:186-224 */
{
    uint64_t n_blks = e->n_words / BLK_WORDS + 1;
    uint64_t last = e->n_words / BLK_WORDS * BLK_WORDS;
    uint64_t i, k, acc[6] = {0, 0, 0, 0, 0, 0};
    int j;
    e->ibits = floor_log2_u32((uint32_t)(e->mcnt[0] / n_blks)) + 4;
    e->n_frames = ((e->mcnt[0] + (1ull << e->ibits) - 1) >> e->ibits) + 1;
    e->frame = (uint64_t *)calloc(e->n_frames * 7, 8);
    for (i = BLK_WORDS, k = 1; i <= last; i += BLK_WORDS) {
        uint64_t h[7], sum = 0;
        hdr_read(e->w + i, h);
        for (j = 0; j < 6; ++j) acc[j] += h[j + 1], sum += acc[j];
        while (sum >= k << e->ibits) ++k;
        if (k < e->n_frames) {
            e->frame[k * 7] = i;
            for (j = 0; j < 6; ++j) e->frame[k * 7 + 1 + j] = acc[j];
        }
    }
    for (k = 1; k < e->n_frames; ++k)
        if (e->frame[k * 7] == 0)
            memcpy(e->frame + k * 7, e->frame + (k - 1) * 7, 7 * 8);
}

static void enc_finish(enc_t *s) /* rld.c:226-236 */
{
    orc_rld_t *e = s->e;
    int i;
    if (s->pend_l) enc_emit(s, s->pend_l, s->pend_c);
    enc_new_block(s);
    e->n_words = s->p;
    for (i = 0; i < 7; ++i) e->mcnt[i] = s->run[i];
    e->cnt[0] = 0;
    for (i = 1; i < 7; ++i) e->cnt[i] = e->cnt[i - 1] + s->run[i];
    build_frames(e);
}

orc_rld_t *orc_rld_from_bwt(const uint8_t *bwt, uint64_t n) /* build.c:11-31 */
{
    orc_rld_t *e = (orc_rld_t *)calloc(1, sizeof(orc_rld_t));
    enc_t s;
    uint64_t i, k;
    enc_begin(&s, e);
    for (i = 0; i < n; i += k) {
        for (k = 1; i + k < n && bwt[i + k] == bwt[i]; ++k) {}
        enc_push(&s, (int64_t)k, bwt[i]);
    }
    enc_finish(&s);
    return e;
}

/* ------------------------------------------------------------------------------------------
 * File I/O (rld.c:242-325)
 * ---------------------------------------------------------------------------------------- */
orc_rld_t *orc_rld_load(const char *fn)
{
    FILE *fp = fopen(fn, "rb");
    orc_rld_t *e;
    char magic[4];
    if (fp == 0) return 0;
    e = (orc_rld_t *)calloc(1, sizeof(orc_rld_t));
    if (fread(magic, 1, 4, fp) != 4) { fclose(fp); free(e); return 0; }
    if (memcmp(magic, "RLD\2", 4) != 0) {
        /* raw run-length bytes `len<<3 | sym` after a 4-byte magic (rld.c:295-308;
         * writer ropebwt.c:132-136): re-encode. */
        enc_t s;
        uint8_t *buf = (uint8_t *)malloc(0x10000);
        size_t l, i;
        enc_begin(&s, e);
        while ((l = fread(buf, 1, 0x10000, fp)) != 0)
            for (i = 0; i < l; ++i)
                if (buf[i] >> 3) enc_push(&s, buf[i] >> 3, buf[i] & 7);
        free(buf);
        fclose(fp);
        enc_finish(&s);
        return e;
    } else {
        uint32_t a;
        uint64_t h[3], n_blks;
        int i;
        if (fread(&a, 4, 1, fp) != 1 || (a >> 16) != 6 || (a & 0xffff) != 3) goto fail; /* DNA, sbits=3 only */
        if (fread(h, 8, 3, fp) != 3) goto fail;
        e->n_words = h[1] / 8; e->n_frames = h[2];
        if (fread(e->mcnt + 1, 8, 6, fp) != 6) goto fail;
        e->cnt[0] = 0;
        for (i = 1; i < 7; ++i) e->cnt[i] = e->cnt[i - 1] + e->mcnt[i];
        e->mcnt[0] = e->cnt[6];
        e->w = (uint64_t *)calloc(e->n_words + BLK_WORDS, 8);
        if (fread(e->w, 8, e->n_words, fp) != e->n_words) goto fail;
        e->frame = (uint64_t *)malloc(e->n_frames * 7 * 8);
        if (fread(e->frame, 56, e->n_frames, fp) != e->n_frames) goto fail;
        fclose(fp);
        n_blks = e->n_words / BLK_WORDS + 1;
        e->ibits = floor_log2_u32((uint32_t)(e->mcnt[0] / n_blks)) + 4;
        return e;
    }
fail:
    fclose(fp);
    orc_rld_free(e);
    return 0;
}

int orc_rld_dump(const orc_rld_t *e, const char *fn)
{
    FILE *fp = fopen(fn, "wb");
    uint32_t a = 6u << 16 | 3;
    uint64_t zero = 0, n_bytes = e->n_words * 8;
    if (fp == 0) return -1;
    fwrite("RLD\2", 1, 4, fp);
    fwrite(&a, 4, 1, fp);
    fwrite(&zero, 8, 1, fp);
    fwrite(&n_bytes, 8, 1, fp);
    fwrite(&e->n_frames, 8, 1, fp);
    fwrite(e->mcnt + 1, 8, 6, fp);
    fwrite(e->w, 8, e->n_words, fp);
    fwrite(e->frame, 56, e->n_frames, fp);
    fclose(fp);
    return 0;
}

void orc_rld_free(orc_rld_t *e)
{
    if (e == 0) return;
    free(e->w); free(e->frame); free(e);
}

uint64_t orc_rld_decode_all(const orc_rld_t *e, uint8_t *out) /* cmd.c:90-106 via rld_dec */
{
    uint64_t last = e->n_words / BLK_WORDS * BLK_WORDS, i, n = 0;
    for (i = 0; i < last; i += BLK_WORDS) {
        uint64_t bit = (i + hdr_words(e->w + i)) * 64, len;
        int c;
        while ((len = get_run(e->w, &bit, i + BLK_WORDS, &c)) != 0) {
            memset(out + n, c, len);
            n += len;
        }
    }
    return n;
}

/* ------------------------------------------------------------------------------------------
 * Rank (rld.c:352-492)
 * ---------------------------------------------------------------------------------------- */

/* rld.c:352-392: frame row, then walk block headers.  On return *blk = word offset of the
 * block holding position k, cnt[] = symbol counts before it, *z = their total; returns the
 * number of symbols through the end of that block. */
static inline uint64_t locate(const orc_rld_t *e, uint64_t k, uint64_t cnt[6], uint64_t *z, uint64_t *blk)
{
    const uint64_t *f = e->frame + (k >> e->ibits) * 7;
    uint64_t b = f[0], sum = 0, h[7];
    int j;
    for (j = 0; j < 6; ++j) sum += (cnt[j] = f[j + 1]);
    for (;;) {
        hdr_read(e->w + b + BLK_WORDS, h); /* next block's header = size of block b */
        if (sum + h[0] > k) break;
        for (j = 0; j < 6; ++j) cnt[j] += h[j + 1];
        sum += h[0];
        b += BLK_WORDS;
    }
    *z = sum; *blk = b;
    return sum + h[0];
}

int orc_rank1a(const orc_rld_t *e, uint64_t k, uint64_t ok[6]) /* rld.c:424-446 */
{
    uint64_t z, blk, bit, len;
    int a = -1;
    if (k == (uint64_t)-1) { memset(ok, 0, 48); return -1; }
    ++tl_counters.rank1a;
    locate(e, k, ok, &z, &blk);
    bit = (blk + hdr_words(e->w + blk)) * 64;
    ++k; /* position -> length */
    for (;;) {
        len = get_run(e->w, &bit, blk + BLK_WORDS, &a);
        if (z + len >= k) break;
        z += len; ok[a] += len;
    }
    ok[a] += k - z;
    return a;
}

void orc_rank2a(const orc_rld_t *e, uint64_t k, uint64_t l, uint64_t ok[6], uint64_t ol[6]) /* rld.c:457-492 */
{
    uint64_t z, y, blk, bit, len;
    int a = -1;
    ++tl_counters.rank2a;
    if (k == (uint64_t)-1) {
        memset(ok, 0, 48);
        orc_rank1a(e, l, ol);
        --tl_counters.rank1a; /* accounted as the rank2a's single block */
        return;
    }
    y = locate(e, k, ok, &z, &blk);
    bit = (blk + hdr_words(e->w + blk)) * 64;
    ++k;
    for (;;) {
        len = get_run(e->w, &bit, blk + BLK_WORDS, &a);
        if (z + len >= k) break;
        z += len; ok[a] += len;
    }
    if (y > l) { /* l lives in the same block: keep decoding */
        ++l;
        memcpy(ol, ok, 48);
        ok[a] += k - z;
        if (z + len < l) {
            z += len; ol[a] += len;
            for (;;) {
                len = get_run(e->w, &bit, blk + BLK_WORDS, &a);
                if (z + len >= l) break;
                z += len; ol[a] += len;
            }
        }
        ol[a] += l - z;
    } else {
        ok[a] += k - z;
        ++tl_counters.rank2a_spill;
        orc_rank1a(e, l, ol);
        --tl_counters.rank1a; /* accounted as a spill */
    }
}
