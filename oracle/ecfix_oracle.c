/* oracle/ecfix_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU statement of the correction pass of `fermi correct`: ec_fix1 / ec_fix (correct.c:121-256) against a table of
 * solid k-mers given as (bucket, key, val) triples (what ec_collect leaves in solid[bucket], correct.c:35-87).  It
 * follows the reference step for step on purpose -- it is the checker of the product's GPU kernel
 * (fermi_amd/csrc/fmd_ecfix.hip), which is organised differently (one lane per read, a hash table, its own queue):
 *   save_state   correct.c:98-110      ec_fix1   correct.c:121-220      ec_fix (two strands)   correct.c:232-246
 *   heap         ksort.h:125-146 with ku128_ylt (mag.c:22)
 * Pinned by tests/test_oracle_golden.py: with the golden table of tiny.fmd it reproduces `fermi correct -t1`'s
 * tiny.ec.fq byte for byte.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define RATIO_FACTOR  10   /* correct.c:112-119 */
#define DIFF_FACTOR   13
#define MAX_HEAP     256
#define MAX_SC_DIFF   60
#define MAX_QUAL      40
#define MISS_PENALTY  10
#define MIN_OCC        5
#define MIN_OCC_RATIO 0.8

/* ---- the solid k-mer table: per suffix bucket, keys sorted by (key >> 2) ------------------- */
typedef struct {
    int suf_len;
    uint64_t suf_num;
    uint64_t *off;        /* suf_num + 1 */
    uint32_t *key; uint8_t *val;
    int key_bits;         /* every key >> 2 is below 2^key_bits */
} solid_t;

static int solid_build(solid_t *t, int suf_len, uint64_t n, const uint32_t *bucket, const uint32_t *key, const uint8_t *val)
{
    uint64_t i, b;
    t->suf_len = suf_len; t->suf_num = 1ull << (2 * suf_len);
    t->off = (uint64_t *)calloc(t->suf_num + 1, 8);
    t->key = (uint32_t *)malloc((n + 1) * 4); t->val = (uint8_t *)malloc(n + 1);
    if (!t->off || !t->key || !t->val) return -1;
    for (i = 0; i < n; ++i) ++t->off[bucket[i] + 1];
    for (b = 0; b < t->suf_num; ++b) t->off[b + 1] += t->off[b];
    uint64_t *cur = (uint64_t *)malloc(t->suf_num * 8);
    if (!cur) return -1;
    memcpy(cur, t->off, t->suf_num * 8);
    for (i = 0; i < n; ++i) { uint64_t p = cur[bucket[i]]++; t->key[p] = key[i]; t->val[p] = val[i]; }
    free(cur);
    for (b = 0; b < t->suf_num; ++b) { /* insertion sort inside a bucket (buckets are small) */
        uint64_t lo = t->off[b], hi = t->off[b + 1], a, c;
        for (a = lo + 1; a < hi; ++a) {
            uint32_t k = t->key[a]; uint8_t v = t->val[a];
            for (c = a; c > lo && (t->key[c - 1] >> 2) > (k >> 2); --c) { t->key[c] = t->key[c - 1]; t->val[c] = t->val[c - 1]; }
            t->key[c] = k; t->val[c] = v;
        }
    }
    uint32_t top = 0;
    for (i = 0; i < n; ++i) top |= t->key[i] >> 2;
    t->key_bits = 1;
    while (t->key_bits < 30 && (top >> t->key_bits)) ++t->key_bits;
    return 0;
}
/* kh_get(solid, h, q): the entry whose key agrees with q above the low two bits (correct.c:17-20).
 * The keys of a bucket are the remaining bases of the k-mers that end in the bucket's suffix: sorted and close
 * to uniform, so the position is estimated first (q / 2^key_bits of the way through the bucket) and the search
 * gallops from there -- one or two cache lines instead of the ~6 misses of a bisection over ~1000 keys.  With
 * 64 threads the fix pass is bound by the host's rate of random DRAM accesses, not by its cores. */
static inline int64_t solid_get(const solid_t *t, uint64_t x)
{
    const uint64_t b = x & (t->suf_num - 1);
    const uint32_t q = (uint32_t)(x >> (t->suf_len << 1) << 2) >> 2;
    uint64_t lo = t->off[b], hi = t->off[b + 1];
    if (lo == hi) return -1;
    uint64_t i = lo + (uint64_t)(((unsigned __int128)q * (hi - lo)) >> t->key_bits);
    if (i >= hi) i = hi - 1;
    uint32_t k = t->key[i] >> 2;
    if (k == q) return (int64_t)i;
    if (k < q) { /* the entry, if any, lies in (i, hi): double the step until a key >= q */
        uint64_t step = 1;
        lo = i + 1;
        for (;;) {
            const uint64_t j = i + step;
            if (j >= hi) break;
            k = t->key[j] >> 2;
            if (k == q) return (int64_t)j;
            if (k > q) { hi = j; break; }
            lo = j + 1; step <<= 1;
        }
    } else {     /* in [lo, i) */
        uint64_t step = 1;
        hi = i;
        for (;;) {
            if (i < lo + step) break;
            const uint64_t j = i - step;
            k = t->key[j] >> 2;
            if (k == q) return (int64_t)j;
            if (k < q) { lo = j + 1; break; }
            hi = j; step <<= 1;
        }
    }
    while (lo < hi) {
        uint64_t mid = (lo + hi) >> 1;
        k = t->key[mid] >> 2;
        if (k < q) lo = mid + 1; else if (k > q) hi = mid; else return (int64_t)mid;
    }
    return -1;
}
static void solid_free(solid_t *t) { free(t->off); free(t->key); free(t->val); }

/* ---- best-first search state (correct.c:92-110) -------------------------------------------- */
typedef struct { uint64_t x, y; } st_t;
typedef struct {
    st_t *heap; size_t hn, hm;
    uint64_t *stack; size_t sn, sm;
} fix_t;

static inline int st_lt(const st_t *a, const st_t *b) { return (int64_t)a->y > (int64_t)b->y; } /* mag.c:22 */
static void heap_up(size_t n, st_t *l)                 /* ksort.h:136-146 */
{
    size_t i, k = n - 1;
    st_t tmp = l[k];
    while (k) {
        i = (k - 1) >> 1;
        if (st_lt(&tmp, &l[i])) break;
        l[k] = l[i]; k = i;
    }
    l[k] = tmp;
}
static void heap_down(size_t i, size_t n, st_t *l)     /* ksort.h:125-135 */
{
    size_t k = i;
    st_t tmp = l[i];
    while ((k = (k << 1) + 1) < n) {
        if (k != n - 1 && st_lt(&l[k], &l[k + 1])) ++k;
        if (st_lt(&l[k], &tmp)) break;
        l[i] = l[k]; i = k;
    }
    l[i] = tmp;
}
static void push_stack(fix_t *f, uint64_t v)
{
    if (f->sn == f->sm) { f->sm = f->sm ? f->sm << 1 : 256; f->stack = (uint64_t *)realloc(f->stack, f->sm * 8); }
    f->stack[f->sn++] = v;
}
static void push_heap(fix_t *f, st_t v)
{
    if (f->hn == f->hm) { f->hm = f->hm ? f->hm << 1 : 256; f->heap = (st_t *)realloc(f->heap, f->hm * sizeof(st_t)); }
    f->heap[f->hn++] = v;
}
static void save_state(fix_t *f, const st_t *p, int c, int score, int shift, int has_match)
{
    st_t w;
    if (score < 0) score = 0;
    if (c >= 4) c = 0;
    w.x = (uint64_t)c << shift | p->x >> 2;
    /* y: score:16 | position in stack:32 | position in read:16 */
    w.y = (uint64_t)((p->y >> 48) + (uint64_t)score) << 48 | (uint64_t)f->sn << 16 | ((p->y & 0xffff) - 1);
    /* stack element: read position:32 | base:3 | has_match:1 | parent position in stack:28 */
    push_stack(f, ((p->y & 0xffff) - 1) << 32 | (uint32_t)c << 29 | (uint32_t)has_match << 28 | (uint32_t)(p->y >> 16));
    push_heap(f, w);
    heap_up(f->hn, f->heap);
}

/* correct.c:121-220.  s: nt6 bases (modified in place), qual: phred+33 (modified in place). */
static int ec_fix1(int w, int step, const solid_t *solid, int len, char *s, char *qual, fix_t *fa, uint64_t *n_query)
{
    int i, q, l, shift = (w - 1) << 1, n_rst = 0, qsum, no_hits = 1, score_diff;
    st_t z, rst[2];
    if (len <= w) return 0xffff;
    fa->hn = fa->sn = 0;
    z.x = z.y = 0;
    for (i = len - 1, l = 0; i > 0 && l < w; --i) { /* the initial k-mer */
        if (s[i] == 5) z.x = 0, l = 0;
        else z.x = (uint64_t)(s[i] - 1) << shift | z.x >> 2, ++l;
    }
    if (i == 0) return 0xffff;
    push_stack(fa, 0);
    z.y = (uint64_t)(i + 1);
    push_heap(fa, z);
    while (fa->hn) {
        int64_t k;
        z = fa->heap[0];
        fa->heap[0] = fa->heap[--fa->hn];
        heap_down(0, fa->hn, fa->heap);
        if ((z.y & 0xffff) == 0) {
            rst[n_rst++] = z;
            if (n_rst == 2) break;
            continue;
        }
        if (n_rst && (int)(z.y >> 48) > (int)(rst[0].y >> 48) + MAX_SC_DIFF) break;
        i = (int)(z.y & 0xffff) - 1;
        q = qual[i] - 33 < MAX_QUAL ? qual[i] - 33 : MAX_QUAL;
        if (q < 3) q = 3;
        k = solid_get(solid, z.x);
        ++*n_query;
        if (k >= 0) { /* this (k+1)-mer is solid */
            no_hits = 0;
            if (s[i] != (int)(solid->key[k] & 3) + 1) { /* the read base differs from the best base */
                int v = solid->val[k];
                int tmp, penalty, max = (v & 7) ? (v & 7) * (v >> 3) : v >> 3;
                penalty = (max - (v & 7)) * DIFF_FACTOR;
                if (max - (v & 7) < 1) penalty = 1;
                tmp = (v & 7) ? (v >> 3) * RATIO_FACTOR : 10000;
                if (tmp < penalty) penalty = tmp;
                tmp = (7 - (v & 7)) * DIFF_FACTOR;
                if (tmp < penalty) penalty = tmp;
                if (penalty < 1) penalty = 1;
                if (s[i] != 5 && (fa->hn + 2 <= MAX_HEAP || penalty < q))
                    save_state(fa, &z, s[i] - 1, penalty, shift, 1);                     /* the read path */
                if (s[i] == 5 || fa->hn + 2 <= MAX_HEAP || penalty > q)
                    save_state(fa, &z, (int)(solid->key[k] & 3), q, shift, 1);           /* the stack path */
            } else { /* the read base is the best base; try to jump ahead (correct.c:177-199) */
                st_t z0 = z;
                int i0 = i;
                int v = solid->val[k], occ_last = (v & 7) ? (v & 7) * ((v >> 3) + 1) : v >> 3;
                if ((v & 7) <= 0 && step > 1) {
                    while (i0 > 0) {
                        int64_t k2;
                        for (i = (int)(z.y & 0xffff) - 1, l = 0; i >= 1 && l < step && s[i] < 5; --i, ++l)
                            z.x = (uint64_t)(s[i] - 1) << shift | z.x >> 2;
                        if (s[i] == 5) break;
                        k2 = solid_get(solid, z.x);
                        ++*n_query;
                        if (k2 >= 0 && s[i] == (int)(solid->key[k2] & 3) + 1) {
                            int v2 = solid->val[k2], occ = (v2 & 7) ? (v2 & 7) * ((v2 >> 3) + 1) : v2 >> 3;
                            if ((v2 & 7) <= 1 && occ >= MIN_OCC && (double)occ / occ_last >= MIN_OCC_RATIO) {
                                z.y = z.y >> 16 << 16 | (uint64_t)(i + 1);
                                z0 = z; i0 = i;
                                occ_last = occ;
                            } else break;
                        } else break;
                    }
                }
                save_state(fa, &z0, s[i0] - 1, 0, shift, 1);
            }
        } else save_state(fa, &z, s[i] - 1, MISS_PENALTY + (MAX_QUAL - q), shift, 0);
    }
    score_diff = n_rst == 1 ? MAX_SC_DIFF : (int)(rst[1].y >> 48) - (int)(rst[0].y >> 48);
    if (score_diff >= MAX_SC_DIFF) score_diff = MAX_SC_DIFF;
    if (rst[0].y >> 48 == 0) return score_diff << 18; /* no corrections */
    qsum = 0; l = (int)(uint32_t)(rst[0].y >> 16);
    while (l) { /* backtrack */
        i = (int)(fa->stack[l] >> 32);
        if ((uint32_t)(s[i] - 1) != (uint32_t)fa->stack[l] >> 29) {
            s[i] = (char)(((uint32_t)fa->stack[l] >> 29) + 1);
            qsum += qual[i] - 33;
        } else if (((uint32_t)fa->stack[l] >> 28 & 1) && qual[i] < 37) qual[i] = 37;
        l = (int)((uint32_t)fa->stack[l] << 4 >> 4);
    }
    return qsum | score_diff << 18 | no_hits << 17;
}

static void rev(int l, char *s) { int i; for (i = 0; i < l >> 1; ++i) { char t = s[i]; s[i] = s[l - 1 - i]; s[l - 1 - i] = t; } }
static void revcomp(int l, char *s)
{
    int i;
    rev(l, s);
    for (i = 0; i < l; ++i) s[i] = (char)((s[i] >= 1 && s[i] <= 4) ? 5 - s[i] : s[i]);
}


void *orc_ectab_new(int suf_len, uint64_t n, const uint32_t *bucket, const uint32_t *key, const uint8_t *val)
{
    solid_t *t = (solid_t *)calloc(1, sizeof(solid_t));
    if (!t) return 0;
    if (solid_build(t, suf_len, n, bucket, key, val)) { solid_free(t); free(t); return 0; }
    return t;
}
void orc_ectab_free(void *t) { if (t) { solid_free((solid_t *)t); free(t); } }

/* ec_fix (correct.c:232-246) for n reads: seqs = nt6 codes, quals = phred + 33, read i = bytes [off[i], off[i+1]);
 * both rewritten in place, info[i] = the value at correct.c:246 (before the lower-case count). */
void orc_ecfix_batch(void *tab, int w, int step, size_t n, uint8_t *seqs, uint8_t *quals, const uint64_t *off, int32_t *info)
{
    fix_t fa;
    uint64_t n_query = 0;
    size_t i;
    memset(&fa, 0, sizeof(fa));
    for (i = 0; i < n; ++i) {
        char *s = (char *)seqs + off[i], *q = (char *)quals + off[i];
        const int l = (int)(off[i + 1] - off[i]);
        int ret0, ret1;
        revcomp(l, s); rev(l, q);
        ret0 = ec_fix1(w, step, (const solid_t *)tab, l, s, q, &fa, &n_query);
        rev(l, q); revcomp(l, s);
        if (ret0 != 0xffff) {
            ret1 = ec_fix1(w, step, (const solid_t *)tab, l, s, q, &fa, &n_query);
            info[i] = ((ret0 & 0xffff) + (ret1 & 0xffff)) | (ret0 >> 18 < ret1 >> 18 ? ret0 >> 18 : ret1 >> 18) << 18;
            if ((ret0 >> 17 & 1) && (ret1 >> 17 & 1)) info[i] |= 1 << 16;
        } else info[i] = ret0;
    }
    free(fa.heap); free(fa.stack);
}
