/* correct_cmd.c -- `fermi correct` (cmd.c:253-291 -> fm6_ec_correct, correct.c:305-456).
 *
 * Phase 1, the k-mer harvest that is 90 % of the reference's run time (fm6_traverse + ec_collect,
 * correct.c:35-87), runs on the GPU (fmd_kmer_collect).  Phase 2 never touches the index: it is a
 * best-first search over look-ups in the harvested table (ec_fix1 / ec_fix, correct.c:121-256) and
 * stays on the host; it is restated here so that corrected bases, qualities, header numbers and
 * read filtering come out byte-identical (heap discipline of ksort.h:125-146 included, because
 * ties between equal-score states are broken by heap position).
 */
#include <ctype.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "fmd_host.h"

static double now_s(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec; }

#define MAX_KMER      27   /* correct.c:303 */
#define RATIO_FACTOR  10   /* correct.c:112-119 */
#define DIFF_FACTOR   13
#define MAX_HEAP     256
#define MAX_SC_DIFF   60
#define MAX_QUAL      40
#define MISS_PENALTY  10
#define MIN_OCC        5
#define MIN_OCC_RATIO 0.8
#define BATCH_SIZE 1000000 /* correct.c:281 */

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
    t->key = (uint32_t *)fmdh_big_alloc((n + 1) * 4); t->val = (uint8_t *)fmdh_big_alloc(n + 1);
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
static int ec_fix1(const fmdh_ecopt_t *opt, const solid_t *solid, int len, char *s, char *qual, fix_t *fa, uint64_t *n_query)
{
    int i, q, l, shift = (opt->w - 1) << 1, n_rst = 0, qsum, no_hits = 1, score_diff;
    st_t z, rst[2];
    if (len <= opt->w) return 0xffff;
    fa->hn = fa->sn = 0;
    z.x = z.y = 0;
    for (i = len - 1, l = 0; i > 0 && l < opt->w; --i) { /* the initial k-mer */
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
                if ((v & 7) <= 0 && opt->step > 1) {
                    while (i0 > 0) {
                        int64_t k2;
                        for (i = (int)(z.y & 0xffff) - 1, l = 0; i >= 1 && l < opt->step && s[i] < 5; --i, ++l)
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

/* ec_fix for one read (correct.c:232-253): reverse-complement strand first, then forward */
static int fix_read(const fmdh_ecopt_t *opt, const solid_t *solid, char *seq, char *qual, fix_t *fa, char **buf, size_t *buf_m, uint64_t *n_query)
{
    int l = (int)strlen(seq), j, ret0, ret1, n_lower, info;
    if ((size_t)l + 1 > *buf_m) { *buf_m = (size_t)l + 256; *buf = (char *)realloc(*buf, *buf_m); }
    char *s = *buf;
    for (j = 0; j < l; ++j) s[j] = (char)fmdh_nt6[(unsigned char)seq[j]];
    revcomp(l, s); rev(l, qual);
    ret0 = ec_fix1(opt, solid, l, s, qual, fa, n_query);
    rev(l, qual); revcomp(l, s);
    if (ret0 != 0xffff) {
        ret1 = ec_fix1(opt, solid, l, s, qual, fa, n_query);
        info = ((ret0 & 0xffff) + (ret1 & 0xffff)) | (ret0 >> 18 < ret1 >> 18 ? ret0 >> 18 : ret1 >> 18) << 18;
        if ((ret0 >> 17 & 1) && (ret1 >> 17 & 1)) info |= 1 << 16;
    } else info = ret0;
    for (j = 0, n_lower = 0; j < l; ++j) {
        seq[j] = fmdh_nt6[(unsigned char)seq[j]] == (uint8_t)s[j] ? (char)toupper(seq[j]) : "$acgtn"[(int)s[j]];
        if (islower((unsigned char)seq[j])) ++n_lower, qual[j] = 36;
    }
    if ((double)n_lower / l > opt->max_corr) info |= 1 << 16;
    if (info >> 18 <= 10) info |= 1 << 16;
    return info;
}

/* ec_fix worker threads (correct.c:281-290, `-t`): the reads of a batch are dealt to the T threads in blocks; every
 * read is corrected independently against the read-only table, so the output does not depend on T */
#include <pthread.h>
#define FIX_BLOCK 32
static int g_fix_threads = 1;
void fmdh_correct_set_threads(int n) { g_fix_threads = n > 0 ? n : 1; }

typedef struct { const fmdh_ecopt_t *opt; const solid_t *solid; char **seqs, **quals; int *info; size_t nb; int start, step; uint64_t n_query; } fixjob_t;
static void *fix_worker(void *d)
{
    fixjob_t *w = (fixjob_t *)d;
    fix_t fa; memset(&fa, 0, sizeof(fa));
    char *buf = 0; size_t buf_m = 0, i, j;
    uint64_t n_query = 0;   /* counted locally: the jobs sit side by side in memory and the count moves at every look-up */
    /* blocks of FIX_BLOCK consecutive reads, dealt round-robin: neighbours in the batch buffer stay with one thread */
    for (i = (size_t)w->start * FIX_BLOCK; i < w->nb; i += (size_t)w->step * FIX_BLOCK)
        for (j = i; j < i + FIX_BLOCK && j < w->nb; ++j)
            w->info[j] = fix_read(w->opt, w->solid, w->seqs[j], w->quals[j], &fa, &buf, &buf_m, &n_query);
    w->n_query = n_query;
    free(fa.heap); free(fa.stack); free(buf);
    return 0;
}
static void fix_batch(const fmdh_ecopt_t *opt, const solid_t *solid, char **seqs, char **quals, int *info, size_t nb, uint64_t *n_query)
{
    int T = g_fix_threads, t;
    if ((size_t)T > (nb + FIX_BLOCK - 1) / FIX_BLOCK) T = nb ? (int)((nb + FIX_BLOCK - 1) / FIX_BLOCK) : 1;
    pthread_t *tid = (pthread_t *)calloc((size_t)T, sizeof(pthread_t));
    fixjob_t *w = (fixjob_t *)calloc((size_t)T, sizeof(fixjob_t));
    for (t = 0; t < T; ++t) {
        w[t].opt = opt; w[t].solid = solid; w[t].seqs = seqs; w[t].quals = quals; w[t].info = info; w[t].nb = nb; w[t].start = t; w[t].step = T;
        if (t + 1 < T) pthread_create(&tid[t], 0, fix_worker, &w[t]);
    }
    fix_worker(&w[T - 1]);
    for (t = 0; t + 1 < T; ++t) pthread_join(tid[t], 0);
    for (t = 0; t < T; ++t) *n_query += w[t].n_query;
    free(tid); free(w);
}

/* Phase 2 alone: correct the reads of fq_path against a harvested table (opt->w must be set). */
/* Three-stage pipeline over batches of BATCH_SIZE reads (correct.c:372-441 reads, corrects and prints one
 * batch after the other): the caller's thread parses batch k+1 while the fix workers are on batch k and a
 * writer thread prints batch k-1.  A ring of three batches; each slot goes free -> filled -> fixed -> free. */
typedef struct {
    char *buf; size_t buf_l, buf_m;      /* sequences and qualities of the batch, back to back, NUL-terminated */
    size_t *off;                         /* 2 * nb offsets into buf */
    char **seqs, **quals; int *info;
    size_t nb; uint64_t pre_id;
    int last, state;                     /* state: 0 free, 1 filled, 2 fixed */
} ecbatch_t;
typedef struct {
    const fmdh_ecopt_t *opt; const solid_t *solid; FILE *out;
    ecbatch_t b[3];
    pthread_mutex_t mu; pthread_cond_t cv;
    uint64_t n_query;
    double t_fix, t_write;               /* busy time of the two worker stages (FMD_TIMING) */
} ecpipe_t;

static void pipe_wait(ecpipe_t *p, ecbatch_t *b, int want)
{
    pthread_mutex_lock(&p->mu);
    while (b->state != want) pthread_cond_wait(&p->cv, &p->mu);
    pthread_mutex_unlock(&p->mu);
}
static void pipe_set(ecpipe_t *p, ecbatch_t *b, int st)
{
    pthread_mutex_lock(&p->mu);
    b->state = st;
    pthread_cond_broadcast(&p->cv);
    pthread_mutex_unlock(&p->mu);
}
static void *pipe_fixer(void *d)
{
    ecpipe_t *p = (ecpipe_t *)d;
    for (unsigned k = 0;; ++k) {
        ecbatch_t *b = &p->b[k % 3];
        pipe_wait(p, b, 1);
        const double t0 = now_s();
        for (size_t i = 0; i < b->nb; ++i) { b->seqs[i] = b->buf + b->off[2 * i]; b->quals[i] = b->buf + b->off[2 * i + 1]; }
        fix_batch(p->opt, p->solid, b->seqs, b->quals, b->info, b->nb, &p->n_query);
        p->t_fix += now_s() - t0;
        const int last = b->last;
        pipe_set(p, b, 2);
        if (last) return 0;
    }
}
static void *pipe_writer(void *d)
{
    ecpipe_t *p = (ecpipe_t *)d;
    const fmdh_ecopt_t *opt = p->opt;
    FILE *out = p->out;
    char hdr[64];
    for (unsigned kb = 0;; ++kb) {
        ecbatch_t *b = &p->b[kb % 3];
        pipe_wait(p, b, 2);
        const int *info = b->info;
        const double t0 = now_s();
        for (size_t a = 0; a < b->nb; ++a) {
            const uint64_t k = b->pre_id + a;
            int is_bad = 0;
            if (opt->is_paired) { /* a pair is bad when either end is (correct.c:401-410, one thread) */
                if (info[a] >> 16 & 1) is_bad = 1;
                else if (k & 1) { if (a >= 1 && (info[a - 1] >> 16 & 1)) is_bad = 1; }
                else if (a + 1 < b->nb && (info[a + 1] >> 16 & 1)) is_bad = 1;
            } else if (info[a] >> 16 & 1) is_bad = 1;
            if (!is_bad || opt->keep_bad) {
                int tmp = (int)strlen(b->seqs[a]);
                if (opt->trim_l && opt->trim_l < tmp) tmp = opt->trim_l;
                const int hl = snprintf(hdr, sizeof(hdr), "@%lld%c%d%c%d\n", (long long)(opt->is_paired ? k >> 1 : k), opt->is_paired ? ' ' : '_',
                                        info[a] & 0xffff, opt->is_paired ? ' ' : '_', info[a] >> 18);
                fwrite(hdr, 1, (size_t)hl, out);
                fwrite(b->seqs[a], 1, (size_t)tmp, out); fwrite("\n+\n", 1, 3, out); fwrite(b->quals[a], 1, (size_t)tmp, out); fputc('\n', out);
            }
        }
        const int last = b->last;
        p->t_write += now_s() - t0;
        pipe_set(p, b, 0);
        if (last) return 0;
    }
}
static void batch_put(ecbatch_t *b, const char *s, size_t l, size_t slot)
{
    if (b->buf_l + l + 1 > b->buf_m) { while (b->buf_l + l + 1 > b->buf_m) b->buf_m = b->buf_m ? b->buf_m << 1 : 1 << 20; b->buf = (char *)realloc(b->buf, b->buf_m); }
    b->off[slot] = b->buf_l;
    memcpy(b->buf + b->buf_l, s, l); b->buf[b->buf_l + l] = 0;
    b->buf_l += l + 1;
}

int fmdh_correct_reads(const fmdh_ecopt_t *opt, int suf_len, uint64_t n, const uint32_t *bucket, const uint32_t *key, const uint8_t *val,
                       const char *fq_path, FILE *out)
{
    solid_t solid;
    const int timing = getenv("FMD_TIMING") != 0;
    const double t_begin = now_s();
    double t_read = 0;
    memset(&solid, 0, sizeof(solid));
    if (solid_build(&solid, suf_len, n, bucket, key, val)) { fprintf(stderr, "[E::%s] out of memory\n", __func__); return 1; }
    const double t_table = now_s() - t_begin;
    fmdh_seqio_t *io = fmdh_seq_open(fq_path);
    if (!io) { fprintf(stderr, "[E::%s] cannot open `%s'\n", __func__, fq_path); solid_free(&solid); return 1; }
    ecpipe_t p;
    memset(&p, 0, sizeof(p));
    p.opt = opt; p.solid = &solid; p.out = out;
    pthread_mutex_init(&p.mu, 0); pthread_cond_init(&p.cv, 0);
    for (int i = 0; i < 3; ++i) {
        p.b[i].off = (size_t *)malloc(2 * BATCH_SIZE * sizeof(size_t));
        p.b[i].seqs = (char **)malloc(BATCH_SIZE * sizeof(char *)); p.b[i].quals = (char **)malloc(BATCH_SIZE * sizeof(char *));
        p.b[i].info = (int *)malloc(BATCH_SIZE * sizeof(int));
    }
    pthread_t t_fix, t_out;
    pthread_create(&t_fix, 0, pipe_fixer, &p);
    pthread_create(&t_out, 0, pipe_writer, &p);
    uint64_t id = 0;
    char *q15 = 0; size_t q15_m = 0;
    for (unsigned kb = 0;; ++kb) { /* batches of BATCH_SIZE reads, output in input order */
        ecbatch_t *b = &p.b[kb % 3];
        pipe_wait(&p, b, 0);
        const double t0 = now_s();
        b->nb = 0; b->buf_l = 0; b->pre_id = id; b->last = 0;
        while (b->nb < BATCH_SIZE) {
            const int ret = fmdh_seq_read(io);
            if (ret < 0) { b->last = 1; break; }
            batch_put(b, fmdh_seq_bases(io), (size_t)ret, 2 * b->nb);
            if (fmdh_seq_qual(io) == 0) { /* no quality: phred 15 (correct.c:431-436) */
                if ((size_t)ret + 1 > q15_m) { q15_m = (size_t)ret * 2 + 64; q15 = (char *)realloc(q15, q15_m); memset(q15, 33 + 15, q15_m); }
                batch_put(b, q15, (size_t)ret, 2 * b->nb + 1);
            } else batch_put(b, fmdh_seq_qual(io), (size_t)ret, 2 * b->nb + 1);
            ++b->nb; ++id;
        }
        const int last = b->last;
        t_read += now_s() - t0;
        pipe_set(&p, b, 1);
        if (last) break;
    }
    pthread_join(t_fix, 0); pthread_join(t_out, 0);
    if (timing) fprintf(stderr, "[M::%s] table %.3f s; pipeline %.3f s with stages busy for: parse %.3f s, fix %.3f s, print %.3f s\n", __func__, t_table,
                        now_s() - t_begin - t_table, t_read, p.t_fix, p.t_write);
    for (int i = 0; i < 3; ++i) { free(p.b[i].buf); free(p.b[i].off); free(p.b[i].seqs); free(p.b[i].quals); free(p.b[i].info); }
    free(q15);
    pthread_mutex_destroy(&p.mu); pthread_cond_destroy(&p.cv);
    fmdh_seq_close(io);
    solid_free(&solid);
    return 0;
}

int fmdh_correct_kmer(uint64_t n_symbols) /* the automatic k-mer length, correct.c:313-318 */
{
    int w = (int)(log((double)n_symbols) / log(4) + 8.499);
    return w >= MAX_KMER ? MAX_KMER : w;
}

int fmdh_correct(const char *fmd_path, const char *fq_path, int device, fmdh_ecopt_t *opt, FILE *out)
{
    fmd_dev_t *d = 0;
    fmd_info_t finfo;
    const int timing = getenv("FMD_TIMING") != 0; /* phase times on stderr */
    double t0 = now_s(), t1;
    int rc = fmd_dev_open_file(device, fmd_path, &d);
    if (rc) { fprintf(stderr, "[E::%s] cannot load `%s': %s\n", __func__, fmd_path, fmd_strerror(rc)); return 1; }
    fmd_dev_info(d, &finfo);
    if (opt->w < 0) opt->w = fmdh_correct_kmer(finfo.mcnt[0]);
    const int suf_len = opt->w > 15 ? opt->w - 15 : 1; /* correct.c:319 */
    /* phase 1 on the GPU */
    uint32_t *bucket = 0, *key = 0; uint8_t *val = 0; uint64_t n = 0; int64_t cnt[2];
    if (timing) { t1 = now_s(); fprintf(stderr, "[M::%s] index load + transcode: %.3f s\n", __func__, t1 - t0); t0 = t1; }
    rc = fmd_kmer_collect(d, opt->w, opt->min_occ, suf_len, &bucket, &key, &val, &n, cnt);
    fmd_dev_close(d);
    if (rc) { fprintf(stderr, "[E::%s] k-mer harvest failed: %s\n", __func__, fmd_strerror(rc)); return 1; }
    if (timing) { t1 = now_s(); fprintf(stderr, "[M::%s] harvest of %llu solid %d-mers (GPU + copy): %.3f s\n", __func__, (unsigned long long)n, opt->w + 1, t1 - t0); t0 = t1; }
    rc = fmdh_correct_reads(opt, suf_len, n, bucket, key, val, fq_path, out);
    if (timing) { t1 = now_s(); fprintf(stderr, "[M::%s] table build + correction + output: %.3f s\n", __func__, t1 - t0); t0 = t1; }
    fmd_host_free(bucket); fmd_host_free(key); fmd_host_free(val);
    return rc;
}
