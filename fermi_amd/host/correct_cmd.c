/* correct_cmd.c -- `fermi correct` (cmd.c:253-291 -> fm6_ec_correct, correct.c:305-456) on the GPU.
 *
 * Phase 1, the k-mer harvest that is 90 % of the reference's run time (fm6_traverse + ec_collect, correct.c:35-87):
 * fmd_kmer_collect.  Phase 2, the best-first correction of every read against the harvested table (ec_fix1 / ec_fix,
 * correct.c:121-246): fmd_ecfix_batch -- one lane per read, the table in a device hash table.  What is left for the
 * host is what the reference does around them: FASTQ parsing, the lower-case marking and the read filter
 * (correct.c:247-252), the pair rule and the printing (correct.c:396-425).
 *
 * Batches of BATCH_SIZE reads (correct.c:281) move through a ring of three slots: the caller's thread parses batch
 * k+1 while the GPU corrects batch k and a writer thread marks, filters and prints batch k-1.
 */
#include <ctype.h>
#include <fcntl.h>
#include <math.h>
#include <sys/stat.h>
#include <unistd.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "fmd_host.h"

static double now_s(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec; }

#define MAX_KMER      27   /* correct.c:303 */
#define BATCH_SIZE 1000000 /* correct.c:281 */

static int g_host_threads = 1;
void fmdh_correct_set_threads(int n) { g_host_threads = n > 0 ? n : 1; }   /* `-t`: threads of the marking pass (output independent of n) */

/* ---- one batch: the reads as given (ASCII), as the GPU sees them (nt6 codes) and their qualities ---------------- */
typedef struct {
    char *ascii; uint8_t *nt6, *qual;    /* three parallel byte arrays: read i = bytes [off[i], off[i+1]) of each */
    uint64_t *off;                       /* nb + 1 */
    int32_t *info;
    size_t nb, bytes, cap_bytes, cap_reads;
    uint64_t first_id;
    int last, state;                     /* state: 0 free, 1 parsed, 2 corrected */
} batch_t;

typedef struct {
    const fmdh_ecopt_t *opt; FILE *out;
    fmd_ectab_t *tab[FMDH_MAX_GPUS]; int n_tab;   /* one copy of the table per GPU: a batch is split over them */
    batch_t b[3];
    pthread_mutex_t mu; pthread_cond_t cv;
    int failed;               /* set by any of the three pipeline threads: atomic accesses only */
    double t_gpu, t_write;               /* busy time of the two worker stages (FMD_TIMING) */
} pipe_t;

static void slot_wait(pipe_t *p, batch_t *b, int want)
{
    pthread_mutex_lock(&p->mu);
    while (b->state != want) pthread_cond_wait(&p->cv, &p->mu);
    pthread_mutex_unlock(&p->mu);
}
static void slot_set(pipe_t *p, batch_t *b, int st)
{
    pthread_mutex_lock(&p->mu);
    b->state = st;
    pthread_cond_broadcast(&p->cv);
    pthread_mutex_unlock(&p->mu);
}

static int batch_room(batch_t *b, size_t more)
{
    if (b->bytes + more + 16 <= b->cap_bytes) return 0;
    size_t m = b->cap_bytes ? b->cap_bytes : (size_t)128 << 20;
    while (b->bytes + more + 16 > m) m <<= 1;
    char *a = (char *)realloc(b->ascii, m);
    if (a) b->ascii = a;
    uint8_t *n = (uint8_t *)realloc(b->nt6, m);
    if (n) b->nt6 = n;
    uint8_t *q = (uint8_t *)realloc(b->qual, m);
    if (q) b->qual = q;
    if (!a || !n || !q) return -1;
    b->cap_bytes = m;
    return 0;
}

static int batch_reads_room(batch_t *b, size_t n_reads)
{
    if (n_reads <= b->cap_reads) return 0;
    size_t m = b->cap_reads ? b->cap_reads : BATCH_SIZE;
    while (m < n_reads) m <<= 1;
    uint64_t *o = (uint64_t *)realloc(b->off, (m + 1) * sizeof(uint64_t));
    if (o) b->off = o;
    int32_t *f = (int32_t *)realloc(b->info, m * sizeof(int32_t));
    if (f) b->info = f;
    if (!o || !f) return -1;
    b->cap_reads = m;
    return 0;
}

/* a piece of a span parsed by seqpar.c into its place in a batch (reads r0 .., bytes o0 ..): one thread per piece */
typedef struct { batch_t *b; const fmdh_ppart_t *p; size_t r0, o0; } fill_t;
static void *fill_main(void *d)
{
    fill_t *f = (fill_t *)d;
    batch_t *b = f->b;
    const fmdh_ppart_t *p = f->p;
    size_t o = f->o0, i;
    memcpy(b->ascii + o, p->seq, p->bytes);
    for (i = 0; i < p->bytes; ++i) { const uint8_t q = (uint8_t)p->qual[i]; b->qual[o + i] = q ? q : 33 + 15; }   /* no quality: phred 15 (correct.c:431-436) */
    for (i = 0; i < p->n; ++i) { o += p->len[i]; b->off[f->r0 + i + 1] = o; }
    return 0;
}

/* ASCII -> nt6 of a byte range of a batch: the parser (one thread, the pipeline's critical stage) only copies; the codes the GPU
 * wants are made here, on the `-t` host threads, when the batch reaches the second stage */
typedef struct { batch_t *b; size_t lo, hi; } enc_t;
static void *enc_main(void *d)
{
    enc_t *e = (enc_t *)d;
    const unsigned char *a = (const unsigned char *)e->b->ascii;
    uint8_t *o = e->b->nt6;
    for (size_t i = e->lo; i < e->hi; ++i) o[i] = fmdh_nt6[a[i]];
    return 0;
}
static void encode_batch(batch_t *b)
{
    int T = g_host_threads < 1 ? 1 : (g_host_threads > 64 ? 64 : g_host_threads), t;
    pthread_t tid[64];
    enc_t e[64];
    int started[64];
    if (b->bytes < ((size_t)1 << 20)) T = 1;
    for (t = 0; t < T; ++t) { e[t].b = b; e[t].lo = b->bytes * (size_t)t / (size_t)T; e[t].hi = b->bytes * (size_t)(t + 1) / (size_t)T; }
    for (t = 1; t < T; ++t) started[t] = pthread_create(&tid[t], 0, enc_main, &e[t]) == 0;
    enc_main(&e[0]);
    for (t = 1; t < T; ++t) { if (started[t]) pthread_join(tid[t], 0); else enc_main(&e[t]); }
}

/* one GPU's slice [lo, hi) of a batch */
typedef struct { pipe_t *p; batch_t *b; int g; size_t lo, hi; int rc; } gslice_t;
static void *gslice_main(void *d)
{
    gslice_t *s = (gslice_t *)d;
    s->rc = s->hi > s->lo ? fmd_ecfix_batch(s->p->tab[s->g], s->hi - s->lo, s->b->nt6, s->b->qual, s->b->off + s->lo, s->p->opt->step, s->b->info + s->lo) : 0;
    return 0;
}
/* ec_fix (correct.c:221-246) of a batch: reads are independent given the table, so with several GPUs each takes a contiguous share */
static int fix_batch(pipe_t *p, batch_t *b)
{
    gslice_t sl[FMDH_MAX_GPUS];
    pthread_t tid[FMDH_MAX_GPUS];
    int started[FMDH_MAX_GPUS], g, rc = 0;
    const int G = p->n_tab;
    if (G == 1) return fmd_ecfix_batch(p->tab[0], b->nb, b->nt6, b->qual, b->off, p->opt->step, b->info);
    for (g = 0; g < G; ++g) { sl[g].p = p; sl[g].b = b; sl[g].g = g; sl[g].lo = b->nb * (size_t)g / (size_t)G; sl[g].hi = b->nb * (size_t)(g + 1) / (size_t)G; sl[g].rc = 0; }
    for (g = 1; g < G; ++g) started[g] = pthread_create(&tid[g], 0, gslice_main, &sl[g]) == 0;
    gslice_main(&sl[0]);
    for (g = 1; g < G; ++g) { if (started[g]) pthread_join(tid[g], 0); else gslice_main(&sl[g]); }
    for (g = 0; g < G; ++g) if (sl[g].rc && !rc) rc = sl[g].rc;
    return rc;
}

/* stage 2: the GPU corrects a parsed batch in place (nt6 + qual) */
static void *stage_gpu(void *d)
{
    pipe_t *p = (pipe_t *)d;
    for (unsigned k = 0;; ++k) {
        batch_t *b = &p->b[k % 3];
        slot_wait(p, b, 1);
        const double t0 = now_s();
        if (b->nb && !__atomic_load_n(&p->failed, __ATOMIC_RELAXED)) encode_batch(b);
        if (b->nb && !__atomic_load_n(&p->failed, __ATOMIC_RELAXED)) {
            const int rc = fix_batch(p, b);
            if (rc) { fprintf(stderr, "[E::%s] correction pass failed: %s\n", __func__, fmd_strerror(rc)); __atomic_store_n(&p->failed, 1, __ATOMIC_RELAXED); }
        }
        p->t_gpu += now_s() - t0;
        const int last = b->last;
        slot_set(p, b, 2);
        if (last) return 0;
    }
}

/* What the reference does to a corrected read before printing it, on a slice [lo, hi) of a batch and in two passes
 * because the pair rule looks at the neighbour's verdict:
 *   pass 0 (correct.c:247-252): corrected bases in lower case with quality 36 ('$'); bit 16 of info = "too many
 *          corrections / too close a second best";
 *   pass 1 (correct.c:396-425): drop bad reads (a pair is bad when either end is), format the FASTQ records of the
 *          slice into its own buffer -- the writer thread then has one fwrite per slice. */
typedef struct { const fmdh_ecopt_t *opt; batch_t *b; size_t lo, hi; int pass; char *text; size_t text_l, text_m; int failed; int fd; off_t at; } __attribute__((aligned(128))) slice_t;   /* fd, at: pass 2 (a regular file: every slice written at its own offset) */
static inline size_t put_dec(char *p, long long v)   /* decimal digits of v, as %lld prints them; -> their number */
{
    char t[24];
    size_t n = 0, i;
    unsigned long long u = v < 0 ? 0ull - (unsigned long long)v : (unsigned long long)v;
    do { t[n++] = (char)('0' + u % 10); u /= 10; } while (u);
    if (v < 0) t[n++] = '-';
    for (i = 0; i < n; ++i) p[i] = t[n - 1 - i];
    return n;
}
static void *slice_main(void *d)
{
    slice_t *w = (slice_t *)d;
    batch_t *b = w->b;
    const fmdh_ecopt_t *opt = w->opt;
    if (w->pass == 2) {
        size_t done = 0;
        while (done < w->text_l) { const ssize_t k = pwrite(w->fd, w->text + done, w->text_l - done, w->at + (off_t)done); if (k <= 0) { w->failed = 2; break; } done += (size_t)k; }
        return 0;
    }
    if (w->pass == 0) {
        for (size_t i = w->lo; i < w->hi; ++i) {
            char *a = b->ascii + b->off[i];
            const uint8_t *s = b->nt6 + b->off[i];
            uint8_t *q = b->qual + b->off[i];
            const int l = (int)(b->off[i + 1] - b->off[i]);
            int n_lower = 0, info = b->info[i];
            for (int j = 0; j < l; ++j) {   /* (toupper / islower of the "C" locale, in line: two library calls per base were most of this stage at 5*10^9 bases) */
                const unsigned char c = (unsigned char)a[j];
                const unsigned char up = (unsigned char)(c - 'a') < 26u ? (unsigned char)(c - 32) : c;
                const unsigned char o = fmdh_nt6[c] == s[j] ? up : (unsigned char)"$acgtn"[s[j]];
                const int low = (unsigned char)(o - 'a') < 26u;
                a[j] = (char)o;
                n_lower += low;
                if (low) q[j] = 36;
            }
            if ((double)n_lower / l > opt->max_corr) info |= 1 << 16;
            if (info >> 18 <= 10) info |= 1 << 16;
            b->info[i] = info;
        }
        return 0;
    }
    {
        const size_t need = 2 * (size_t)(b->off[w->hi] - b->off[w->lo]) + (w->hi - w->lo) * 48 + 64;
        if (need > w->text_m) { char *t = (char *)realloc(w->text, need); if (!t) { w->failed = 1; return 0; } w->text = t; w->text_m = need; }
    }
    size_t text_l = 0;   /* (a local: the slices sit side by side, and a store per read into a neighbour's cache line is felt by sixteen threads) */
    for (size_t a = w->lo; a < w->hi; ++a) {
        const uint64_t k = b->first_id + a;
        const int32_t *info = b->info;
        int is_bad = 0;
        if (opt->is_paired) { /* batches hold whole pairs (BATCH_SIZE is even) */
            if (info[a] >> 16 & 1) is_bad = 1;
            else if (k & 1) { if (a >= 1 && (info[a - 1] >> 16 & 1)) is_bad = 1; }
            else if (a + 1 < b->nb && (info[a + 1] >> 16 & 1)) is_bad = 1;
        } else if (info[a] >> 16 & 1) is_bad = 1;
        if (is_bad && !opt->keep_bad) continue;
        int len = (int)(b->off[a + 1] - b->off[a]);
        if (opt->trim_l && opt->trim_l < len) len = opt->trim_l;
        char *o = w->text + text_l;
        {   /* "@%lld%c%d%c%d\n" (correct.c:411-416) without printf: 5*10^7 calls of it were a tenth of this stage */
            const char sep = opt->is_paired ? ' ' : '_';
            *o++ = '@'; o += put_dec(o, (long long)(opt->is_paired ? k >> 1 : k));
            *o++ = sep; o += put_dec(o, (long long)(info[a] & 0xffff));
            *o++ = sep; o += put_dec(o, (long long)(info[a] >> 18));
            *o++ = '\n';
        }
        memcpy(o, b->ascii + b->off[a], (size_t)len); o += len;
        memcpy(o, "\n+\n", 3); o += 3;
        memcpy(o, b->qual + b->off[a], (size_t)len); o += len;
        *o++ = '\n';
        text_l = (size_t)(o - w->text);
    }
    w->text_l = text_l;
    return 0;
}

/* stage 3: mark, filter, format on g_host_threads threads; print in order */
#define MAX_SLICES 64
static void *stage_print(void *d)
{
    pipe_t *p = (pipe_t *)d;
    slice_t sl[MAX_SLICES];
    pthread_t tid[MAX_SLICES];
    char started[MAX_SLICES];
    memset(sl, 0, sizeof(sl));
    /* a regular file that is not in append mode: the slices of a batch are written concurrently, each at its offset (one fwrite of 11 GB was what
     * `correct` of 5*10^7 reads waited for: mark + print busy 2.76 of 2.85 s) */
    int out_fd = -1;
    off_t out_at = 0;
    {
        struct stat sb;
        const int fd = fileno(p->out);
        fflush(p->out);
        if (fd >= 0 && !getenv("FMD_CORRECT_ONE_WRITER") && fstat(fd, &sb) == 0 && S_ISREG(sb.st_mode) && !(fcntl(fd, F_GETFL) & O_APPEND)) { out_at = lseek(fd, 0, SEEK_CUR); if (out_at >= 0) out_fd = fd; }
    }
    for (unsigned kb = 0;; ++kb) {
        batch_t *b = &p->b[kb % 3];
        slot_wait(p, b, 2);
        const double t0 = now_s();
        if (!__atomic_load_n(&p->failed, __ATOMIC_RELAXED) && b->nb) {
            int T = g_host_threads < MAX_SLICES ? g_host_threads : MAX_SLICES, t, pass;
            if ((size_t)T > b->nb / 4096 + 1) T = (int)(b->nb / 4096 + 1);
            for (pass = 0; pass < 2; ++pass) {
                for (t = 0; t < T; ++t) {
                    sl[t].opt = p->opt; sl[t].b = b; sl[t].pass = pass;
                    sl[t].lo = b->nb * (size_t)t / (size_t)T; sl[t].hi = b->nb * (size_t)(t + 1) / (size_t)T;
                    started[t] = t > 0 && pthread_create(&tid[t], 0, slice_main, &sl[t]) == 0;
                }
                for (t = 0; t < T; ++t) if (!started[t]) slice_main(&sl[t]);
                for (t = 1; t < T; ++t) if (started[t]) pthread_join(tid[t], 0);
            }
            for (t = 0; t < T; ++t) if (sl[t].failed) { fprintf(stderr, "[E::%s] out of memory\n", __func__); __atomic_store_n(&p->failed, 1, __ATOMIC_RELAXED); break; }
            if (!__atomic_load_n(&p->failed, __ATOMIC_RELAXED) && out_fd >= 0) {
                for (t = 0; t < T; ++t) { sl[t].pass = 2; sl[t].fd = out_fd; sl[t].at = out_at; out_at += (off_t)sl[t].text_l; started[t] = t > 0 && pthread_create(&tid[t], 0, slice_main, &sl[t]) == 0; }
                for (t = 0; t < T; ++t) if (!started[t]) slice_main(&sl[t]);
                for (t = 1; t < T; ++t) if (started[t]) pthread_join(tid[t], 0);
                for (t = 0; t < T; ++t) if (sl[t].failed) { fprintf(stderr, "[E::%s] write error\n", __func__); __atomic_store_n(&p->failed, 1, __ATOMIC_RELAXED); break; }
            } else if (!__atomic_load_n(&p->failed, __ATOMIC_RELAXED))
                for (t = 0; t < T; ++t)
                    if (fwrite(sl[t].text, 1, sl[t].text_l, p->out) != sl[t].text_l) { fprintf(stderr, "[E::%s] write error\n", __func__); __atomic_store_n(&p->failed, 1, __ATOMIC_RELAXED); break; }
        }
        const int last = b->last;
        p->t_write += now_s() - t0;
        slot_set(p, b, 0);
        if (last) break;
    }
    for (int t = 0; t < MAX_SLICES; ++t) free(sl[t].text);
    if (out_fd >= 0 && lseek(out_fd, out_at, SEEK_SET) < 0) __atomic_store_n(&p->failed, 1, __ATOMIC_RELAXED);   /* whoever writes next goes on behind the records */
    return 0;
}

/* Phase 2 alone: correct the reads of fq_path against a harvested table (opt->w must be set). */
int fmdh_correct_reads(const fmdh_ecopt_t *opt, int device, int suf_len, uint64_t n, const uint32_t *bucket, const uint32_t *key, const uint8_t *val,
                       const char *fq_path, FILE *out)
{
    return fmdh_correct_reads_multi(opt, 1, &device, suf_len, n, bucket, key, val, fq_path, out);
}

static void free_tabs(pipe_t *p) { for (int g = 0; g < p->n_tab; ++g) fmd_ectab_free(p->tab[g]); p->n_tab = 0; }

int fmdh_correct_reads_multi(const fmdh_ecopt_t *opt, int n_dev, const int *devices, int suf_len, uint64_t n, const uint32_t *bucket, const uint32_t *key,
                             const uint8_t *val, const char *fq_path, FILE *out)
{
    const int timing = getenv("FMD_TIMING") != 0;
    const double t_begin = now_s();
    double t_read = 0;
    pipe_t p;
    memset(&p, 0, sizeof(p));
    if (n_dev < 1 || n_dev > FMDH_MAX_GPUS) return 1;
    for (int g = 0; g < n_dev; ++g) {   /* the whole table on every GPU (one 8-byte slot per solid k-mer) */
        const int rc = fmd_ectab_build(devices[g], opt->w, suf_len, n, bucket, key, val, &p.tab[g]);
        if (rc) { fprintf(stderr, "[E::%s] cannot load the k-mer table on GPU %d: %s\n", __func__, devices[g], fmd_strerror(rc)); free_tabs(&p); return 1; }
        p.n_tab = g + 1;
    }
    const double t_table = now_s() - t_begin;
    fmdh_seqio_t *io = fmdh_seq_open(fq_path);
    if (!io) { fprintf(stderr, "[E::%s] cannot open `%s'\n", __func__, fq_path); free_tabs(&p); return 1; }
    p.opt = opt; p.out = out;
    pthread_mutex_init(&p.mu, 0); pthread_cond_init(&p.cv, 0);
    for (int i = 0; i < 3; ++i) if (batch_reads_room(&p.b[i], BATCH_SIZE)) __atomic_store_n(&p.failed, 1, __ATOMIC_RELAXED);
    /* a plain file is parsed by several threads (seqpar.c), a span of it per batch; gzip and stdin by the one reader below */
    int pt = g_host_threads > 1 ? g_host_threads : 8;
    { const char *e = getenv("FMD_HOST_THREADS"); if (e && atoi(e) > 0) pt = atoi(e); }
    fmdh_pseq_t *pr = fmdh_pseq_open(fq_path, pt, (size_t)220 << 20);   /* (~10^6 reads of 100 bases: the reference's batch, correct.c:281) */
    char *carry_s = 0, *carry_q = 0; size_t carry_l = 0; int have_carry = 0;   /* pairs: a batch holds whole pairs -- an odd read waits for the next batch */
    pthread_t t_gpu, t_out;
    int have_gpu_thread = 0, have_out_thread = 0;
    if (!__atomic_load_n(&p.failed, __ATOMIC_RELAXED)) {
        have_gpu_thread = pthread_create(&t_gpu, 0, stage_gpu, &p) == 0;
        have_out_thread = have_gpu_thread && pthread_create(&t_out, 0, stage_print, &p) == 0;
        if (!have_gpu_thread || !have_out_thread) __atomic_store_n(&p.failed, 1, __ATOMIC_RELAXED);
    }
    uint64_t id = 0;
    for (unsigned kb = 0; have_out_thread; ++kb) { /* batches of BATCH_SIZE reads, output in input order */
        batch_t *b = &p.b[kb % 3];
        slot_wait(&p, b, 0);
        const double t0 = now_s();
        b->nb = 0; b->bytes = 0; b->first_id = id; b->last = 0;
        b->off[0] = 0;
        if (pr) {
            fmdh_ppart_t *parts = 0;
            int np = 0, k;
            const int got = fmdh_pseq_next(pr, &parts, &np);
            if (got <= 0) { b->last = 1; if (got < 0) { fprintf(stderr, "[E::%s] out of memory\n", __func__); __atomic_store_n(&p.failed, 1, __ATOMIC_RELAXED); } np = 0; }
            size_t tot_n = have_carry ? 1 : 0, tot_b = have_carry ? carry_l : 0;
            for (k = 0; k < np; ++k) { tot_n += parts[k].n; tot_b += parts[k].bytes; }
            if (batch_room(b, tot_b) || batch_reads_room(b, tot_n + 1)) { fprintf(stderr, "[E::%s] out of memory\n", __func__); __atomic_store_n(&p.failed, 1, __ATOMIC_RELAXED); b->last = 1; np = 0; tot_n = tot_b = 0; have_carry = 0; }
            if (have_carry && tot_n) { memcpy(b->ascii, carry_s, carry_l); memcpy(b->qual, carry_q, carry_l); b->off[1] = carry_l; }
            {
                fill_t fl[64]; pthread_t ft[64]; int fs[64];
                size_t r0 = have_carry ? 1 : 0, o0 = have_carry ? carry_l : 0;
                for (k = 0; k < np; ++k) { fl[k].b = b; fl[k].p = &parts[k]; fl[k].r0 = r0; fl[k].o0 = o0; r0 += parts[k].n; o0 += parts[k].bytes; }
                for (k = 1; k < np; ++k) fs[k] = pthread_create(&ft[k], 0, fill_main, &fl[k]) == 0;
                if (np) fill_main(&fl[0]);
                for (k = 1; k < np; ++k) { if (fs[k]) pthread_join(ft[k], 0); else fill_main(&fl[k]); }
            }
            have_carry = 0;
            b->nb = tot_n; b->bytes = tot_b;
            if (opt->is_paired && (b->nb & 1) && !b->last) {   /* the odd read at the end opens the next batch */
                const size_t l = (size_t)(b->off[b->nb] - b->off[b->nb - 1]);
                char *cs = (char *)realloc(carry_s, l + 1), *cq = (char *)realloc(carry_q, l + 1);
                if (cs) carry_s = cs;
                if (cq) carry_q = cq;
                if (!cs || !cq) { __atomic_store_n(&p.failed, 1, __ATOMIC_RELAXED); b->last = 1; }
                else { memcpy(carry_s, b->ascii + b->off[b->nb - 1], l); memcpy(carry_q, b->qual + b->off[b->nb - 1], l); carry_l = l; have_carry = 1; --b->nb; b->bytes -= l; }
            }
            id += b->nb;
        }
        while (!pr && b->nb < BATCH_SIZE) {
            const int len = fmdh_seq_read(io);
            if (len < 0) { b->last = 1; break; }
            if (batch_room(b, (size_t)len)) { fprintf(stderr, "[E::%s] out of memory\n", __func__); __atomic_store_n(&p.failed, 1, __ATOMIC_RELAXED); b->last = 1; break; }
            const char *s = fmdh_seq_bases(io), *q = fmdh_seq_qual(io);
            memcpy(b->ascii + b->bytes, s, (size_t)len);                  /* (nt6 codes: encode_batch, second stage) */
            if (q) memcpy(b->qual + b->bytes, q, (size_t)len);
            else memset(b->qual + b->bytes, 33 + 15, (size_t)len);      /* no quality: phred 15 (correct.c:431-436) */
            b->bytes += (size_t)len;
            b->off[++b->nb] = b->bytes;
            ++id;
        }
        const int last = b->last;
        t_read += now_s() - t0;
        slot_set(&p, b, 1);
        if (last) break;
    }
    if (have_gpu_thread && !have_out_thread) { p.b[0].nb = 0; p.b[0].last = 1; slot_set(&p, &p.b[0], 1); } /* let the lone stage end */
    if (have_gpu_thread) pthread_join(t_gpu, 0);
    if (have_out_thread) pthread_join(t_out, 0);
    if (timing) fprintf(stderr, "[M::%s] table upload %.3f s; pipeline %.3f s with stages busy for: parse %.3f s, correction (GPU + copies) %.3f s, mark + print %.3f s\n",
                        __func__, t_table, now_s() - t_begin - t_table, t_read, p.t_gpu, p.t_write);
    for (int i = 0; i < 3; ++i) { free(p.b[i].ascii); free(p.b[i].nt6); free(p.b[i].qual); free(p.b[i].off); free(p.b[i].info); }
    pthread_mutex_destroy(&p.mu); pthread_cond_destroy(&p.cv);
    fmdh_seq_close(io);
    fmdh_pseq_close(pr);
    free(carry_s); free(carry_q);
    free_tabs(&p);
    return __atomic_load_n(&p.failed, __ATOMIC_RELAXED) ? 1 : 0;
}

int fmdh_correct_kmer(uint64_t n_symbols) /* the automatic k-mer length, correct.c:313-318 */
{
    int w = (int)(log((double)n_symbols) / log(4) + 8.499);
    return w >= MAX_KMER ? MAX_KMER : w;
}

/* `correct -g a,b,..`: GPU g of G harvests the trees of the k-mer trie whose last base is c = g, g + G, .. (fm6_traverse + ec_collect are
 * sharded over the reference's threads by suffix bucket, correct.c:346-356; the trie is a forest rooted at the last base, so whole
 * trees are the natural shards: at most four GPUs harvest), every GPU then holds the whole table and corrects a contiguous share of
 * each batch of reads.  Output bytes are those of one GPU. */
typedef struct { const char *fmd_path; int device, w, min_occ, suf_len, seeds, rc; uint32_t *bucket, *key; uint8_t *val; uint64_t n; int64_t cnt[2]; uint64_t n_sym; } hv_t;
static void *hv_main(void *d)
{
    hv_t *h = (hv_t *)d;
    fmd_dev_t *dev = 0;
    h->rc = fmd_dev_open_file(h->device, h->fmd_path, &dev);
    if (h->rc) return 0;
    h->rc = fmd_kmer_collect_seeds(dev, h->w, h->min_occ, h->suf_len, h->seeds, &h->bucket, &h->key, &h->val, &h->n, h->cnt);
    fmd_dev_close(dev);
    return 0;
}

int fmdh_correct_multi(const char *fmd_path, const char *fq_path, int n_dev, const int *devices, fmdh_ecopt_t *opt, FILE *out)
{
    if (n_dev < 1 || n_dev > FMDH_MAX_GPUS) return 1;
    if (n_dev == 1) return fmdh_correct(fmd_path, fq_path, devices[0], opt, out);
    const int timing = getenv("FMD_TIMING") != 0;
    double t0 = now_s(), t1;
    int rc, g;
    if (opt->w < 0) {   /* the automatic k needs the symbol count (correct.c:313-318): read it from the header through one load */
        fmd_dev_t *d = 0; fmd_info_t fi;
        rc = fmd_dev_open_file(devices[0], fmd_path, &d);
        if (rc) { fprintf(stderr, "[E::%s] cannot load `%s': %s\n", __func__, fmd_path, fmd_strerror(rc)); return 1; }
        fmd_dev_info(d, &fi); fmd_dev_close(d);
        opt->w = fmdh_correct_kmer(fi.mcnt[0]);
    }
    const int suf_len = opt->w > 15 ? opt->w - 15 : 1, H = n_dev < 4 ? n_dev : 4;
    hv_t hv[4];
    pthread_t tid[4];
    int started[4];
    memset(hv, 0, sizeof(hv));
    for (g = 0; g < H; ++g) {
        hv[g].fmd_path = fmd_path; hv[g].device = devices[g]; hv[g].w = opt->w; hv[g].min_occ = opt->min_occ; hv[g].suf_len = suf_len;
        for (int c = g; c < 4; c += H) hv[g].seeds |= 1 << c;
    }
    for (g = 1; g < H; ++g) started[g] = pthread_create(&tid[g], 0, hv_main, &hv[g]) == 0;
    hv_main(&hv[0]);
    for (g = 1; g < H; ++g) { if (started[g]) pthread_join(tid[g], 0); else hv_main(&hv[g]); }
    uint64_t n = 0, o = 0;
    rc = 0;
    for (g = 0; g < H; ++g) { if (hv[g].rc && !rc) rc = hv[g].rc; n += hv[g].n; }
    uint32_t *bucket = 0, *key = 0; uint8_t *val = 0;
    if (!rc) {
        bucket = (uint32_t *)malloc((n ? n : 1) * 4); key = (uint32_t *)malloc((n ? n : 1) * 4); val = (uint8_t *)malloc(n ? n : 1);
        if (!bucket || !key || !val) rc = FMD_E_NOMEM;
        else for (g = 0; g < H; ++g) {
            memcpy(bucket + o, hv[g].bucket, hv[g].n * 4); memcpy(key + o, hv[g].key, hv[g].n * 4); memcpy(val + o, hv[g].val, hv[g].n);
            o += hv[g].n;
        }
    }
    for (g = 0; g < H; ++g) { fmd_host_free(hv[g].bucket); fmd_host_free(hv[g].key); fmd_host_free(hv[g].val); }
    if (rc) { fprintf(stderr, "[E::%s] k-mer harvest failed: %s\n", __func__, fmd_strerror(rc)); free(bucket); free(key); free(val); return 1; }
    if (timing) { t1 = now_s(); fprintf(stderr, "[M::%s] index load + harvest of %llu solid %d-mers on %d GPUs (trees by last base): %.3f s\n", __func__, (unsigned long long)n, opt->w + 1, H, t1 - t0); t0 = t1; }
    rc = fmdh_correct_reads_multi(opt, n_dev, devices, suf_len, n, bucket, key, val, fq_path, out);
    if (timing) { t1 = now_s(); fprintf(stderr, "[M::%s] table upload + correction on %d GPUs + output: %.3f s\n", __func__, n_dev, t1 - t0); }
    free(bucket); free(key); free(val);
    return rc;
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
    rc = fmdh_correct_reads(opt, device, suf_len, n, bucket, key, val, fq_path, out);
    if (timing) { t1 = now_s(); fprintf(stderr, "[M::%s] table upload + correction + output: %.3f s\n", __func__, t1 - t0); t0 = t1; }
    fmd_host_free(bucket); fmd_host_free(key); fmd_host_free(val);
    return rc;
}
