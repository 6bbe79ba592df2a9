/* remap_cmd.c -- `fermi remap [-l skip] [-c min_pcv] [-D max_dist] [-r rank] <reads.fmd> <contigs.fq>`
 * (cmd.c:218-251 -> fm6_remap, smem.c:353-394): for every contig, the reads that match it over their
 * full length (SMEMs closed by a sentinel on both sides) give the per-base coverage string, the
 * support count and -- with a rank file -- the paired-end coverage used to break contigs.
 *
 * The reference walks each contig with fm6_miter_next (smem.c:96-102) on one CPU thread per contig.
 * Here the GPU first computes the forward reach of every contig position (fmd_reach_batch), which
 * turns the iterator's chain of start positions into a pointer chase, and then runs every
 * fm6_smem1_core call of that chain as an independent work item (fmd_smem_win_batch, full-length
 * matches only); what follows -- paircov, mask_pcv, the two output formats,
 * the insert-size line on stderr -- is restated for the host.  The pair table is a restatement of
 * the open-addressing table the reference uses (khash 0.2.6 semantics: double hashing, two flag bits
 * per bucket, growth at 0.77, in-place rehash), because the order in which left-over mates are
 * listed in the UR:Z tag is that table's bucket order, and the table lives across contigs.
 * Output = `fermi remap -t1` (the reference's -tN interleaves contigs nondeterministically). */
#include <ctype.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "fmd_host.h"

#define MASK30 0x3fffffffull

/* ---- u64 -> u64 table ------------------------------------------------------------------------ */
typedef struct {
    uint32_t n_buckets, size, n_occupied, upper_bound;
    uint32_t *flags;          /* 2 bits per bucket: bit 1 = empty, bit 0 = deleted */
    uint64_t *keys, *vals;
} pmap_t;

#define PM_EMPTY(f, i) ((f[(i) >> 4] >> (((i) & 0xfU) << 1)) & 2)
#define PM_DEL(f, i) ((f[(i) >> 4] >> (((i) & 0xfU) << 1)) & 1)
#define PM_EITHER(f, i) ((f[(i) >> 4] >> (((i) & 0xfU) << 1)) & 3)
#define PM_SET_DEL(f, i) (f[(i) >> 4] |= 1u << (((i) & 0xfU) << 1))
#define PM_CLR_EMPTY(f, i) (f[(i) >> 4] &= ~(2u << (((i) & 0xfU) << 1)))
#define PM_CLR_BOTH(f, i) (f[(i) >> 4] &= ~(3u << (((i) & 0xfU) << 1)))
#define PM_FWORDS(m) ((m) < 16 ? 1 : (m) >> 4)

static inline uint32_t pm_hash(uint64_t key) { return (uint32_t)(key >> 33 ^ key ^ key << 11); }
static inline uint32_t pm_step(uint32_t k, uint32_t mask) { return ((k >> 3 ^ k << 3) | 1) & mask; }

static void pm_free(pmap_t *h) { free(h->flags); free(h->keys); free(h->vals); memset(h, 0, sizeof(*h)); }
static void pm_clear(pmap_t *h)
{
    if (h->flags) { memset(h->flags, 0xaa, PM_FWORDS(h->n_buckets) * 4); h->size = h->n_occupied = 0; }
}
static uint32_t pm_get(const pmap_t *h, uint64_t key) /* bucket, or n_buckets when absent */
{
    if (h->n_buckets == 0) return 0;
    const uint32_t mask = h->n_buckets - 1, k = pm_hash(key), inc = pm_step(k, mask);
    uint32_t i = k & mask;
    const uint32_t last = i;
    while (!PM_EMPTY(h->flags, i) && (PM_DEL(h->flags, i) || h->keys[i] != key)) {
        i = (i + inc) & mask;
        if (i == last) return h->n_buckets;
    }
    return PM_EITHER(h->flags, i) ? h->n_buckets : i;
}
static void pm_resize(pmap_t *h, uint32_t want)
{
    uint32_t nb = want, j;
    --nb; nb |= nb >> 1; nb |= nb >> 2; nb |= nb >> 4; nb |= nb >> 8; nb |= nb >> 16; ++nb; /* next power of two */
    if (nb < 4) nb = 4;
    if (h->size >= (uint32_t)(nb * 0.77 + 0.5)) return; /* requested size is too small */
    uint32_t *nf = (uint32_t *)malloc(PM_FWORDS(nb) * 4);
    memset(nf, 0xaa, PM_FWORDS(nb) * 4);
    if (h->n_buckets < nb) {
        h->keys = (uint64_t *)realloc(h->keys, (size_t)nb * 8);
        h->vals = (uint64_t *)realloc(h->vals, (size_t)nb * 8);
    }
    const uint32_t nmask = nb - 1;
    for (j = 0; j != h->n_buckets; ++j) {
        if (PM_EITHER(h->flags, j)) continue;
        uint64_t key = h->keys[j], val = h->vals[j];
        PM_SET_DEL(h->flags, j);
        for (;;) { /* place it; an element already sitting at the target in the OLD layout is carried on */
            const uint32_t k = pm_hash(key), inc = pm_step(k, nmask);
            uint32_t i = k & nmask;
            while (!PM_EMPTY(nf, i)) i = (i + inc) & nmask;
            PM_CLR_EMPTY(nf, i);
            if (i < h->n_buckets && PM_EITHER(h->flags, i) == 0) {
                uint64_t t = h->keys[i]; h->keys[i] = key; key = t;
                t = h->vals[i]; h->vals[i] = val; val = t;
                PM_SET_DEL(h->flags, i);
            } else { h->keys[i] = key; h->vals[i] = val; break; }
        }
    }
    if (h->n_buckets > nb) {
        h->keys = (uint64_t *)realloc(h->keys, (size_t)nb * 8);
        h->vals = (uint64_t *)realloc(h->vals, (size_t)nb * 8);
    }
    free(h->flags);
    h->flags = nf; h->n_buckets = nb; h->n_occupied = h->size;
    h->upper_bound = (uint32_t)(nb * 0.77 + 0.5);
}
static uint32_t pm_put(pmap_t *h, uint64_t key)
{
    if (h->n_occupied >= h->upper_bound) {
        if (h->n_buckets > (h->size << 1)) pm_resize(h, h->n_buckets - 1); /* only tombstones to clear */
        else pm_resize(h, h->n_buckets + 1);
    }
    const uint32_t mask = h->n_buckets - 1, k = pm_hash(key);
    uint32_t x = h->n_buckets, site = h->n_buckets, i = k & mask;
    if (PM_EMPTY(h->flags, i)) x = i;
    else {
        const uint32_t inc = pm_step(k, mask), last = i;
        while (!PM_EMPTY(h->flags, i) && (PM_DEL(h->flags, i) || h->keys[i] != key)) {
            if (PM_DEL(h->flags, i)) site = i;
            i = (i + inc) & mask;
            if (i == last) { x = site; break; }
        }
        if (x == h->n_buckets) x = (PM_EMPTY(h->flags, i) && site != h->n_buckets) ? site : i;
    }
    if (PM_EMPTY(h->flags, x)) { h->keys[x] = key; PM_CLR_BOTH(h->flags, x); ++h->size; ++h->n_occupied; }
    else if (PM_DEL(h->flags, x)) { h->keys[x] = key; PM_CLR_BOTH(h->flags, x); ++h->size; }
    return x;
}
static void pm_del(pmap_t *h, uint32_t x)
{
    if (x != h->n_buckets && !PM_EITHER(h->flags, x)) { PM_SET_DEL(h->flags, x); --h->size; }
}

/* ---- per-contig work (smem.c:139-303) -------------------------------------------------------- */
typedef struct { uint64_t x, y; } u128_t;
struct fmdh_remap_state {
    fmdh_remapopt_t opt;
    const uint64_t *sorted;
    uint64_t n_seq;
    pmap_t h;
    uint64_t rec[3];
    uint8_t *cov; size_t cov_m;          /* cov + pcv */
    u128_t *unp; size_t unp_n, unp_m;
    char *line; size_t line_l, line_m;
};

fmdh_remap_state_t *fmdh_remap_new(const fmdh_remapopt_t *opt, const uint64_t *sorted, uint64_t n_seq)
{
    fmdh_remap_state_t *st = (fmdh_remap_state_t *)calloc(1, sizeof(*st));
    st->opt = *opt; st->sorted = sorted; st->n_seq = n_seq;
    if (sorted == 0) { st->opt.skip = -1; st->opt.min_pcv = 0; } /* no rank -> index map: nothing is broken (smem.c:233) */
    return st;
}

static void o_putc(fmdh_remap_state_t *st, int c)
{
    if (st->line_l + 2 > st->line_m) { st->line_m = st->line_m ? st->line_m << 1 : 1024; st->line = (char *)realloc(st->line, st->line_m); }
    st->line[st->line_l++] = (char)c;
}
static void o_putsn(fmdh_remap_state_t *st, const char *s, size_t n) { for (size_t i = 0; i < n; ++i) o_putc(st, s[i]); }
static void o_puts(fmdh_remap_state_t *st, const char *s) { o_putsn(st, s, strlen(s)); }
static void o_putl(fmdh_remap_state_t *st, long long v) { char b[32]; snprintf(b, sizeof(b), "%lld", v); o_puts(st, b); }
static void unp_push(fmdh_remap_state_t *st, uint64_t x, uint64_t y)
{
    if (st->unp_n == st->unp_m) { st->unp_m = st->unp_m ? st->unp_m << 1 : 16; st->unp = (u128_t *)realloc(st->unp, st->unp_m * sizeof(u128_t)); }
    st->unp[st->unp_n].x = x; st->unp[st->unp_n].y = y; ++st->unp_n;
}

/* Case = verdict of the paired coverage (smem.c:201-224): between the first and the last base
 * with pcv >= min_pcv a base is upper case iff its own pcv is high enough; the two flanks are kept
 * upper case only when they are shorter than 2*skip (the ends no pair can cover); a contig without
 * any supported base stays upper case. */
static void mask_pcv(int l, char *seq, const uint8_t *pcv, int skip, int min_pcv)
{
    int first = 0, last = l, i;
    while (first < l && pcv[first] < min_pcv) ++first;
    if (first < l) while (pcv[last - 1] < min_pcv) --last;
    const int none = first == l, head_up = first < skip << 1, tail_up = l - last < skip << 1;
    for (i = 0; i < l; ++i) {
        const int up = none ? 1 : i < first ? head_up : i >= last ? tail_up : pcv[i] >= min_pcv;
        seq[i] = (up ? "$ACGTN" : "$acgtn")[(int)seq[i]];
    }
}

void fmdh_remap_contig(fmdh_remap_state_t *st, const char *name, const char *comment, int len, uint8_t *si, const fmd_intv_t *mem, size_t n_mem, FILE *out)
{
    const uint64_t mask = MASK30 << 32 | MASK30;
    const int skip = st->opt.skip, min_pcv = st->opt.min_pcv, max_dist = st->opt.max_dist;
    int n_supp = 0, j;
    size_t m;
    if ((size_t)(len + 1) * 2 > st->cov_m) { st->cov_m = (size_t)(len + 1) * 2; st->cov = (uint8_t *)realloc(st->cov, st->cov_m); }
    memset(st->cov, 0, (size_t)(len + 1) * 2);
    uint8_t *cov = st->cov, *pcv = st->cov + len + 1;
    st->unp_n = 0;
    if (st->h.n_buckets >= 256) pm_free(&st->h); /* smem.c:241-244 */
    /* paircov (smem.c:139-199) over the full-length matches in the order the iterator yields them */
    for (m = 0; m < n_mem; ++m) {
        const fmd_intv_t *p = &mem[m];
        if (!(p->info >> 63 && p->x[1] < st->n_seq)) continue;
        const int end0 = (int)(p->info & MASK30);
        for (j = (int)(p->info >> 32 & MASK30); j < end0; ++j) if (cov[j] < 255) ++cov[j];
        ++n_supp;
        if (skip <= 0 || st->sorted == 0) continue;
        for (uint64_t l = 0; l < p->x[2]; ++l) {
            const uint64_t k = st->sorted[p->x[1] + l] >> 2; /* x[1]: the interval of the reverse strand */
            if ((k & 1) == 0) { /* reverse strand: look for the mate */
                int beg = 0, end = 0, to_add = 0;
                const uint32_t kk = pm_get(&st->h, k);
                if (st->h.n_buckets && kk != st->h.n_buckets) {
                    beg = (int)(st->h.vals[kk] >> 32);
                    end = (int)(p->info & MASK30);
                    if (end - beg < max_dist) { ++st->rec[0]; st->rec[1] += (uint64_t)(end - beg); st->rec[2] += (uint64_t)((end - beg) * (end - beg)); }
                    else to_add = 1;
                } else to_add = 1;
                if (to_add) { unp_push(st, k ^ 1, p->info & mask); continue; }
                beg += skip; end -= skip;
                if (beg > end) { const int t = beg; beg = end; end = t; }
                if (beg < 0) beg = 0;
                if (end > len) end = len;
                for (j = beg; j < end; ++j) if (pcv[j] < 255) ++pcv[j];
                pm_del(&st->h, kk);
            } else { /* forward strand: remember it */
                const uint32_t kk = pm_put(&st->h, k ^ 3);
                st->h.vals[kk] = p->info & mask;
            }
        }
    }
    for (uint32_t kk = 0; kk != st->h.n_buckets; ++kk)
        if (!PM_EITHER(st->h.flags, kk)) unp_push(st, st->h.keys[kk] ^ 2, st->h.vals[kk]);
    pm_clear(&st->h);

    for (j = 0; j < len; ++j) cov[j] = cov[j] + 33 < 126 ? (uint8_t)(cov[j] + 33) : 126;
    si[len] = 0;
    if (min_pcv > 0) { /* break the contig where the paired coverage is low (smem.c:254-273) */
        int beg, k;
        mask_pcv(len, (char *)si, pcv, skip, min_pcv);
        for (j = 0; j < len; ++j) if (isupper(si[j])) break;
        beg = j;
        for (j = beg + 1, k = 0; j <= len; ++j) {
            if ((islower(si[j]) || j == len) && isupper(si[j - 1])) {
                st->line_l = 0;
                o_putc(st, '@'); o_puts(st, name); o_putc(st, '_'); o_putl(st, k);
                o_putc(st, '\t'); o_putl(st, j - beg); o_putc(st, '\t'); o_putl(st, n_supp); o_putc(st, '\n');
                o_putsn(st, (char *)si + beg, (size_t)(j - beg)); o_putsn(st, "\n+\n", 3);
                o_putsn(st, (char *)cov + beg, (size_t)(j - beg)); o_putc(st, '\n');
                fwrite(st->line, 1, st->line_l, out);
                ++k;
            }
            if (isupper(si[j]) && islower(si[j - 1])) beg = j;
        }
    } else {
        st->line_l = 0;
        o_putc(st, '@'); o_puts(st, name);
        if (comment) { /* "<number> <rest>": the number is replaced by the support (smem.c:277-284) */
            char *q;
            strtol(comment, &q, 10);
            if (q != comment && isspace((unsigned char)*q)) { o_putc(st, '\t'); o_putl(st, n_supp); o_putc(st, '\t'); o_puts(st, q + 1); }
        }
        if (st->unp_n) {
            o_putsn(st, "\tUR:Z:", 6);
            for (size_t u = 0; u < st->unp_n; ++u) {
                o_putl(st, (long long)st->unp[u].x); o_putc(st, ',');
                o_putl(st, (long long)(st->unp[u].y >> 32)); o_putc(st, ',');
                o_putl(st, (long long)(st->unp[u].y << 32 >> 32)); o_putc(st, ';');
            }
        }
        o_putc(st, '\n');
        for (j = 0; j < len; ++j) si[j] = (uint8_t)"$ACGTN"[si[j]];
        o_putsn(st, (char *)si, (size_t)len); o_putsn(st, "\n+\n", 3);
        o_putsn(st, (char *)cov, (size_t)len); o_putc(st, '\n');
        fwrite(st->line, 1, st->line_l, out);
    }
}

void fmdh_remap_finish(fmdh_remap_state_t *st, FILE *err)
{
    if (err) { /* smem.c:383-387; the pipeline reads `cap` off this line */
        const double avg = (double)st->rec[1] / (double)st->rec[0];
        const double std = sqrt((double)st->rec[2] / (double)st->rec[0] - avg * avg);
        fprintf(err, "[M::fm6_remap] avg = %.2f std = %.2f cap = %d\n", avg, std, (int)(avg + std * 2. + 1.499));
    }
    pm_free(&st->h); free(st->cov); free(st->unp); free(st->line); free(st);
}

/* ---- the command ----------------------------------------------------------------------------- */
#define REMAP_BATCH_BASES (1 << 26)
#define REMAP_BATCH_CALLS (1 << 18)

typedef struct { char *name, *comment; int len; uint64_t off; } contig_t;

/* One batch of contigs.  (1) forward reach of every position on the GPU; (2) the chain of start
 * positions fm6_miter_next visits, x -> x + reach[x], is a pointer chase over that array; (3) every
 * fm6_smem1_core call on the chain is an independent GPU work item (full-length matches only);
 * (4) paircov + printing per contig, in input order. */
static int remap_batch(fmd_dev_t *d, fmdh_remap_state_t *st, contig_t *ctg, size_t n_ctg, uint8_t *bases, uint64_t tot, uint32_t *p_max_len, FILE *out)
{
    size_t n_call = 0, m_call = 1024, i, w;
    int rc = 0;
    uint32_t *reach = (uint32_t *)malloc((size_t)tot * 4);
    fmd_smem_win_t *calls = (fmd_smem_win_t *)malloc(m_call * sizeof(*calls));
    size_t *first = (size_t *)malloc((n_ctg + 1) * sizeof(size_t));
    uint32_t *n_mem = 0;
    fmd_intv_t *mem = 0;
    if (!reach || !calls || !first) { rc = 1; goto done; }
    rc = fmd_reach_batch(d, (size_t)tot, bases, reach);
    if (rc) { fprintf(stderr, "[E::%s] %s\n", __func__, fmd_strerror(rc)); rc = 1; goto done; }
    for (i = 0; i < n_ctg; ++i) {
        first[i] = n_call;
        for (uint32_t x = 0; x < (uint32_t)ctg[i].len;) {
            if (n_call == m_call) { m_call <<= 1; calls = (fmd_smem_win_t *)realloc(calls, m_call * sizeof(*calls)); }
            const uint32_t r = reach[ctg[i].off + x];
            if (r) { /* (a base the index lacks starts no match: the reference's iterator would not return from it) */
                fmd_smem_win_t *c = &calls[n_call++];
                c->seq_off = ctg[i].off; c->seq_len = (uint32_t)ctg[i].len; c->start = x; c->stop = x + 1; c->reserved = FMD_SMEM_WIN_F_FULL;
            }
            x += r ? r : 1;
        }
    }
    first[n_ctg] = n_call;
    /* the calls in GPU batches; a contig is printed once all its calls are in */
    n_mem = (uint32_t *)malloc((n_call ? n_call : 1) * 4);
    {
        uint32_t max_mem = 64;
        size_t done_calls = 0, next_ctg = 0;
        fmd_intv_t *all = 0; size_t all_n = 0, all_m = 0;      /* matches of the calls [kept_from, done_calls) */
        while (rc == 0 && (done_calls < n_call || next_ctg < n_ctg)) {
            const size_t nb = n_call - done_calls < REMAP_BATCH_CALLS ? n_call - done_calls : REMAP_BATCH_CALLS;
            if (nb) {
                for (;;) { /* grow the capacities until no call overflows */
                    int over = 0;
                    free(mem);
                    mem = (fmd_intv_t *)malloc(nb * (size_t)max_mem * sizeof(*mem));
                    if (!mem) { rc = 1; break; }
                    rc = fmd_smem_win_batch(d, nb, bases, tot, calls + done_calls, 0, *p_max_len, max_mem, mem, n_mem + done_calls);
                    if (rc) { fprintf(stderr, "[E::%s] %s\n", __func__, fmd_strerror(rc)); rc = 1; break; }
                    for (w = 0; w < nb; ++w) over |= (int)(n_mem[done_calls + w] >> 31);
                    if (!over) break;
                    if (max_mem >= 65536 || *p_max_len >= (1u << 20)) { rc = 1; break; }
                    max_mem *= 2; *p_max_len *= 2; /* either bound may be the one that was hit */
                }
                if (rc) break;
                for (w = 0; w < nb; ++w) { /* append, in chain order */
                    const uint32_t k = n_mem[done_calls + w];
                    if (all_n + k > all_m) { all_m = (all_n + k) * 2 + 1024; all = (fmd_intv_t *)realloc(all, all_m * sizeof(*all)); }
                    memcpy(all + all_n, mem + w * (size_t)max_mem, k * sizeof(*all)); all_n += k;
                }
                done_calls += nb;
            }
            /* print the contigs that are complete */
            size_t consumed = 0;
            while (next_ctg < n_ctg && first[next_ctg + 1] <= done_calls) {
                size_t n = 0;
                for (w = first[next_ctg]; w < first[next_ctg + 1]; ++w) n += n_mem[w];
                fmdh_remap_contig(st, ctg[next_ctg].name, ctg[next_ctg].comment, ctg[next_ctg].len, bases + ctg[next_ctg].off, all + consumed, n, out);
                consumed += n; ++next_ctg;
            }
            if (consumed) { memmove(all, all + consumed, (all_n - consumed) * sizeof(*all)); all_n -= consumed; }
        }
        free(all);
    }
done:
    free(mem); free(n_mem); free(calls); free(first); free(reach);
    return rc;
}

int fmdh_remap(const char *fmd_path, const char *contig_path, int device, const fmdh_remapopt_t *opt, const char *rank_file, FILE *out)
{
    fmd_dev_t *d = 0;
    fmd_info_t info;
    uint64_t *sorted = 0;
    int rc = fmd_dev_open_file(device, fmd_path, &d), l;
    if (rc) { fprintf(stderr, "[E::%s] cannot load `%s': %s\n", __func__, fmd_path, fmd_strerror(rc)); return 1; }
    fmd_dev_info(d, &info);
    if (rank_file) { /* load_sorted, cmd.c:173-182 */
        FILE *fp = fopen(rank_file, "rb");
        sorted = (uint64_t *)malloc(info.mcnt[1] * 8);
        if (!fp || !sorted || fread(sorted, 8, info.mcnt[1], fp) != info.mcnt[1]) {
            fprintf(stderr, "[E::%s] cannot read the rank file `%s'\n", __func__, rank_file);
            if (fp) fclose(fp);
            free(sorted); fmd_dev_close(d); return 1;
        }
        fclose(fp);
    }
    fmdh_seqio_t *io = fmdh_seq_open(contig_path);
    if (!io) { fprintf(stderr, "[E::%s] cannot open `%s'\n", __func__, contig_path); free(sorted); fmd_dev_close(d); return 1; }
    fmdh_remap_state_t *st = fmdh_remap_new(opt, sorted, info.mcnt[1]);
    contig_t *ctg = 0; size_t n_ctg = 0, m_ctg = 0, cap = 1 << 20;
    uint8_t *bases = (uint8_t *)malloc(cap);
    uint64_t tot = 0;
    uint32_t max_len = 256;
    for (;;) {
        l = fmdh_seq_read(io);
        if (l < 0 || tot >= REMAP_BATCH_BASES) {
            if (n_ctg && (rc = remap_batch(d, st, ctg, n_ctg, bases, tot, &max_len, out)) != 0) break;
            for (size_t i = 0; i < n_ctg; ++i) { free(ctg[i].name); free(ctg[i].comment); }
            n_ctg = 0; tot = 0;
            if (l < 0) break;
        }
        if (n_ctg == m_ctg) { m_ctg = m_ctg ? m_ctg << 1 : 256; ctg = (contig_t *)realloc(ctg, m_ctg * sizeof(*ctg)); }
        /* each contig starts on a 4-byte boundary and keeps one byte for the terminator the printer writes */
        const uint64_t off = (tot + 3) & ~3ull;
        if (off + (size_t)l + 16 > cap) { while (off + (size_t)l + 16 > cap) cap <<= 1; bases = (uint8_t *)realloc(bases, cap); }
        const char *s = fmdh_seq_bases(io);
        for (int i = 0; i < l; ++i) bases[off + i] = fmdh_nt6[(unsigned char)s[i]]; /* seq_nt6_table, smem.c:239-240 */
        bases[off + l] = 0;
        ctg[n_ctg].name = strdup(fmdh_seq_name(io));
        ctg[n_ctg].comment = fmdh_seq_comment(io) ? strdup(fmdh_seq_comment(io)) : 0;
        ctg[n_ctg].len = l; ctg[n_ctg].off = off;
        ++n_ctg; tot = off + (uint64_t)l + 1;
    }
    fmdh_remap_finish(st, rc ? 0 : stderr);
    free(ctg); free(bases); free(sorted);
    fmdh_seq_close(io);
    fmd_dev_close(d);
    return rc;
}
