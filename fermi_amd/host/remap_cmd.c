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

/* ---- per-contig work (smem.c:139-303) --------------------------------------------------------
 * Told in this file's own terms.  A "hit" is a read that matches the contig over its full length; it covers the span
 * [beg, end) of the contig.  With a rank file every hit also names a sequence id, ids come in fours per read pair
 * (two reads x two strands), and the pairing rule is: a hit on the forward strand of one mate WAITS in the pair book
 * under the id its partner will show up with; a hit on the reverse strand looks its id up -- found and close enough,
 * the two spans (trimmed by `skip` at both ends) add to the paired coverage and the waiting entry is closed; otherwise
 * the hit is listed as unpaired.  What still waits when the contig ends is unpaired too and is listed in the book's
 * bucket order.  Insert sizes of the closed pairs feed the avg / std / cap line. */
typedef struct { int beg, end; } span_t;
typedef struct { uint64_t id, where; } loose_t;            /* an unpaired hit: sequence id, packed span */
struct fmdh_remap_state {
    fmdh_remapopt_t opt;
    const uint64_t *sorted;
    uint64_t n_seq;
    pmap_t book;                                           /* waiting forward-strand hits: id -> packed span */
    uint64_t n_pairs, sum_isize, sum_isize2;               /* closed pairs (smem.c:171-173) */
    uint64_t bases_since_reset, reset_bases;               /* the reference starts a new table every 2^28 contig bases (smem.c:380, :237) */
    uint64_t n_resets;
    uint8_t *depth; size_t depth_m;                        /* single + paired coverage of the current contig */
    loose_t *loose; size_t n_loose, m_loose;
    char *text; size_t text_l, text_m;                     /* output record under construction */
};
#define SPAN_MASK (MASK30 << 32 | MASK30)
static inline span_t span_of(uint64_t info) { span_t s; s.beg = (int)(info >> 32 & MASK30); s.end = (int)(info & MASK30); return s; }

fmdh_remap_state_t *fmdh_remap_new(const fmdh_remapopt_t *opt, const uint64_t *sorted, uint64_t n_seq)
{
    fmdh_remap_state_t *st = (fmdh_remap_state_t *)calloc(1, sizeof(*st));
    if (!st) return 0;
    st->opt = *opt; st->sorted = sorted; st->n_seq = n_seq;
    st->reset_bases = (uint64_t)1 << 28;
    { const char *e = getenv("FMD_REMAP_TABLE_BASES"); if (e && atoll(e) > 0) st->reset_bases = (uint64_t)atoll(e); } /* tests: force the restart */
    if (sorted == 0) { st->opt.skip = -1; st->opt.min_pcv = 0; } /* no rank -> index map: nothing is paired, nothing is broken (smem.c:233) */
    return st;
}
uint64_t fmdh_remap_table_resets(const fmdh_remap_state_t *st) { return st->n_resets; }

static void text_room(fmdh_remap_state_t *st, size_t more)
{
    if (st->text_l + more + 1 <= st->text_m) return;
    while (st->text_l + more + 1 > st->text_m) st->text_m = st->text_m ? st->text_m << 1 : 4096;
    st->text = (char *)realloc(st->text, st->text_m);
}
static void text_bytes(fmdh_remap_state_t *st, const void *p, size_t n) { text_room(st, n); memcpy(st->text + st->text_l, p, n); st->text_l += n; }
static void text_fmt(fmdh_remap_state_t *st, const char *fmt, long long a, long long b, long long c)
{
    text_room(st, 96);
    st->text_l += (size_t)snprintf(st->text + st->text_l, 96, fmt, a, b, c);
}
static void loose_add(fmdh_remap_state_t *st, uint64_t id, uint64_t where)
{
    if (st->n_loose == st->m_loose) { st->m_loose = st->m_loose ? st->m_loose << 1 : 16; st->loose = (loose_t *)realloc(st->loose, st->m_loose * sizeof(loose_t)); }
    st->loose[st->n_loose].id = id; st->loose[st->n_loose].where = where; ++st->n_loose;
}
static inline void depth_add(uint8_t *d, int from, int to) { for (int j = from; j < to; ++j) if (d[j] < 255) ++d[j]; }

/* one read of one hit against the pair book (smem.c:158-192) */
static void book_visit(fmdh_remap_state_t *st, uint64_t id, uint64_t where, int len, uint8_t *paired)
{
    if (id & 1) { /* forward strand: wait for the mate, which will come with the id of this read's partner strand */
        const uint32_t slot = pm_put(&st->book, id ^ 3);
        st->book.vals[slot] = where;
        return;
    }
    const uint32_t slot = pm_get(&st->book, id);
    const int waiting = st->book.n_buckets && slot != st->book.n_buckets;
    span_t both;                                            /* from the mate's start to this hit's end */
    both.beg = waiting ? span_of(st->book.vals[slot]).beg : 0;
    both.end = span_of(where).end;
    if (!waiting || both.end - both.beg >= st->opt.max_dist) { loose_add(st, id ^ 1, where); return; }
    {
        const int isize = both.end - both.beg;
        ++st->n_pairs; st->sum_isize += (uint64_t)isize; st->sum_isize2 += (uint64_t)(isize * isize);
    }
    both.beg += st->opt.skip; both.end -= st->opt.skip;
    if (both.beg > both.end) { const int t = both.beg; both.beg = both.end; both.end = t; }
    depth_add(paired, both.beg < 0 ? 0 : both.beg, both.end > len ? len : both.end);
    pm_del(&st->book, slot);
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

/* the contig cut at its lower-case stretches, one FASTQ record per upper-case piece (smem.c:254-273) */
static void print_pieces(fmdh_remap_state_t *st, const char *name, int len, const char *seq, const uint8_t *cov, int n_hits, FILE *out)
{
    int from = -1, piece = 0;
    for (int j = 0; j <= len; ++j) {
        const int up = j < len && isupper((unsigned char)seq[j]);
        if (up && from < 0) from = j;
        if (!up && from >= 0) {
            st->text_l = 0;                       /* every maximal upper-case run is a piece */
            text_bytes(st, "@", 1); text_bytes(st, name, strlen(name));
            text_fmt(st, "_%lld\t%lld\t%lld\n", piece, j - from, n_hits);
            text_bytes(st, seq + from, (size_t)(j - from)); text_bytes(st, "\n+\n", 3);
            text_bytes(st, cov + from, (size_t)(j - from)); text_bytes(st, "\n", 1);
            fwrite(st->text, 1, st->text_l, out);
            ++piece;
            from = -1;
        }
    }
}

/* the whole contig, its support in the header, unpaired hits in a UR:Z tag (smem.c:275-300) */
static void print_whole(fmdh_remap_state_t *st, const char *name, const char *comment, int len, uint8_t *bases, const uint8_t *cov, int n_hits, FILE *out)
{
    st->text_l = 0;
    text_bytes(st, "@", 1); text_bytes(st, name, strlen(name));
    if (comment) { /* "<number> <rest>": the number is replaced by the support (smem.c:277-284) */
        char *rest;
        strtol(comment, &rest, 10);
        if (rest != comment && isspace((unsigned char)*rest)) { text_fmt(st, "\t%lld\t", n_hits, 0, 0); text_bytes(st, rest + 1, strlen(rest + 1)); }
    }
    if (st->n_loose) {
        text_bytes(st, "\tUR:Z:", 6);
        for (size_t u = 0; u < st->n_loose; ++u)
            text_fmt(st, "%lld,%lld,%lld;", (long long)st->loose[u].id, (long long)(st->loose[u].where >> 32), (long long)(st->loose[u].where << 32 >> 32));
    }
    text_bytes(st, "\n", 1);
    for (int j = 0; j < len; ++j) bases[j] = (uint8_t)"$ACGTN"[bases[j]];
    text_bytes(st, bases, (size_t)len); text_bytes(st, "\n+\n", 3);
    text_bytes(st, cov, (size_t)len); text_bytes(st, "\n", 1);
    fwrite(st->text, 1, st->text_l, out);
}

void fmdh_remap_contig(fmdh_remap_state_t *st, const char *name, const char *comment, int len, uint8_t *bases, const fmd_intv_t *mem, size_t n_mem, FILE *out)
{
    const int pairing = st->opt.skip > 0 && st->sorted != 0;
    int n_hits = 0;
    if ((size_t)(len + 1) * 2 > st->depth_m) { st->depth_m = (size_t)(len + 1) * 2; st->depth = (uint8_t *)realloc(st->depth, st->depth_m); }
    memset(st->depth, 0, (size_t)(len + 1) * 2);
    uint8_t *cov = st->depth, *paired = st->depth + len + 1;
    st->n_loose = 0;
    if (st->book.n_buckets >= 256) pm_free(&st->book);       /* a table that grew large is dropped before the next contig (smem.c:241-244) */
    /* the hits, in the order the iterator yields them */
    for (size_t m = 0; m < n_mem; ++m) {
        const fmd_intv_t *hit = &mem[m];
        if (!(hit->info >> 63) || hit->x[1] >= st->n_seq) continue;      /* closed by sentinels on both sides = a whole read */
        const span_t sp = span_of(hit->info);
        depth_add(cov, sp.beg, sp.end);
        ++n_hits;
        if (!pairing) continue;
        for (uint64_t r = 0; r < hit->x[2]; ++r)                          /* x[1]: the interval of the reverse strand */
            book_visit(st, st->sorted[hit->x[1] + r] >> 2, hit->info & SPAN_MASK, len, paired);
    }
    for (uint32_t slot = 0; slot != st->book.n_buckets; ++slot)           /* still waiting: unpaired, in bucket order */
        if (!PM_EITHER(st->book.flags, slot)) loose_add(st, st->book.keys[slot] ^ 2, st->book.vals[slot]);
    pm_clear(&st->book);

    for (int j = 0; j < len; ++j) cov[j] = cov[j] + 33 < 126 ? (uint8_t)(cov[j] + 33) : 126;
    bases[len] = 0;
    if (st->opt.min_pcv > 0) {
        mask_pcv(len, (char *)bases, paired, st->opt.skip, st->opt.min_pcv);
        print_pieces(st, name, len, (const char *)bases, cov, n_hits, out);
    } else print_whole(st, name, comment, len, bases, cov, n_hits, out);
    /* the reference hands its contigs to paircov_all in batches of >= 2^28 bases and every batch starts with a new table
     * (smem.c:237, :380): the batch ends with the contig that reaches the limit */
    st->bases_since_reset += (uint64_t)len;
    if (st->bases_since_reset >= st->reset_bases) { pm_free(&st->book); st->bases_since_reset = 0; ++st->n_resets; }
}

void fmdh_remap_finish(fmdh_remap_state_t *st, FILE *err)
{
    if (err) { /* smem.c:383-387; the pipeline reads `cap` off this line */
        const double avg = (double)st->sum_isize / (double)st->n_pairs;
        const double std = sqrt((double)st->sum_isize2 / (double)st->n_pairs - avg * avg);
        fprintf(err, "[M::fm6_remap] avg = %.2f std = %.2f cap = %d\n", avg, std, (int)(avg + std * 2. + 1.499));
    }
    pm_free(&st->book); free(st->depth); free(st->loose); free(st->text); free(st);
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
