/* fmd_host.h -- host-side C helpers of the MI355X FMD path (plain C, no GPU code).
 * File formats follow the reference byte for byte so that files are interchangeable with fermi:
 *   RLD\2 container  rld.c:242-263 (layout), rld.c:111-175 (block/run encoding), rld.c:186-224 (frames)
 *   RLE\6 stream     ropebwt.c:132-136
 */
#ifndef FMD_HOST_H
#define FMD_HOST_H
#include <stddef.h>
#include <stdint.h>
#include <string.h>
#include <stdio.h>
#include <pthread.h>
#include "fmd_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* Encode a run-length byte stream (`len<<3 | sym`, len 1..31, adjacent equal symbols are
 * merged like rld_enc does, rld.c:177-184) into an RLD\2 .fmd file.  Returns 0 or -errno. */
int fmdh_write_rld_from_rle6(const uint8_t *runs, uint64_t n_bytes, const char *path);
/* Write the same stream as an RLE\6 .fmd (what `fermi ropebwt -b` emits). */
int fmdh_write_rle6(const uint8_t *runs, uint64_t n_bytes, const char *path);
/* Plain BWT string (nt6 bytes) -> RLD\2 file. */
int fmdh_write_rld_from_bwt(const uint8_t *bwt, uint64_t n, const char *path);

/* nt6 conversion table (seq.c:12-21) and helpers */
extern const uint8_t fmdh_nt6[256];
/* cmd.c:457-463: an even-length read equal to its own reverse complement loses its last base.
 * Returns the (possibly reduced) length. */
uint32_t fmdh_trim_palindrome(const uint8_t *s, uint32_t len);

/* ---- `fermi unitig` on top of the GPU overlap table (unitig.c:227-362, mag.c:149-174) ----
 * The table holds one PACKED row per sequence id (include/fmd_hip.h, fmd_ovlp_pack_dev): the record, its neighbours
 * and its bases.  It is kept in the shards it was computed in: with N GPUs, id i is row i / N of shard i % N (the
 * reference's worker interleave, unitig.c:333, 398-399), so nothing is re-interleaved after the GPUs finish. */
typedef struct {
    uint64_t n;                  /* rows */
    fmd_ovlp_rec_t *rec;         /* n packed records */
    uint64_t *off;               /* n: byte offset of the row's variable part inside chunk[row >> chunk_shift] */
    uint8_t **chunk;             /* ceil(n / 2^chunk_shift) buffers (fmd_ovlp_packed_free) */
    uint32_t chunk_shift, max_nei, seq_stride;
} fmdh_ovlp_shard_t;
typedef struct {
    uint64_t n;                  /* sequence ids 0 .. n-1 */
    int n_shards;
    fmdh_ovlp_shard_t *shard;    /* n_shards */
    /* the few rows whose lists did not fit the capacities of the first pass, recomputed with roomier ones */
    uint32_t *side_of;           /* n entries: row in `side`, 0xffffffff = not there; or NULL */
    fmdh_ovlp_shard_t side;
    /* optional, filled by fmdh_ovlp_table_link(): */
    uint32_t *row_of;            /* n entries: `$read$` interval start k[0] -> smallest id with that interval, 0xffffffff = none */
    struct fmdh_link *link;      /* n entries */
} fmdh_ovlp_table_t;
/* what the walk needs to step from a row to the next without touching the neighbour list: rows of the unique
 * neighbour and of its reverse strand (0xffffffff: none / more than one neighbour) */
typedef struct fmdh_link { uint32_t nxt, rev; } fmdh_link_t;
typedef struct { const fmd_ovlp_rec_t *rec; const fmd_intv_t *nei; const uint8_t *var; uint32_t max_nei; } fmdh_row_t;
static inline fmdh_row_t fmdh_table_row(const fmdh_ovlp_table_t *t, uint64_t id)
{
    const fmdh_ovlp_shard_t *s;
    uint64_t r;
    fmdh_row_t x;
    if (t->side_of && t->side_of[id] != 0xffffffffu) { s = &t->side; r = t->side_of[id]; }
    else if (t->n_shards == 1) { s = &t->shard[0]; r = id; }
    else { s = &t->shard[id % (uint64_t)t->n_shards]; r = id / (uint64_t)t->n_shards; }
    x.rec = &s->rec[r];
    x.var = s->chunk[r >> s->chunk_shift] + s->off[r];
    x.nei = (const fmd_intv_t *)x.var;
    x.max_nei = s->max_nei;
    return x;
}
/* bases [from, from + n) of a row (the sequence, then the bases fm6_get_nei appended), as nt6 codes */
static inline void fmdh_row_bases(const fmdh_row_t *x, uint32_t from, uint32_t n, char *dst)
{
    uint32_t j = 0;
    if (!(x->rec->flags & FMD_OVLP_F_PACK4)) {   /* 2 bits per base: four at a time once `from + j` is a multiple of four */
        const uint8_t *s = x->var + fmd_ovlp_row_nei(x->rec, x->max_nei) * 32;
        for (; j < n && ((from + j) & 3); ++j) dst[j] = (char)(((s[(from + j) >> 2] >> (2 * ((from + j) & 3))) & 3) + 1);
        for (; j + 4 <= n; j += 4) {
            const uint32_t b = s[(from + j) >> 2];
            const uint32_t v = ((b & 3u) | (b & 0xcu) << 6 | (b & 0x30u) << 12 | (b & 0xc0u) << 18) + 0x01010101u;   /* little endian: base j in the low byte */
            memcpy(dst + j, &v, 4);
        }
    }
    for (; j < n; ++j) dst[j] = (char)fmd_ovlp_row_base(x->rec, x->max_nei, x->var, from + j);
}
/* ---- the walk's own table (slim_table.c): 32 bytes per sequence id that a plain step of the walk reads as ONE line, and a short variable part.
 * What `unitig` holds in host memory: 44.5 bytes per id on 100-base reads where the packed rows above took 150-190. */
#define FMDH_SLIM_CHUNK_SHIFT 22
#define FMDH_W_ST_MASK 3u          /* bits & 3: 0 = a read the walk can extend, 1 = short (unitig.c:288), 2 = contained (:292), 3 = flagged (a capacity was exceeded) */
#define FMDH_W_ST_SHORT 1
#define FMDH_W_ST_CONTAINED 2
#define FMDH_W_ST_INVALID 3
#define FMDH_W_PLAIN 4u            /* one neighbour, check_left decided, the appended bases in the line, nothing wide: the step reads w[id] and the line of w[nxt] only */
#define FMDH_W_CL 8u               /* check_left (unitig.c:206-225) < 0 on the edge to the unique neighbour */
#define FMDH_W_UNDEC 16u           /* that edge is not decided (or the neighbour has no row): the table is incomplete */
#define FMDH_W_BIG 32u             /* the variable part is the record itself (64 bytes), its neighbours (32 each), all bases 4 bits each */
#define FMDH_W_EXTVAR 64u          /* the appended bases are in the variable part, 4 bits each */
#define FMDH_W_XVAR 128u           /* the variable part is in xvar (rows computed again), voff in units of 8 bytes */
#define FMDH_W_EXT_INLINE 24
/* What a line does NOT hold because another one does: k[1] of `$read$` is k[0] of the read's other strand (the bi-interval of a string is
 * (interval, interval of its reverse complement, size) and '$' is its own complement: exact.c:72-88), and the other strand's row is id ^ 1 --
 * the other half of the same 64 bytes; the rank fm_retrieve returns (exact.c:59-70: the place of the read's '$' among the sentinels) lies
 * in [k[0], k[0] + k[2]), so one byte of it is new; the length is the table's common one unless the row says otherwise. */
typedef struct {
    uint32_t nxt;                  /* id of the unique neighbour's row, 0xffffffff = none */
    uint32_t far;                  /* the row eight accepted links on (a prefetch hint; 0xffffffff = none / not built) */
    uint32_t k0;                   /* k[0] of the record (every row that is not short or flagged, W_BIG ones included when it fits): k[1] of row id ^ 1 */
    uint32_t voff;                 /* where the variable part starts in its chunk (bytes) */
    uint16_t rbeg;                 /* 0xffff = fm6_get_nei returned -1 */
    uint8_t ext[6];                /* the appended bases, (code - 1) in 2 bits each, first one lowest */
    uint8_t ext_len, n_nei, k2, bits;
    uint8_t dr;                    /* rank - k[0] */
    uint8_t vfl;                   /* FMDH_V_* */
    uint16_t ov;                   /* a table linked on the device, ONE neighbour: the overlap length (its x[0] is k0 of the row the link leads to) */
} fmdh_wrec_t;                     /* 32 bytes */
/* variable part of a row that is not W_BIG: [len u16: V_LEN_VAR], the neighbours, [appended bases, 4 bits each: W_EXTVAR],
 * [the sequence: 2 bits per base, or 4 with V_SEED_N: V_HAS_SEED -- even ids, and rows computed again].  On reads of one length with one
 * neighbour each: the sequence of every other row and nothing else.
 * The neighbours: n_nei x {x[0] of `$neighbour$` u32, overlap u16} where there are several, and nothing for ONE (w.ov; its x[0] is k0 of the row the
 * link leads to -- a table the host links carries it in w.nxt until then). */
#define FMDH_V_HAS_OVLP 1u         /* overlap_intv found a candidate (rec.n_ovlp != 0) */
#define FMDH_V_SEED_N 2u
#define FMDH_V_HAS_SEED 4u
#define FMDH_V_RES_SHIFT 3         /* rec.reserved (0 / 1 / 2) as it arrived */
#define FMDH_V_LEN_VAR 32u         /* the length is not the table's len0: it leads the variable part */
typedef struct fmdh_slim {
    uint64_t n; int n_shards; uint32_t chunk_shift; uint64_t cps;   /* ids; variable parts are kept per shard (id % n_shards) in chunks of 2^chunk_shift rows, cps chunks per shard */
    fmdh_wrec_t *w;
    uint8_t **var; uint64_t *var_len;
    uint8_t *xvar; uint64_t x_len, x_cap;
    uint32_t max_nei;                     /* the longest neighbour list of any row */
    uint64_t big_k2;                      /* widest k[2] a line holds (255) */
    double t_add[4];                      /* FMD_TIMING: seconds in fmdh_slim_add's sizes pass, allocation, rows pass; calls */
    int32_t len0; int len0_set;           /* the length rows have unless they say otherwise (that of the first row that arrived) */
    int host_link, linked;                /* host_link: fmdh_slim_link_host will link the rows -- until it has (linked), w.nxt = x[0] of the unique neighbour, w.far = rec.lfork */
    uint32_t *row_of;                     /* k[0] -> the smallest id with that interval; until fmdh_slim_finalize */
    uint64_t *und; uint32_t *und_rev; uint64_t n_und, m_und;   /* rows whose check_left is open, ascending, and the row of the neighbour's reverse strand */
    pthread_mutex_t mu;
} fmdh_slim_t;
fmdh_slim_t *fmdh_slim_new(uint64_t n, int n_shards, int host_link, uint32_t chunk_shift);
void fmdh_slim_free(fmdh_slim_t *s);
uint64_t fmdh_slim_bytes(const fmdh_slim_t *s);
int fmdh_host_threads(void);       /* FMD_HOST_THREADS, default 16 */
double fmdh_thp_gb(void);          /* anonymous memory in transparent huge pages, GB (-1: unknown) */
double fmdh_rss_gb(int peak);       /* resident set of this process now (VmRSS of /proc/self/status) or its peak so far (VmHWM -- NOT ru_maxrss, which
                                    * inherits the high-water mark of whatever exec'ed the program), in GB: the FMD_TIMING lines */
void fmdh_par_for(int nt, void (*fn)(void *ctx, int tid, int nt), void *ctx);   /* fn(ctx, tid, nt) on nt threads (at most 64), joined */
/* rows (chunk << FMDH_SLIM_CHUNK_SHIFT) .. + nr of shard g (id = g + n_shards * row) from packed rows as fmd_ovlp_pack_dev writes them: rec[nr], off[nr]
 * (offsets into var); the three buffers are the caller's and are not kept.  Chunks of different shards may be added concurrently. */
int fmdh_slim_add(fmdh_slim_t *s, int g, uint64_t chunk, const fmd_ovlp_rec_t *rec, const uint64_t *off, const uint8_t *var, uint32_t max_nei, uint64_t nr, int n_threads);
int fmdh_slim_replace(fmdh_slim_t *s, const uint64_t *ids, const fmd_ovlp_rec_t *rec, const uint64_t *off, const uint8_t *var, uint32_t max_nei, uint64_t n, int n_threads);
int fmdh_slim_add_ids(fmdh_slim_t *s, const uint64_t *ids, const fmd_ovlp_rec_t *rec, const uint64_t *off, const uint8_t *var, uint32_t max_nei, uint64_t n, int n_threads);
int fmdh_slim_link_fold(fmdh_slim_t *s, uint64_t first, uint64_t n, const struct fmdh_link *link, const uint8_t *reserved);
int fmdh_slim_link_host(fmdh_slim_t *s, int n_threads);
void fmdh_slim_undecided(const fmdh_slim_t *s, const uint64_t **ids, uint64_t *n);
int fmdh_slim_set_reserved(fmdh_slim_t *s, const uint64_t *ids, const uint16_t *vals, uint64_t n);
int fmdh_slim_finalize(fmdh_slim_t *s, int n_threads);
int fmdh_slim_from_table(const fmdh_ovlp_table_t *t, int n_threads, fmdh_slim_t **out);
int fmdh_slim_stats(const fmdh_ovlp_table_t *t, uint64_t out[6]);

static inline const uint8_t *fmdh_slim_var(const fmdh_slim_t *s, uint64_t id)
{
    const fmdh_wrec_t *w = &s->w[id];
    if (w->bits & FMDH_W_XVAR) return s->xvar + (uint64_t)w->voff * 8;
    if (s->n_shards == 1) return s->var[id >> s->chunk_shift] + w->voff;
    return s->var[(id % (uint64_t)s->n_shards) * s->cps + ((id / (uint64_t)s->n_shards) >> s->chunk_shift)] + w->voff;
}
/* a row as the walk's general code reads it */
typedef struct {
    const uint8_t *var, *nei;      /* nei: the neighbour entries (6 bytes each where there are several; 32 in a W_BIG row) */
    uint64_t rank, k[3];
    int32_t len, rbeg, ext_len, n_nei, n_stored;
    int status, has_ovlp, reserved, big, nei_bytes;   /* nei_bytes: of the whole neighbour block */
    uint8_t bits, vflags;
} fmdh_rowv_t;
static inline void fmdh_slim_row(const fmdh_slim_t *s, uint64_t id, fmdh_rowv_t *v)
{
    const fmdh_wrec_t *w = &s->w[id];
    const uint8_t *p = fmdh_slim_var(s, id);
    v->var = p; v->bits = w->bits; v->status = (int)(w->bits & FMDH_W_ST_MASK); v->big = (w->bits & FMDH_W_BIG) != 0;
    if (v->big) {
        fmd_ovlp_rec_t r;
        memcpy(&r, p, 64);
        v->rank = r.rank; v->k[0] = r.k[0]; v->k[1] = r.k[1]; v->k[2] = r.k[2]; v->len = r.len; v->rbeg = r.rbeg; v->ext_len = r.ext_len; v->n_nei = r.n_nei;
        v->has_ovlp = r.n_ovlp != 0; v->reserved = r.reserved; v->vflags = FMDH_V_HAS_SEED | FMDH_V_SEED_N;
        v->nei = p + 64;
    } else {
        const uint64_t other = id ^ 1;
        v->vflags = w->vfl; v->has_ovlp = (w->vfl & FMDH_V_HAS_OVLP) != 0; v->reserved = (w->vfl >> FMDH_V_RES_SHIFT) & 3;
        if (w->vfl & FMDH_V_LEN_VAR) { uint16_t ln; memcpy(&ln, p, 2); v->len = ln; p += 2; } else v->len = s->len0;
        v->k[0] = w->k0; v->k[1] = other < s->n ? s->w[other].k0 : ~0ull; v->k[2] = w->k2; v->rank = (uint64_t)w->k0 + w->dr;
        v->rbeg = w->rbeg == 0xffff ? -1 : (int32_t)w->rbeg; v->ext_len = w->ext_len; v->n_nei = w->n_nei;
        v->nei = p;
    }
    if (v->status != 0) { v->n_nei = 0; v->rbeg = -1; v->ext_len = 0; }
    v->n_stored = v->n_nei;
    v->nei_bytes = v->big ? v->n_stored * 32 : v->n_stored == 1 ? 0 : v->n_stored * 6;
}
/* neighbour k of row id: x[0] (and x[1] where the table keeps it, else ~0) of `$neighbour$`, the overlap length */
static inline void fmdh_slim_nei(const fmdh_slim_t *s, uint64_t id, const fmdh_rowv_t *v, int k, uint64_t *x0, uint64_t *x1, uint64_t *info)
{
    if (v->big) { fmd_intv_t e; memcpy(&e, v->nei + (size_t)k * 32, 32); *x0 = e.x[0]; *x1 = e.x[1]; *info = e.info; }
    else if (v->n_stored == 1) {
        const uint32_t nxt = s->w[id].nxt;
        *info = s->w[id].ov; *x1 = ~0ull; *x0 = ~0ull;
        if (nxt != 0xffffffffu) {
            if (s->w[nxt].bits & FMDH_W_BIG) { fmd_ovlp_rec_t r; memcpy(&r, fmdh_slim_var(s, nxt), 64); *x0 = r.k[0]; }
            else *x0 = s->w[nxt].k0;
        }
    } else { uint32_t a; uint16_t c; const uint8_t *q = v->nei + (size_t)k * 6; memcpy(&a, q, 4); memcpy(&c, q + 4, 2); *x0 = a; *x1 = ~0ull; *info = c; }
}
/* the bases fm6_get_nei appended (unitig.c:139), nt6 codes */
static inline void fmdh_slim_ext(const fmdh_slim_t *s, uint64_t id, const fmdh_rowv_t *v, char *dst)
{
    int j;
    if (v->big) { const uint8_t *q = v->nei + v->nei_bytes; for (j = 0; j < v->ext_len; ++j) { const int z = v->len + j; dst[j] = (char)((q[z >> 1] >> (4 * (z & 1))) & 15); } }
    else if (v->bits & FMDH_W_EXTVAR) { const uint8_t *q = v->nei + v->nei_bytes; for (j = 0; j < v->ext_len; ++j) dst[j] = (char)((q[j >> 1] >> (4 * (j & 1))) & 15); }
    else { const uint8_t *q = s->w[id].ext; for (j = 0; j < v->ext_len; ++j) dst[j] = (char)(((q[j >> 2] >> (2 * (j & 3))) & 3) + 1); }
}
/* the sequence of a row that holds its own (V_HAS_SEED), nt6 codes; 0 = this row does not */
static inline int fmdh_slim_own_seq(const fmdh_rowv_t *v, char *dst)
{
    const uint8_t *q;
    int j;
    if (!(v->vflags & FMDH_V_HAS_SEED) || v->status != 0) return 0;
    if (v->big) { q = v->nei + v->nei_bytes; for (j = 0; j < v->len; ++j) dst[j] = (char)((q[j >> 1] >> (4 * (j & 1))) & 15); return 1; }
    q = v->nei + v->nei_bytes + ((v->bits & FMDH_W_EXTVAR) ? (size_t)(v->ext_len + 1) / 2 : 0);
    if (v->vflags & FMDH_V_SEED_N) for (j = 0; j < v->len; ++j) dst[j] = (char)((q[j >> 1] >> (4 * (j & 1))) & 15);
    else {
        for (j = 0; j + 4 <= v->len; j += 4) {
            const uint32_t b = q[j >> 2], x = ((b & 3u) | (b & 0xcu) << 6 | (b & 0x30u) << 12 | (b & 0xc0u) << 18) + 0x01010101u;
            memcpy(dst + j, &x, 4);
        }
        for (; j < v->len; ++j) dst[j] = (char)(((q[j >> 2] >> (2 * (j & 3))) & 3) + 1);
    }
    return 1;
}

/* One parallel pass over a complete table (n_threads host threads): row_of, link, and check_left_simple's verdict
 * (unitig.c:186-204) for every row with a unique neighbour, decided from the lfork of the neighbour's reverse strand
 * (include/fmd_hip.h) and written to rec.reserved (0 / 1).  Rows it cannot decide are returned in *undecided
 * (malloc'ed ids, *n_undecided of them; rec.reserved stays 2): the caller runs fmd_ovlp_check_left on those. */
int fmdh_ovlp_table_link(fmdh_ovlp_table_t *t, int n_threads, uint64_t **undecided, uint64_t *n_undecided);
/* Build the table of all n_seq sequence ids on the GPUs devices[0..n_dev): one host thread and one replica of the index
 * per device, GPU g = the rows of ids g, g + n_dev, ..., streamed chunk by chunk into the slim table; rows that overflow the
 * capacities are recomputed (device 0) with the capacities raised until they fit.  A device may be listed more than once (two
 * replicas on one GPU).  *out is released with fmdh_slim_free. */
int fmdh_slim_build(const char *fmd_path, int n_dev, const int *devices, int min_match, fmdh_slim_t **out, uint64_t *n_seq);
void fmdh_slim_last_build(double out[4]);   /* seconds of the last build: slowest replica's load, slowest replica's rows, whole build; bytes of the table */
void fmdh_ovlp_table_free(fmdh_ovlp_table_t *t);
/* Replays the single-threaded walk and writes the MAG records `fermi unitig -t1` prints. */
int fmdh_unitig_walk(const fmdh_ovlp_table_t *t, uint64_t n_seq, int min_match, const uint64_t *sorted /* or NULL */, FILE *out);
/* flags: FMDH_WALK_FULL_RECORDS = a record that holds a base other than A/C/G/T is written whole, NUL included, as mag_g_print writes
 * a graph (mag.c:176-188, fwrite) -- `fermi unitig` cuts it at the NUL (unitig.c:354, fputs), which is the default here */
#define FMDH_WALK_FULL_RECORDS 1
int fmdh_unitig_walk_opt(const fmdh_ovlp_table_t *t, uint64_t n_seq, int min_match, const uint64_t *sorted /* or NULL */, FILE *out, int flags);
/* the walk itself: over the slim table (writes the skip list into t->w[].far) */
int fmdh_unitig_walk_slim(fmdh_slim_t *t, uint64_t n_seq, int min_match, const uint64_t *sorted /* or NULL */, FILE *out, int flags);
/* Whole command: replicate the .fmd on the listed GPUs, build the table, walk, print.  `fermi unitig -l min_match <fn>`
 * (cmd.c:184-216); the reference's fm6_unitig gives seeds i = j (mod n_threads) to worker j (unitig.c:394-404), here
 * GPU g computes the rows of ids i = g (mod n_dev) and ONE deterministic walk consumes them (the output is that of -t1
 * whatever n_dev is). */
int fmdh_unitig(const char *fmd_path, int n_dev, const int *devices, int min_match, const char *rank_file /* -r, or NULL */, FILE *out);

/* ---- the root of an N-PROCESS overlap job (include/fmd_hip.h, fmd_ovlp_dist_*: one process per GPU, RCCL) as `unitig` needs it: the root keeps no packed
 * table (fmd_ovlp_dist_cfg_t.host_table = 2) -- every piece of every peer is folded into the slim rows as it arrives (row_sink = fmdh_dist_root_sink,
 * sink_ctx = the object), 44.5 bytes per id instead of 125 -- and when the step has returned, fmdh_dist_root_finish does on the root's own GPU what
 * fmdh_slim_build does after its rows: the rows that exceeded a capacity again, links and check_left by host threads, the plain steps.  The table it hands
 * over is the one fmdh_unitig_walk_slim walks (unitig.c:394-404: the reference joins its workers' vertices into one graph; here the graph is walked from
 * the table).  (fermi_amd/host/ovlp_table.c) */
typedef struct fmdh_dist_root fmdh_dist_root_t;
fmdh_dist_root_t *fmdh_dist_root_new(uint64_t n_seq, uint32_t max_len);
int fmdh_dist_root_sink(void *ctx, uint64_t n_rows, const uint32_t *ids, const fmd_ovlp_rec_t *prec, const uint64_t *off, const uint8_t *var, uint32_t max_nei);
uint64_t fmdh_dist_root_rows(const fmdh_dist_root_t *r);       /* rows folded so far */
int fmdh_dist_root_finish(fmdh_dist_root_t *r, fmd_dev_t *dev, int min_match, fmdh_slim_t **out);   /* 0 and *out (the caller's: fmdh_slim_free), or 1; frees r either way */
void fmdh_dist_root_free(fmdh_dist_root_t *r);
/* `fermi seqsort <reads.fmd>` (seqsort.c:37-70): *sorted is malloc'ed, n = mcnt[1] entries */
int fmdh_seqsort(const char *fmd_path, int device, uint64_t **sorted, uint64_t *n);

/* ---- FASTA/FASTQ input (kseq.h semantics) ---- */
typedef struct fmdh_seqio fmdh_seqio_t;
fmdh_seqio_t *fmdh_seq_open(const char *fn);           /* "-" = stdin; gzip transparent */
int fmdh_seq_read(fmdh_seqio_t *io);                   /* length, -1 = end of file, -2 = bad quality */
const char *fmdh_seq_name(const fmdh_seqio_t *io);
char *fmdh_seq_bases(fmdh_seqio_t *io);
char *fmdh_seq_qual(fmdh_seqio_t *io);                 /* NULL for FASTA */
const char *fmdh_seq_comment(const fmdh_seqio_t *io);  /* NULL when the header has none */
void fmdh_seq_close(fmdh_seqio_t *io);
fmdh_seqio_t *fmdh_seq_open_mem(const void *p, size_t n);   /* the same reader over bytes in memory */
size_t fmdh_seq_mem_pos(const fmdh_seqio_t *io);
int fmdh_seq_between_records(const fmdh_seqio_t *io);
/* ---- seqpar.c: a plain (not compressed) FASTA/FASTQ FILE parsed by several threads, records in file order, the bytes fmdh_seq_read gives.
 * The file is mapped; a span of it is cut at guessed record starts, every piece parsed by its own fmdh_seq reader, and a piece counts only
 * if the piece before it ended exactly where it starts, between two records (the first piece starts where the previous span ended: no guess)
 * -- otherwise the span is parsed again by one reader.  fmdh_pseq_open returns NULL for input that cannot be mapped (stdin, gzip): the
 * caller then reads it with fmdh_seq_read as before. */
typedef struct { char *seq, *qual; uint32_t *len; size_t n, bytes; int has_qual, bad; size_t m_bytes, m_n; } fmdh_ppart_t;   /* reads of a piece, back to back */
typedef struct fmdh_pseq fmdh_pseq_t;
fmdh_pseq_t *fmdh_pseq_open(const char *fn, int n_threads, size_t span_bytes);
int fmdh_pseq_next(fmdh_pseq_t *r, fmdh_ppart_t **parts, int *n_parts);   /* 1 = parts[0 .. *n_parts) hold the next reads; 0 = end of file; -2 = truncated quality */
void fmdh_pseq_close(fmdh_pseq_t *r);

/* `fermi exact [-s] <idx> <src.fa>` (cmd.c:292-331) */
int fmdh_exact(const char *fmd_path, const char *fa_path, int device, int self_match, FILE *out);

/* `fermi chkbwt [-p] [-r] <idx>` (cmd.c:47-130) and `fermi unpack [-i INT]... <idx>` (cmd.c:132-171) */
int fmdh_chkbwt(const char *fmd_path, int device, int plain, int check_rank, FILE *out);
int fmdh_unpack(const char *fmd_path, int device, int n_list, const uint64_t *list /* or NULL: all */, FILE *out);

/* `fermi remap [-l skip] [-c min_pcv] [-D max_dist] [-r rank] <reads.fmd> <contigs.fq>` (cmd.c:218-251,
 * smem.c:114-394): coverage of every contig by the reads that match it full length, paired-end
 * coverage when a rank file is given; prints what `fermi remap -t1` prints. */
typedef struct { int skip, min_pcv, max_dist; } fmdh_remapopt_t;
int fmdh_remap(const char *fmd_path, const char *contig_path, int device, const fmdh_remapopt_t *opt, const char *rank_file, FILE *out);
/* the per-contig part alone (paircov + printing, smem.c:139-303) over the contig's full-length matches,
 * sorted by start (state = fmdh_remap_new(); one state per run: the pair table lives across contigs) */
typedef struct fmdh_remap_state fmdh_remap_state_t;
fmdh_remap_state_t *fmdh_remap_new(const fmdh_remapopt_t *opt, const uint64_t *sorted /* or NULL */, uint64_t n_seq);
void fmdh_remap_contig(fmdh_remap_state_t *st, const char *name, const char *comment, int len, uint8_t *nt6 /* len + 1 bytes, overwritten */,
                       const fmd_intv_t *mem, size_t n_mem, FILE *out);
void fmdh_remap_finish(fmdh_remap_state_t *st, FILE *err);
uint64_t fmdh_remap_table_resets(const fmdh_remap_state_t *st);  /* how often the pair table was started afresh (every 2^28 contig bases, smem.c:380) */ /* the `avg = .. std = .. cap = ..` line, then frees st */

/* ---- the in-memory API (fermi.h:119-123): reads in ONE buffer of l bytes, a NUL after every read, letters or nt6 codes.
 * fmdh_api_unitig = fm6_api_unitig (unitig.c:413-434: fm6_build2 -- no palindrome trimming, build.c:52-70 --, then unitig_core with one
 * thread) with the graph written as mag_g_print would print it before any cleaning (mag.c:149-174): the records of `fermi unitig`.
 * min_match < 0: a third of the lower-quartile read length (unitig.c:418-421).  seq is converted to nt6 in place, as the reference does.
 * fmdh_api_correct = fm6_api_correct (correct.c:464-511): k-mer harvest (k = kmer, or 19), ec_fix over every read; on return seq holds
 * upper-case letters where a base was kept and lower-case ones where it was corrected, qual 36 ('$') under the corrected bases; qual may
 * be NULL (quality 20 everywhere).  The reference leaves the jump heuristic's step uninitialised there (fmecopt_t opt on the stack,
 * correct.c:471-474): it is a parameter here (the CLI default is 5, 0 switches the heuristic off). */
int fmdh_api_unitig(int device, int min_match, int64_t l, char *seq, FILE *out);
int fmdh_api_correct(int device, int kmer, int step, int64_t l, char *seq, char *qual);
int fmdh_api_seqlen(int64_t l, const char *seq, double quantile);   /* fm6_api_seqlen, seq.c:430-446 */
int fmdh_slim_build_dev(fmd_dev_t *dev, int min_match, fmdh_slim_t **out, uint64_t *n_seq_out);

/* `fermi build -o out.fmd <in.fa>` (cmd.c:378-484); no_fr = trim palindromes (default 1) */
int fmdh_build(const char *fa_path, const char *out_path, int device, int max_len, int no_fr);

/* `fermi correct` (cmd.c:253-291, correct.c:305-456); defaults = cmd.c:258 */
typedef struct { int w, min_occ, keep_bad, is_paired, trim_l, step; float max_corr; } fmdh_ecopt_t; /* = fmecopt_t, fermi.h:26-29 */
int fmdh_correct(const char *fmd_path, const char *fq_path, int device, fmdh_ecopt_t *opt, FILE *out);
#define FMDH_MAX_GPUS 16
/* `-g a,b,..`: the harvest sharded by the last base of the k-mer (whole trees of the trie, at most four GPUs), the table replicated,
 * every batch of reads split over the GPUs; the output bytes are those of one GPU */
int fmdh_correct_multi(const char *fmd_path, const char *fq_path, int n_dev, const int *devices, fmdh_ecopt_t *opt, FILE *out);
int fmdh_correct_reads_multi(const fmdh_ecopt_t *opt, int n_dev, const int *devices, int suf_len, uint64_t n, const uint32_t *bucket, const uint32_t *key,
                             const uint8_t *val, const char *fq_path, FILE *out);
/* `exact -g a,b,..`: the index replicated, every batch of queries split over the GPUs, output in input order */
int fmdh_exact_multi(const char *fmd_path, const char *fa_path, int n_dev, const int *devices, int self_match, FILE *out);
void fmdh_correct_set_threads(int n);                                        /* `-t`: host threads of the marking pass (output independent of n) */
int fmdh_correct_kmer(uint64_t n_symbols);                                   /* automatic k, correct.c:313-318 */
/* phase 2 only (ec_fix on the GPU, fmd_ecfix_batch; marking, filtering and printing on the host) against an already
 * harvested (bucket, key, val) table */
int fmdh_correct_reads(const fmdh_ecopt_t *opt, int device, int suf_len, uint64_t n, const uint32_t *bucket, const uint32_t *key,
                       const uint8_t *val, const char *fq_path, FILE *out);

#ifdef __cplusplus
}
#endif

/* Large tables that are filled once and then read at random (overlap table, its links): fmd_table_alloc (include/fmd_hip.h) --
 * huge pages where the kernel grants them on request (`unitig` on 10 M reads: 8.2 -> 6.4 s), file pages under FMD_TABLE_DIR.
 * Contents are undefined; release with fmdh_big_free(). */
#define fmdh_big_alloc(bytes) fmd_table_alloc(bytes)
#define fmdh_big_free(p) fmd_table_free(p)

#endif
