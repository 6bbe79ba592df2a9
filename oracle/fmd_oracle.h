/* oracle/fmd_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A plain-C, single-threaded CPU restatement of the FMD-index hot path of lh3/fermi, written
 * from the algorithm description and checked against the compiled
 * reference (oracle/_ref, built from /root/reference in place) and against the committed
 * golden vectors in tests/golden/ (see tests/golden/make_golden.py).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may call into this
 * library.  The product path (fermi_amd/, include/) never links or loads it.
 *
 * Every function cites the reference file:line whose behaviour it restates.
 */
#ifndef FMD_ORACLE_H
#define FMD_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Bi-interval, same field meaning as fmintv_t (fermi.h:13-16). */
typedef struct {
    uint64_t x[3];  /* [0] SA start of W, [1] SA start of revcomp(W), [2] size */
    uint64_t info;
} orc_intv_t;

typedef struct { size_t n, m; orc_intv_t *a; } orc_intv_v;

/* The RLD container held flat (rld.h:20-39 keeps it in 2^23-word chunks; same bytes). */
typedef struct {
    uint64_t n_words;    /* payload length in 64-bit words (n_bytes / 8)            */
    uint64_t *w;         /* payload, little-endian words                            */
    uint64_t n_frames;   /* rank frames, 7 words per row (rld.c:186-224)            */
    uint64_t *frame;
    int ibits;           /* recomputed on load (rld.c:322-323)                      */
    uint64_t cnt[7];     /* cnt[c] = #symbols < c ; cnt[6] = total (rld.c:282-284)  */
    uint64_t mcnt[7];    /* mcnt[0] = total, mcnt[1..6] = marginal $ACGTN           */
} orc_rld_t;

/* instrumentation for bench.py's algorithmic-bytes accounting: per-thread
 * counters of rank calls; orc_counters_read() returns and clears the calling thread's. */
typedef struct { uint64_t rank1a, rank2a, rank2a_spill; } orc_counters_t;
orc_counters_t orc_counters_read(void);
void orc_counters_flush(void);

/* ---- container (rld.c:265-346) ---- */
orc_rld_t *orc_rld_load(const char *fn);          /* accepts "RLD\2" and raw "RLE\6" streams */
orc_rld_t *orc_rld_from_bwt(const uint8_t *bwt, uint64_t n); /* encode a plain nt6 BWT string */
void orc_rld_free(orc_rld_t *e);
int  orc_rld_dump(const orc_rld_t *e, const char *fn);        /* rld.c:242-263 */
/* sequential decode of the whole BWT into bytes (chkbwt -p, cmd.c:106); returns length */
uint64_t orc_rld_decode_all(const orc_rld_t *e, uint8_t *out);

/* ---- rank (rld.c:352-492) ---- */
int  orc_rank1a(const orc_rld_t *e, uint64_t k, uint64_t ok[6]);
void orc_rank2a(const orc_rld_t *e, uint64_t k, uint64_t l, uint64_t ok[6], uint64_t ol[6]);

/* ---- FMD operations (exact.c) ---- */
void orc_set_intv(const orc_rld_t *e, int c, orc_intv_t *ik);                     /* fermi.h:53 */
void orc_extend(const orc_rld_t *e, const orc_intv_t *ik, orc_intv_t ok[6], int is_back); /* exact.c:72 */
void orc_extend0(const orc_rld_t *e, const orc_intv_t *ik, orc_intv_t *ok0, int is_back); /* exact.c:90 */
uint64_t orc_backward_search(const orc_rld_t *e, int len, const uint8_t *str,
                             uint64_t *sa_beg, uint64_t *sa_end);                  /* exact.c:7 */
/* exact.c:59: writes the x-th sequence REVERSED into s (capacity cap), *len = its length,
 * returns the rank of the sequence among sentinels */
int64_t orc_retrieve(const orc_rld_t *e, uint64_t x, uint8_t *s, int cap, int *len);
orc_intv_t *orc_traverse(const orc_rld_t *e, int depth);                                 /* exact.c:141 */

/* ---- SMEM (smem.c:13-80, 104, 397) ---- */
int orc_smem1(const orc_rld_t *e, int len, const uint8_t *q, int x, orc_intv_v *mem, int self_match);
int orc_smem(const orc_rld_t *e, int len, const uint8_t *q, orc_intv_v *mem, int self_match);
int orc_ec_range(const orc_rld_t *e, int w, int min_occ, int suf_len, int b0, int b1, int n_threads, uint32_t **o_bucket,
                 uint32_t **o_key, uint8_t **o_val, uint64_t *o_n, double *secs);
void orc_smem_batch(const orc_rld_t *e, size_t n, int len, const uint8_t *seqs, int self_match, uint32_t max_mem,
                    orc_intv_t *mem, uint32_t *n_mem, int n_threads);

/* ---- overlap discovery (unitig.c:38-179), used == NULL / sorted == NULL form ---- */
typedef struct { size_t n, m; uint8_t *s; } orc_str_t;
int orc_is_contained(const orc_rld_t *e, int min_match, const uint8_t *s, int len,
                     orc_intv_t *intv, orc_intv_v *ovlp);                          /* unitig.c:77 */
/* unitig.c:93; s grows (consensus bases appended); returns rbeg or -1 */
int orc_get_nei(const orc_rld_t *e, int min_match, int beg, orc_str_t *s, orc_intv_v *nei,
                orc_intv_v *prev, orc_intv_v *curr);

int orc_check_left_simple(const orc_rld_t *e, int min_match, int beg, int rbeg, const uint8_t *s, int len);
unsigned orc_left_fork(const orc_rld_t *e, int min_match, const uint8_t *x, int len); /* unitig.c:186 */

/* ---- k-mer harvest for `correct` (correct.c:35-87): appends (key,val) pairs ---- */
typedef struct { size_t n, m; uint32_t *key; uint8_t *val; int64_t cnt[2]; } orc_solid_t;
void orc_ec_collect(const orc_rld_t *e, int w, int min_occ, int suf_len, const orc_intv_t *suf_intv,
                    orc_solid_t *out);

/* ---- helpers (seq.c:12-56) ---- */
extern const unsigned char orc_nt6_table[128];
void orc_revcomp6(int l, uint8_t *s);
void orc_reverse(int l, uint8_t *s);

/* ---- batch drivers used by tests / cpu_baseline (pthread start/step interleave like
 *      unitig.c:394-404) ---- */
void orc_backward_search_batch(const orc_rld_t *e, size_t n, int len, const uint8_t *seqs,
                               uint64_t *cnt, uint64_t *beg, uint64_t *end, int n_threads);

#ifdef __cplusplus
}
#endif
#endif
