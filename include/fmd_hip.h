/* include/fmd_hip.h -- C ABI of libfmdhip.so, the MI355X (gfx950) implementation of fermi's
 * FMD-index hot path.  Plain pointers and sizes only; no C++ or torch types.
 *
 * This is the drop-in boundary: each entry point is the batched, device-side
 * replacement of one function of the reference's internal C API (rld.h:45-58, fermi.h:61-103,
 * unitig.c:77/93, correct.c:35), cited per declaration.  A reference maintainer binds these
 * from cmd.c / unitig.c / correct.c exactly as shown in INTEGRATION.md.
 *
 * Conventions
 *   - every function returns 0 (FMD_OK) or a negative FMD_E_* code; fmd_strerror() names it.
 *   - fmd_dev_t is an opaque handle to an index resident in one GPU's HBM (read-only after
 *     open, like the reference's `const rld_t*`; safe to share between host threads).
 *   - `*_dev` entry points take DEVICE pointers and a hipStream_t (as void*; NULL = default
 *     stream), enqueue work and return without synchronising.  They allocate nothing.
 *   - `*_batch` entry points take HOST pointers: copy in, run the `_dev` path, copy out, sync.
 *   - nt6 alphabet everywhere: $=0 A=1 C=2 G=3 T=4 N=5 (seq.c:12-21).
 *   - There is NO CPU fallback behind these symbols: without a GPU they return FMD_E_NODEV.
 */
#ifndef FMD_HIP_H
#define FMD_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FMD_OK          0
#define FMD_E_NODEV    (-1)  /* no usable HIP device */
#define FMD_E_ARG      (-2)  /* bad argument */
#define FMD_E_FORMAT   (-3)  /* not an RLD\2 (asize 6, sbits 3) or RLE\6 file */
#define FMD_E_IO       (-4)  /* file could not be read/written */
#define FMD_E_NOMEM    (-5)  /* host or device allocation failed */
#define FMD_E_HIP      (-6)  /* a HIP runtime call failed; see fmd_last_hip_error() */
#define FMD_E_OVERFLOW (-7)  /* a fixed-capacity device list overflowed for some item */

/* bi-interval; identical layout to fmintv_t (fermi.h:13-16) */
typedef struct {
    uint64_t x[3]; /* [0] SA start of W, [1] SA start of revcomp(W), [2] size */
    uint64_t info;
} fmd_intv_t;

typedef struct fmd_dev fmd_dev_t;

typedef struct {
    uint64_t cnt[7];     /* cnt[c] = # symbols < c, cnt[6] = total      (rld.c:282-284) */
    uint64_t mcnt[7];    /* mcnt[0] = total, mcnt[1..6] = # of $ACGTN   (rld.c:233)     */
    uint64_t n_blocks;   /* 128-byte device blocks                                      */
    uint64_t hbm_bytes;  /* device bytes held by the index                              */
    int device;
} fmd_info_t;

const char *fmd_strerror(int code);
const char *fmd_last_hip_error(void);
int fmd_device_count(void);

/* ---- index residency: replaces rld_restore / rld_restore_mmap / rld_destroy -------------
 * (rld.c:288, :327, :81).  The on-disk .fmd is unchanged; the GPU transcodes it at upload
 * into fixed-stride 128-byte rank blocks (DESIGN.md "HBM layout"). */
int fmd_dev_open_file(int device, const char *fn, fmd_dev_t **out);            /* RLD\2 or RLE\6 */
int fmd_dev_open_rld(int device, const uint64_t *payload, uint64_t n_words,
                     const uint64_t mcnt[7], fmd_dev_t **out);                  /* RLD\2 payload words (rld.c:242-263) */
int fmd_dev_open_rle6(int device, const uint8_t *runs, uint64_t n_bytes, fmd_dev_t **out); /* len<<3|sym bytes (ropebwt.c:132) */
int fmd_dev_open_bwt(int device, const uint8_t *bwt, uint64_t n, fmd_dev_t **out);          /* plain nt6 BWT string, host */
int fmd_dev_open_bwt_dev(int device, const uint8_t *d_bwt, uint64_t n, fmd_dev_t **out);    /* same, already in HBM */
void fmd_dev_close(fmd_dev_t *h);
/* the work areas the handle keeps between calls of the host-buffer entries (fmd_*_batch, the table jobs) back to the device; -> bytes released.
 * (The reference's per-call vectors are freed per call, e.g. unitig.c:321-325; the handle keeps them because allocating 10-100 GB per call costs more than the call.) */
uint64_t fmd_dev_trim(fmd_dev_t *h);
int fmd_dev_info(const fmd_dev_t *h, fmd_info_t *info);
int fmd_dev_sync(const fmd_dev_t *h, void *stream);
/* measurement aid (no reference counterpart): lines[0] = 64-byte rank blocks, lines[1] = other random lines
 * (prefix-table look-ups) that the kernels launched on this handle have requested since the last reset.  Only
 * libfmdhip_count.so -- the same sources built with -DFMD_COUNT_LINES=1 -- counts (*counting = 1); the shipped
 * library returns zeros and *counting = 0.  Synchronises the device. */
int fmd_dev_line_count(fmd_dev_t *h, uint64_t lines[2], int reset, int *counting);
int fmd_dev_line_count3(fmd_dev_t *h, uint64_t lines[3], int reset, int *counting);   /* ... and lines[2] = 128-byte two-base blocks requested (fmd_dev_build_pairs) */

/* ---- rank: rld_rank1a (rld.c:424) / rld_rank2a (rld.c:457) -------------------------------
 * ok/ol: n rows of 6 counts ($ACGTN) of BWT[0..k] inclusive; k == UINT64_MAX gives zeros.
 * sym (may be NULL): BWT[k], -1 for k == UINT64_MAX. */
int fmd_rank1a_dev(fmd_dev_t *h, void *stream, size_t n, const uint64_t *d_k, uint64_t *d_ok, int8_t *d_sym);
int fmd_rank2a_dev(fmd_dev_t *h, void *stream, size_t n, const uint64_t *d_k, const uint64_t *d_l,
                   uint64_t *d_ok, uint64_t *d_ol);
int fmd_rank1a_batch(fmd_dev_t *h, size_t n, const uint64_t *k, uint64_t *ok, int8_t *sym);
int fmd_rank2a_batch(fmd_dev_t *h, size_t n, const uint64_t *k, const uint64_t *l, uint64_t *ok, uint64_t *ol);

/* ---- fm6_extend (exact.c:72-88): n bi-intervals -> n x 6 bi-intervals (info = 0) --------- */
int fmd_extend_dev(fmd_dev_t *h, void *stream, size_t n, const fmd_intv_t *d_ik, const uint8_t *d_is_back,
                   fmd_intv_t *d_ok);
int fmd_extend_batch(fmd_dev_t *h, size_t n, const fmd_intv_t *ik, const uint8_t *is_back, fmd_intv_t *ok);

/* ---- fm_backward_search (exact.c:7-23) ---------------------------------------------------
 * Read i is seqs[off[i] .. off[i+1]) in nt6.  cnt[i] = occurrences (0 = miss; beg/end are then
 * written as 0, where the reference leaves them untouched). */
int fmd_bsearch_dev(fmd_dev_t *h, void *stream, size_t n, const uint8_t *d_seqs, const uint64_t *d_off,
                    uint64_t *d_cnt, uint64_t *d_beg, uint64_t *d_end);
int fmd_bsearch_batch(fmd_dev_t *h, size_t n, const uint8_t *seqs, const uint64_t *off,
                      uint64_t *cnt, uint64_t *beg, uint64_t *end);

/* ---- fm_retrieve (exact.c:59-70): LF-walk from row x[i] until '$' --------------------------
 * Row i of seqs (stride bytes) receives the sequence REVERSED, exactly as fm_retrieve emits it
 * (callers seq_reverse it, unitig.c:285); len[i] = its length (bases beyond `stride` are
 * dropped, len still counts them); rank[i] = the returned sentinel rank. */
int fmd_retrieve_dev(fmd_dev_t *h, void *stream, size_t n, const uint64_t *d_x, uint8_t *d_seqs, uint32_t stride,
                     uint32_t *d_len, uint64_t *d_rank);
int fmd_retrieve_batch(fmd_dev_t *h, size_t n, const uint64_t *x, uint8_t *seqs, uint32_t stride,
                       uint32_t *len, uint64_t *rank);

/* ---- inspection (`fermi chkbwt`, cmd.c:47-130) ------------------------------------------------
 * export: BWT[first, first+n) as nt6 bytes, decoded from the device layout (chkbwt -p).
 * check_rank: the device layout's rank of every position against the symbols themselves (chkbwt -r):
 * n_bad = positions whose six counts are not those of the previous position plus the symbol there (or,
 * at the last position, not the marginal counts); first_bad = the smallest of them. */
int fmd_dev_export_bwt(fmd_dev_t *h, uint64_t first, uint64_t n, uint8_t *bwt);
int fmd_dev_check_rank(fmd_dev_t *h, uint64_t *n_bad, uint64_t *first_bad);
/* The two-base blocks (16 bits per symbol beside the index; fmd_pair.hip): what the sorted overlap job reads below min_match, two bases per
 * 128-byte line where the rank blocks give one per 64-byte line -- same results, fewer requests.  The job builds them on first use where they fit
 * (FMD_PAIR=0: never, FMD_PAIR=1: whenever the allocation succeeds); this call builds them now (*built = 1 when the handle has them).  The
 * reference has nothing like it: rld_rank2a (rld.c:457) is asked once per base (unitig.c:47-59, exact.c:59-70).
 * fmd_dev_check_pairs: every row's pair step against two single LF steps (FMD_E_ARG when the handle has no two-base blocks). */
int fmd_dev_build_pairs(fmd_dev_t *h, int *built);
int fmd_dev_check_pairs(fmd_dev_t *h, uint64_t *n_bad, uint64_t *first_bad);

/* ---- forward reach ("matching statistics") --------------------------------------------------
 * seqs: n_bytes of nt6 sequences, each followed by at least one 0 byte (the _dev form: 4-byte
 * aligned, a 0 byte at or after seqs[n_bytes - 1], readable to the next multiple of 4).  len[p] = length of
 * the longest prefix of seqs[p ..] (up to its terminator) that occurs in the index; 0 at terminators and
 * at bases the index does not contain.  x -> x + len[x] is the chain of start positions that
 * fm6_miter_next / fm6_smem walk (smem.c:46, :96-102, :404-409): with it, every fm6_smem1_core call
 * of a long sequence is known up front and independent (fmd_smem_win_* with stop = start + 1). */
int fmd_reach_dev(fmd_dev_t *h, void *stream, size_t n_bytes, const uint8_t *d_seqs, uint32_t *d_len);
int fmd_reach_batch(fmd_dev_t *h, size_t n_bytes, const uint8_t *seqs, uint32_t *len);

/* ---- super-maximal exact matches: fm6_smem (smem.c:397-410) = repeated fm6_smem1_core
 * (smem.c:13-80); what `fermi exact [-s]` prints (cmd.c:319-327, smem.c:412-418).
 * mem: n rows of max_mem intervals in the reference's order; info = leftclosed<<63 | beg<<32 | end
 * (FM_MASK30 fields, smem.c:63).  n_mem[i] = number of SMEMs of read i; bit 31 set = the read was
 * longer than max_len or produced more than max_mem SMEMs (row invalid: re-run larger).
 * d_seqs: 4-byte aligned and readable up to the next multiple of 4 past off[n] (the kernels read
 * the sequences one aligned word at a time); same for fmd_bsearch_dev. */
size_t fmd_smem_work_bytes(size_t n, uint32_t max_len);
int fmd_smem_dev(fmd_dev_t *h, void *stream, size_t n, const uint8_t *d_seqs, const uint64_t *d_off, int self_match,
                 uint32_t max_len, uint32_t max_mem, fmd_intv_t *d_mem, uint32_t *d_n_mem, void *d_work, size_t work_bytes);
/* The same chain over WINDOWS of long sequences (what fm6_miter_next does over a contig in `fermi
 * remap`, smem.c:96-102, :151): item i = start positions [start, stop) of the sequence of seq_len
 * bases at seqs + seq_off.  The union over a partition of a sequence into windows is its SMEM set;
 * an SMEM covering a window boundary may be reported by both windows.  max_len = longest match
 * possible (longest sequence in the index + 1). */
typedef struct { uint64_t seq_off; uint32_t seq_len, start, stop, reserved /* flags */; } fmd_smem_win_t;
#define FMD_SMEM_WIN_F_FULL 1u   /* keep only matches closed by a sentinel on both sides (what remap consumes) */
int fmd_smem_win_dev(fmd_dev_t *h, void *stream, size_t n, const uint8_t *d_seqs, const fmd_smem_win_t *d_wins, int self_match,
                     uint32_t max_len, uint32_t max_mem, fmd_intv_t *d_mem, uint32_t *d_n_mem, void *d_work, size_t work_bytes);
int fmd_smem_win_batch(fmd_dev_t *h, size_t n, const uint8_t *seqs, uint64_t seq_bytes, const fmd_smem_win_t *wins, int self_match,
                       uint32_t max_len, uint32_t max_mem, fmd_intv_t *mem, uint32_t *n_mem);
int fmd_smem_batch(fmd_dev_t *h, size_t n, const uint8_t *seqs, const uint64_t *off, int self_match, uint32_t max_len,
                   uint32_t max_mem, fmd_intv_t *mem, uint32_t *n_mem);

/* ---- overlap discovery for unitig construction --------------------------------------------
 * One record per sequence id: the read-only front half of unitig1 (unitig.c:274-300), i.e.
 *   fm_retrieve (exact.c:59) + seq_reverse + fm6_is_contained (unitig.c:77) +
 *   fm6_get_nei(beg = 0, used = NULL, sorted = NULL) (unitig.c:93)
 * which SURVEY.md (fact 3) shows is a pure function of (index, id) and is exactly what the
 * deterministic `fermi unitig -t1` walk consumes.  `rank`, k[0] and k[1] are in the coordinate
 * system of the reference's used/bend/visited bitmaps (unitig.c:289, 299). */
#define FMD_OVLP_SHORT      (-1)  /* len <= min_match (unitig.c:288) */
#define FMD_OVLP_CONTAINED  (-3)  /* fm6_is_contained < 0 (unitig.c:292) */
#define FMD_OVLP_F_FORKED   1u    /* more than one category survived a round (unitig.c:152) */
#define FMD_OVLP_F_OVERFLOW 2u    /* a capacity (max_len, list, max_nei) was exceeded: record invalid, re-run larger */
#define FMD_OVLP_F_FIXED    4u    /* the fake-fork fix-up of unitig.c:158-176 ran */
#define FMD_OVLP_F_PACK4    8u    /* packed rows only (fmd_ovlp_pack_dev): the bases are stored 2 per byte, not 4 */
typedef struct {
    uint64_t rank;     /* fm_retrieve's return value */
    uint64_t k[3];     /* *intv of fm6_is_contained: bi-interval of `$read$` */
    int32_t len;       /* sequence length */
    int32_t status;    /* 0, FMD_OVLP_SHORT or FMD_OVLP_CONTAINED */
    int32_t n_ovlp;    /* candidate intervals from overlap_intv (unitig.c:47-58) */
    int32_t rbeg;      /* fm6_get_nei's return value; -1 = no overlap */
    int32_t ext_len;   /* bases fm6_get_nei appended to the sequence */
    int32_t n_nei;     /* irreducible neighbours found */
    uint32_t flags;    /* FMD_OVLP_F_* */
    uint16_t reserved; /* check_left_simple (unitig.c:186) for the edge to the unique neighbour, when it was computed
                          for this row (fmd_ovlp_check_left_dev): 0 = passes, 1 = potential backward bifurcation (-1);
                          2 = not computed / not applicable */
    uint16_t lfork;    /* what fm6_get_nei's rounds on THIS strand X say about check_left_simple on any edge S -> N whose
                          neighbour N is the reverse complement of X (FMD_LFORK_*): the reads check_left_simple collects on
                          N and pulls back over S are the candidates of X extended forward, round for round */
} fmd_ovlp_rec_t;      /* 64 bytes */
/* lfork = D << 15 | R:  rounds 0 .. R-1 saw every read that starts inside X with >= min_match bases either end or go on
 * with one and the same base (R = FMD_LFORK_ALL: all of them ended, nothing can ever disagree); D = 1: round R has two
 * different bases.  For an edge S -> N with rbeg = fm6_get_nei's return value on S (the walk's check_left_simple(beg = 0,
 * rbeg), unitig.c:186-204 visits rounds 0 .. rbeg-1):   rbeg <= R  => 0;   D && R < rbeg  => -1;   otherwise not
 * decided by this row (lfork = 0 decides nothing) and fmd_ovlp_check_left_dev has to be run on S. */
#define FMD_LFORK_ALL 0x7fffu
static inline int fmd_lfork_decide(uint16_t lfork, int rbeg) /* 0 / -1 as check_left_simple, 1 = undecided */
{
    const int r = lfork & 0x7fff;
    if (rbeg <= r || r == (int)FMD_LFORK_ALL) return 0;
    return (lfork & 0x8000) ? -1 : 1;
}
/* capacity of the per-strand candidate lists kept in the work area: overlap_intv (unitig.c:47-58) pushes at most one
 * interval per suffix longer than min_match, and a round of fm6_get_nei keeps more children than it had parents only
 * where the read set forks -- a strand that needs more sets FMD_OVLP_F_OVERFLOW and is re-run with larger capacities */
static inline uint32_t fmd_ovlp_list_cap(uint32_t max_len, int min_match)
{
    uint32_t d = max_len > (uint32_t)min_match ? max_len - (uint32_t)min_match : 1;
    return d + 8 < 16 ? 16 : d + 8;
}
size_t fmd_ovlp_work_bytes(size_t n, uint32_t max_len, int min_match);
/* d_nei: n x max_nei neighbours {x[0], x[1], x[2] of `$neighbour$`, info = overlap length};
 * d_seq: n rows of seq_stride bytes = the sequence in read order followed by the ext_len appended
 * bases (seq_stride >= 2*max_len is always enough).
 * Stream order is the caller's: the call starts after the work already queued on `stream` and its results
 * belong to `stream` when it returns, although a batch of 2^21 strands or more also runs kernels on a
 * second, library-owned stream in between (joined by events, no host synchronisation). */
int fmd_ovlp_dev(fmd_dev_t *h, void *stream, size_t n, const uint64_t *d_ids, int min_match, uint32_t max_len,
                 uint32_t max_nei, fmd_ovlp_rec_t *d_rec, fmd_intv_t *d_nei, uint8_t *d_seq, uint32_t seq_stride,
                 void *d_work, size_t work_bytes);
/* The same records for ALL n ids in one call, computed in an order that keeps neighbours on the genome in flight together
 * (fm6_unitig hands its workers the ids in input order, unitig.c:394-404; any order gives the same records, and the order decides
 * how often a rank block is found in cache).  Two strands whose last bases lie d positions apart on the genome visit the same rank
 * blocks d steps apart; in input order each such visit is a DRAM miss.  Pass 1 takes every strand 32 bases in and parks it (64 bytes
 * per strand); the strands are sorted by the minimizer of those 32 bases; pass 2 + fm6_get_nei then run batch by batch in that
 * order and write row i of d_rec / d_nei / d_seq for ids[i] with the contents fmd_ovlp_dev gives them: the record in full, the first
 * min(n_nei, max_nei) neighbours, the first len + ext_len bytes of the sequence row (bytes of a row beyond those are unspecified in both).  `batch` strands share the work
 * area of one fmd_ovlp_dev call (0 = n); work_bytes >= fmd_ovlp_sorted_work_bytes(n, batch, ..).  Where the two-pass form does not
 * apply (min_match < 32, FMD_OVLP_SORT=0) the batches are taken in id order. */
size_t fmd_ovlp_sorted_work_bytes(size_t n, size_t batch, uint32_t max_len, int min_match);
int fmd_ovlp_sorted_dev(fmd_dev_t *h, void *stream, size_t n, const uint64_t *d_ids, int min_match, uint32_t max_len,
                        uint32_t max_nei, fmd_ovlp_rec_t *d_rec, fmd_intv_t *d_nei, uint8_t *d_seq, uint32_t seq_stride,
                        void *d_work, size_t work_bytes, size_t batch);
/* Rows that exceeded a capacity, again, alone, larger.  fm6_get_nei has no capacities (its vectors grow, unitig.c:93-179); a row of the calls above
 * that needs more than max_nei neighbours, a longer candidate list or more bases than max_len carries FMD_OVLP_F_OVERFLOW instead of a
 * result.  This entry collects the flagged rows of a finished job on the device (d_side_ids[k] = their ids -- d_ids[i], or i when d_ids is
 * NULL --, d_side_rows[k] = their rows, any order; d_side_rows may be NULL) and runs them through fmd_ovlp_dev with the capacities named
 * here into the side arrays (row k of d_side_rec / d_side_nei [max_nei per row] / d_side_seq [side_stride per row]).  *n_side = rows
 * collected (FMD_E_OVERFLOW when they exceed side_cap: nothing was run), *n_still = rows of the side table that are flagged again
 * (call again on the side table with larger capacities).  Synchronises the stream twice (the counts come back to the host). */
size_t fmd_ovlp_side_work_bytes(size_t side_cap, uint32_t max_len, int min_match);
int fmd_ovlp_rerun_overflow_dev(fmd_dev_t *h, void *stream, size_t n, const uint64_t *d_ids, const fmd_ovlp_rec_t *d_rec, int min_match, uint32_t max_len,
                                uint32_t max_nei, uint64_t side_cap, uint64_t *d_side_ids, uint32_t *d_side_rows, fmd_ovlp_rec_t *d_side_rec, fmd_intv_t *d_side_nei,
                                uint8_t *d_side_seq, uint32_t side_stride, void *d_work, size_t work_bytes, uint64_t *n_side, uint64_t *n_still);
/* Optional second step over the same buffers: check_left_simple (unitig.c:186-204) for every
 * strand with a unique neighbour -> rec.reserved (the unitig walk needs it; plain overlap
 * discovery does not, and records of fmd_ovlp_dev alone carry reserved = 2). */
int fmd_ovlp_check_left_dev(fmd_dev_t *h, void *stream, size_t n, int min_match, uint32_t max_len, fmd_ovlp_rec_t *d_rec,
                            const uint8_t *d_seq, uint32_t seq_stride, void *d_work, size_t work_bytes);
int fmd_ovlp_batch(fmd_dev_t *h, size_t n, const uint64_t *ids, int min_match, uint32_t max_len, uint32_t max_nei,
                   fmd_ovlp_rec_t *rec, fmd_intv_t *nei, uint8_t *seq, uint32_t seq_stride, int with_check_left);

/* ---- compact form of a finished batch: what leaves the GPU (PCIe to the host walk, xGMI to rank 0) ----------
 * The walk (unitig.c:227-317) reads from a row its record, its n_nei neighbours and len + ext_len bases.  Row i becomes
 *   d_prec[i]          the record (flags gains FMD_OVLP_F_PACK4 when the row holds a base other than A/C/G/T)
 *   d_var + d_off[i]   min(n_nei, max_nei) neighbours (32 bytes each), then len + ext_len bases, 4 per byte as
 *                      (base - 1) in 2 bits, first base in the low bits -- or nt6 codes 2 per byte with PACK4 --
 *                      padded to 8 bytes; empty for rows with status != 0 or FMD_OVLP_F_OVERFLOW
 * d_off has n + 1 entries; rows that would end past var_cap are not written (compare d_off[n] with var_cap;
 * fmd_ovlp_pack_max_bytes is always enough).  seq_stride must be a multiple of 8 (FMD_E_ARG otherwise). */
size_t fmd_ovlp_pack_max_bytes(size_t n, uint32_t max_nei, uint32_t seq_stride);
size_t fmd_ovlp_pack_work_bytes(size_t n);
int fmd_ovlp_pack_dev(fmd_dev_t *h, void *stream, size_t n, const fmd_ovlp_rec_t *d_rec, const fmd_intv_t *d_nei, uint32_t max_nei,
                      const uint8_t *d_seq, uint32_t seq_stride, fmd_ovlp_rec_t *d_prec, uint64_t *d_off, uint8_t *d_var, uint64_t var_cap,
                      void *d_work, size_t work_bytes);

/* The link pass over a COMPLETE table on one device (rows = sequence ids 0 .. n-1, fixed-stride records): row_of[k] = the
 * smallest id whose `$read$` interval starts at k; link[i] = rows of the unique neighbour of i and of its reverse strand
 * (what the walk's `cur = neighbour` and check_left's second look need, unitig.c:206-262); rec[i].reserved = 0 / 1 where
 * the lfork of the neighbour's reverse strand decides check_left_simple, left at 2 elsewhere -- those ids are appended
 * to d_undecided (any order; n capacity) for fmd_ovlp_check_left_dev, which only looks at rows still at 2.
 * d_nei_x01: x[0], x[1] of each row's first neighbour at d_nei_x01[i * nei_stride_u64] (a fmd_intv_t array: stride
 * 4 * max_nei; a compact copy: 2). */
typedef struct { uint32_t nxt, rev; } fmd_ovlp_link_t;
int fmd_ovlp_link_dev(fmd_dev_t *h, void *stream, size_t n, fmd_ovlp_rec_t *d_rec, const uint64_t *d_nei_x01, uint32_t nei_stride_u64,
                      uint32_t *d_row_of, fmd_ovlp_link_t *d_link, uint64_t *d_undecided, uint64_t *d_n_undecided);

/* Host form, pipelined: the packed table of one shard of sequence ids -- ids[0..n) when ids != NULL, else
 * first, first + step, ... (the reference's worker interleave, unitig.c:333, 398-399).  Chunks of 2^chunk_shift rows
 * are computed (fmd_ovlp_dev, fmd_ovlp_check_left_dev when asked, fmd_ovlp_pack_dev) while the previous chunk crosses
 * PCIe.  rec[n], off[n] (byte offset of row i inside chunks[i >> chunk_shift]); chunks[] receives
 * ceil(n / 2^chunk_shift) buffers the caller releases with fmd_ovlp_packed_free().  The end of row i's variable part
 * follows from its record (fmd_ovlp_row_bytes). */
int fmd_ovlp_packed_batch(fmd_dev_t *h, const uint64_t *ids, uint64_t first, uint64_t step, size_t n, int min_match, uint32_t max_len,
                          uint32_t max_nei, int with_check_left, fmd_ovlp_rec_t *rec, uint64_t *off, uint32_t chunk_shift, uint8_t **chunks);
void fmd_ovlp_packed_free(uint8_t **chunks, size_t n_chunks);
/* Host memory for tables that are filled once and then read at random -- the chunks above come from here, and a caller that keeps
 * rec / off / row_of / link for a whole .fmd (the reference keeps nothing of the kind: it recomputes, unitig.c:274-300) should too.
 * Default: malloc; large blocks 2 MiB-aligned with transparent huge pages on request.  With FMD_TABLE_DIR=<dir> in the environment,
 * blocks of FMD_TABLE_DIR_MIN bytes (default 32 MiB) and more are pages of unlinked files in <dir> (mmap, MAP_SHARED): the table of a
 * large .fmd then lives in the page cache and on <dir>'s device instead of in anonymous memory, and `unitig` of BASELINE's 7*10^8
 * reads (~ 190 GB of table) runs in whatever RAM the host has, at the speed of that device once the table no longer fits.
 * fmd_table_free() accepts any pointer fmd_table_alloc() returned (and, like free(), NULL). */
void *fmd_table_alloc(size_t bytes);
void fmd_table_free(void *p);
/* The whole table (ids 0 .. n-1) on ONE device: the packed rows as above (no per-row check_left), then fmd_ovlp_link_dev on
 * the device: row_of[n], link[n], rec[i].reserved = check_left_simple's verdict wherever lfork decides it; the ids it leaves
 * open come back in *undecided (malloc'ed: fmd_host_free; ascending) for a fmd_ovlp_packed_batch(ids, with_check_left = 1). */
int fmd_ovlp_packed_table(fmd_dev_t *h, size_t n, int min_match, uint32_t max_len, uint32_t max_nei, fmd_ovlp_rec_t *rec, uint64_t *off,
                          uint32_t chunk_shift, uint8_t **chunks, uint32_t *row_of, fmd_ovlp_link_t *link, uint64_t **undecided, uint64_t *n_undecided);
/* Streamed forms of the two entries above: the packed chunks (2^chunk_shift rows each, in the order of the ids) are handed to fn from
 * pinned staging buffers -- first_row = the chunk's first row of the call, rec[n_rows], off[n_rows] (offsets into var), var_bytes of var --
 * and are the library's again when fn returns (non-zero: the call stops and returns it).  fn runs on the calling thread while the next
 * chunk is computed and the one after crosses PCIe.  Nothing of the table stays in host memory unless fn keeps it: `unitig` keeps
 * ~58 bytes per row (host/slim_table.c) where the reference keeps three bitmaps (unitig.c:390-392) and recomputes. */
typedef int (*fmd_ovlp_rows_fn)(void *ctx, uint64_t first_row, size_t n_rows, const fmd_ovlp_rec_t *rec, const uint64_t *off, const uint8_t *var, uint64_t var_bytes);
int fmd_ovlp_packed_stream(fmd_dev_t *h, const uint64_t *ids, uint64_t first, uint64_t step, size_t n, int min_match, uint32_t max_len,
                           uint32_t max_nei, int with_check_left, uint32_t chunk_shift, fmd_ovlp_rows_fn fn, void *ctx);
/* fmd_ovlp_packed_table as a job in three steps, so that the link pass sees the rows as they are in the END:
 *   rows   ids 0 .. n-1 streamed to fn; the job keeps the fixed-stride records and the first neighbour's coordinates on the device
 *   patch  rows the caller has computed again since (flagged FMD_OVLP_F_OVERFLOW by the first step): rec[m], nei01[2 m] = x[0], x[1] of each
 *          row's first neighbour, replace the job's copies at ids[m]
 *   link   fmd_ovlp_link_dev over the job's records; link[n_rows] and reserved[n_rows] (0 / 1 / 2 as rec.reserved) go to fn in pieces of
 *          2^22 rows; *undecided as fmd_ovlp_packed_table returns it */
typedef struct fmd_ovlp_tabjob fmd_ovlp_tabjob_t;
typedef int (*fmd_ovlp_links_fn)(void *ctx, uint64_t first_row, size_t n_rows, const fmd_ovlp_link_t *link, const uint8_t *reserved);
int fmd_ovlp_tabjob_rows(fmd_dev_t *h, size_t n, int min_match, uint32_t max_len, uint32_t max_nei, uint32_t chunk_shift,
                         fmd_ovlp_rows_fn fn, void *ctx, fmd_ovlp_tabjob_t **job);
int fmd_ovlp_tabjob_patch(fmd_ovlp_tabjob_t *job, size_t m, const uint64_t *ids, const fmd_ovlp_rec_t *rec, const uint64_t *nei01);
int fmd_ovlp_tabjob_link(fmd_ovlp_tabjob_t *job, fmd_ovlp_links_fn fn, void *ctx, uint64_t **undecided, uint64_t *n_undecided);
void fmd_ovlp_tabjob_free(fmd_ovlp_tabjob_t *job);
/* layout of a packed row's variable part, from its (packed) record */
static inline uint32_t fmd_ovlp_row_nei(const fmd_ovlp_rec_t *r, uint32_t max_nei)
{
    if (r->status != 0 || (r->flags & FMD_OVLP_F_OVERFLOW)) return 0;
    return (uint32_t)r->n_nei < max_nei ? (uint32_t)r->n_nei : max_nei;
}
static inline uint32_t fmd_ovlp_row_bytes(const fmd_ovlp_rec_t *r, uint32_t max_nei, uint32_t seq_stride)
{
    uint32_t nb, sb;
    if (r->status != 0 || (r->flags & FMD_OVLP_F_OVERFLOW)) return 0;
    nb = (uint32_t)r->len + (uint32_t)r->ext_len;
    if (nb > seq_stride) nb = seq_stride;
    sb = (r->flags & FMD_OVLP_F_PACK4) ? (nb + 1) / 2 : (nb + 3) / 4;
    return fmd_ovlp_row_nei(r, max_nei) * 32 + ((sb + 7) & ~7u);
}
/* base j of a packed row (nt6 code) */
static inline int fmd_ovlp_row_base(const fmd_ovlp_rec_t *r, uint32_t max_nei, const uint8_t *var, uint32_t j)
{
    const uint8_t *s = var + fmd_ovlp_row_nei(r, max_nei) * 32;
    return (r->flags & FMD_OVLP_F_PACK4) ? (s[j >> 1] >> (4 * (j & 1))) & 15 : ((s[j >> 2] >> (2 * (j & 3))) & 3) + 1;
}

/* ---- the sorted job in its two halves ----------------------------------------------------------------------------
 * fmd_ovlp_sorted_dev = fmd_ovlp_head_dev, then fmd_ovlp_tail_dev over slices of d_order.  A caller that wants to do something
 * between the two -- hand finished rows on while the rest is being computed, exchange the parked strands between GPUs by key
 * (fmd_ovlp_dist_*) -- calls them itself.  fmd_ovlp_two_pass_ok: can this index / these parameters use the two-pass form at all
 * (min_match >= 32, ...)?
 * head: pass 1 of every strand (32 bases in), d_park[n] = 64 bytes per strand, d_keys[n] = the minimizer keys ascending,
 *       d_order[n] = row of the t-th strand in that order; d_rec receives the final record of strands that end inside the head
 *       (shorter than 32 bases: status FMD_OVLP_SHORT; their key is 0xffffffff).
 * tail: pass 2 + fm6_get_nei for np parked strands, slot t of the call = row d_rows[t] of d_park, d_rec, d_nei, d_seq;
 *       work_bytes >= fmd_ovlp_work_bytes(np, ..). */
int fmd_ovlp_two_pass_ok(const fmd_dev_t *h, size_t n, int min_match, uint32_t max_len);
size_t fmd_ovlp_head_work_bytes(size_t n);
int fmd_ovlp_head_dev(fmd_dev_t *h, void *stream, size_t n, const uint64_t *d_ids, int min_match, uint32_t max_len, fmd_ovlp_rec_t *d_rec,
                      void *d_park, uint32_t *d_keys, uint32_t *d_order, void *d_work, size_t work_bytes);
int fmd_ovlp_tail_dev(fmd_dev_t *h, void *stream, size_t np, const uint32_t *d_rows, void *d_park, int min_match, uint32_t max_len, uint32_t max_nei,
                      fmd_ovlp_rec_t *d_rec, fmd_intv_t *d_nei, uint8_t *d_seq, uint32_t seq_stride, void *d_work, size_t work_bytes);
/* fmd_ovlp_pack_dev for the rows d_rows[0..n) of the fixed-stride arrays, in that order ("a piece"): output row t is row d_rows[t];
 * d_pid[t] = its sequence id: d_row_ids[d_rows[t]] when d_row_ids != NULL, else id_first + id_step * d_rows[t]. */
int fmd_ovlp_pack_rows_dev(fmd_dev_t *h, void *stream, size_t n, const uint32_t *d_rows, const uint64_t *d_row_ids, uint64_t id_first, uint64_t id_step,
                           const fmd_ovlp_rec_t *d_rec, const fmd_intv_t *d_nei, uint32_t max_nei, const uint8_t *d_seq, uint32_t seq_stride,
                           uint32_t *d_pid, fmd_ovlp_rec_t *d_prec, uint64_t *d_off, uint8_t *d_var, uint64_t var_cap, void *d_work, size_t work_bytes);

/* ---- N GPUs: the exchange steps of the overlap path behind the C ABI ---------------------------------------------
 * The reference joins N workers over one shared index at no cost (pthreads, unitig.c:394-404); N GPUs hold N replicas of the index
 * and the only data that has to move is (a) the finished rows, to the rank that runs the walk, and (b) -- optional -- the parked
 * strands between pass 1 and pass 2, so that every GPU runs pass 2 on ONE RANGE OF MINIMIZER KEYS of the whole read set instead of on
 * 1/N of the coverage everywhere (strands of one genomic window meet on one GPU and share rank blocks in cache).
 *
 * fmd_comm_t is the transport: two collective calls on DEVICE buffers, enqueued on a HIP stream.  fmd_comm_rccl_* is the one that
 * ships (RCCL over xGMI: ncclAllGather; ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd -- direct, every peer on its own link to
 * the root, no ring); a host may plug in its own (MPI, sockets; the tests plug in torch.distributed/gloo). */
typedef struct { int is_recv, peer; void *d_ptr; size_t bytes; } fmd_comm_op_t;
typedef struct fmd_comm {
    int rank, world;
    void *ctx;
    /* every rank contributes `bytes` from d_send; d_recv receives world x bytes in rank order */
    int (*allgather)(void *ctx, void *stream, const void *d_send, void *d_recv, size_t bytes);
    /* one group of point-to-point transfers (matching sends and receives are posted by the peers in the same call) */
    int (*exchange)(void *ctx, void *stream, int n_ops, const fmd_comm_op_t *ops);
    void (*destroy)(void *ctx);
} fmd_comm_t;
#define FMD_COMM_ID_BYTES 128
int fmd_comm_rccl_unique_id(uint8_t id[FMD_COMM_ID_BYTES]);                       /* ncclGetUniqueId: rank 0 calls, the host carries it to the others */
int fmd_comm_rccl_init(int device, int rank, int world, const uint8_t id[FMD_COMM_ID_BYTES], fmd_comm_t **out);   /* ncclCommInitRank */
int fmd_comm_rccl_version(void);                                                 /* ncclGetVersion; 0 = librccl not loadable */
int fmd_comm_rccl_count(const fmd_comm_t *c);                                    /* ncclCommCount of a communicator made by fmd_comm_rccl_init; -1 otherwise */
void fmd_comm_free(fmd_comm_t *c);                                               /* calls destroy(ctx) and frees c if fmd_comm_rccl_init made it */

/* One pass of overlap discovery over the sequence ids 0 .. n_ids-1 on `world` GPUs, rank r on its own replica of the index:
 *   pass 1 on the ids r, r + world, ... (the reference's worker interleave, unitig.c:333, 398-399);
 *   key_shard: one all-to-all of the parked strands (64 bytes each) by minimizer key range;
 *   pass 2 + fm6_get_nei in `pieces` batches of the sorted order; while batch p + 1 is computed, the rows of batch p are packed
 *   on a second stream and sent to `root` (fmd_ovlp_pack_rows_dev; ids, records, offsets, variable parts).
 * On return (stream-ordered on `stream` for the caller's later work; the call itself synchronises with the host between pieces)
 * the root holds all n_ids rows: fmd_ovlp_dist_table().  The object keeps every buffer between steps. */
typedef struct fmd_ovlp_dist fmd_ovlp_dist_t;
typedef struct {
    uint64_t n_ids;
    int min_match;
    uint32_t max_len, max_nei;
    uint32_t pieces;        /* 0 = default (4, fewer for small shards) */
    int key_shard;          /* 0 / 1; where the two-pass form does not apply (fmd_ovlp_two_pass_ok) the shard is computed in id order */
    int root;
    int host_table;         /* root: 0 = the table stays in HBM, 1 = in pinned host memory (pieces staged through HBM), -1 = by free HBM, 2 = no table: row_sink below */
    size_t batch;           /* 0 = a piece is one batch; else pieces are cut further so that no batch exceeds this many strands */
    /* host_table = 2 (the same on every rank): the root keeps NO table.  Every piece of every peer, once it has landed in pinned host memory, is handed to
     * row_sink -- n_rows packed rows as fmd_ovlp_pack_rows_dev writes them: ids[n_rows], prec[n_rows], off[n_rows + 1] into var -- on the thread that runs the
     * step, while the GPUs compute and pack the next piece; the buffers are reused two pieces later.  What unitig.c:394-404 does by joining its workers into
     * one graph: here the consumer (fermi_amd/host/dist_root.c folds the rows into the 44.5-byte rows `unitig` walks) takes them as they arrive, and the
     * root's memory is what the consumer keeps.  A non-zero return fails the step on every rank (FMD_E_IO).  Only the root's row_sink / sink_ctx are read.
     * A step that fails (any rank, any reason) has handed the sink SOME of its pieces: what the consumer holds then is incomplete -- it starts over. */
    int (*row_sink)(void *ctx, uint64_t n_rows, const uint32_t *ids, const fmd_ovlp_rec_t *prec, const uint64_t *off, const uint8_t *var, uint32_t max_nei);
    void *sink_ctx;
} fmd_ovlp_dist_cfg_t;
typedef struct {            /* of the last step, this rank; milliseconds from HIP events / the host clock */
    double head_ms, key_exchange_ms, tail_ms;      /* compute stream: pass 1 + sort; the all-to-all and the re-sort; all pieces of pass 2 */
    double last_piece_pack_send_ms;                /* comm stream: pack + transfer of the LAST piece (what cannot hide under compute) */
    double gather_exposed_ms;                      /* HIP events: end of this rank's last compute kernel -> end of its part in the gather */
    double step_ms;                                /* host clock, whole step */
    uint64_t rows_computed, rows_sent, bytes_sent, bytes_received, key_rows_sent;
    int pieces, on_host /* 2: rows went to cfg.row_sink */, key_shard, two_pass;
} fmd_ovlp_dist_stats_t;
typedef struct {            /* root only: the table of the last step */
    int on_host;                  /* the arrays below are device (0) or pinned host (1) memory */
    uint64_t n_rows;              /* = n_ids when the step is complete */
    const fmd_ovlp_rec_t *prec;   /* records in arrival order (packed form: fmd_ovlp_pack_dev) */
    const uint32_t *ids;          /* sequence id of each row */
    const uint64_t *vaddr;        /* address of each row's variable part (same memory kind) */
    const uint32_t *row_of_id;    /* n_ids entries: the row that holds sequence id i */
} fmd_ovlp_dist_table_t;
int fmd_ovlp_dist_new(fmd_dev_t *h, fmd_comm_t *comm, const fmd_ovlp_dist_cfg_t *cfg, fmd_ovlp_dist_t **out);
int fmd_ovlp_dist_step(fmd_ovlp_dist_t *d, void *stream, fmd_ovlp_dist_stats_t *stats);
int fmd_ovlp_dist_table(fmd_ovlp_dist_t *d, fmd_ovlp_dist_table_t *t);     /* FMD_E_ARG on a job whose rows went to a sink */
/* this rank's own fixed-stride rows of the last step (tests, parity samples): row j describes sequence id ids[j] */
int fmd_ovlp_dist_local(fmd_ovlp_dist_t *d, uint64_t *n_rows, const uint64_t **d_ids, const fmd_ovlp_rec_t **d_rec, const fmd_intv_t **d_nei, const uint8_t **d_seq, uint32_t *seq_stride);
void fmd_ovlp_dist_free(fmd_ovlp_dist_t *d);

/* ---- k-mer harvest of `fermi correct`: fm6_traverse (exact.c:141) + ec_collect (correct.c:35-87)
 * over ALL 4^suf_len suffix buckets (what worker1 does, correct.c:272-279).  Emits one
 * (bucket, key, val) triple per solid k-mer: bucket = index into `solid[]` (correct.c:346-349),
 * key/val exactly what kh_put/kh_val store (correct.c:71-75).  The order of triples is
 * unspecified for the _dev form (the reference's own order is hash-table iteration order); the host
 * form returns them sorted by (bucket, key), or -- for an index whose frontiers do not fit beside it -- as four sorted parts.  w <= 27, w - suf_len <= 15.
 * d_status (4 x u64): [0] #triples, [1] non-zero = cap overflowed (re-run larger), [2] cnt[0],
 * [3] cnt[1] (correct.c:64-69). */
size_t fmd_kmer_work_bytes(uint64_t cap);
int fmd_kmer_collect_dev(fmd_dev_t *h, void *stream, int w, int min_occ, int suf_len, void *d_work, size_t work_bytes,
                         uint64_t cap, uint32_t *d_bucket, uint32_t *d_key, uint8_t *d_val, uint64_t *d_status);
/* the part of the harvest whose k-mers end in a base of seed_mask (bit c-1 = nt6 base c; 0xf = all): the trie is a forest
 * rooted at the last base, the parts are disjoint and each needs about a quarter of the frontier */
int fmd_kmer_collect_part_dev(fmd_dev_t *h, void *stream, int w, int min_occ, int suf_len, int seed_mask, void *d_work, size_t work_bytes,
                              uint64_t cap, uint32_t *d_bucket, uint32_t *d_key, uint8_t *d_val, uint64_t *d_status);
int fmd_kmer_collect(fmd_dev_t *h, int w, int min_occ, int suf_len, uint32_t **bucket, uint32_t **key, uint8_t **val,
                     uint64_t *n, int64_t cnt[2]);
/* The same for the k-mers that END in one of the bases of seed_mask (bit c-1 = base c): one GPU's shard of a harvest spread over
 * several (`fermi-amd correct -g`; the reference shards ec_collect over its threads by suffix bucket, correct.c:346-356).  Disjoint
 * masks covering 0xf give, together, exactly the triples of fmd_kmer_collect. */
int fmd_kmer_collect_seeds(fmd_dev_t *h, int w, int min_occ, int suf_len, int seed_mask, uint32_t **bucket, uint32_t **key, uint8_t **val,
                           uint64_t *n, int64_t cnt[2]);   /* outputs malloc'ed: fmd_host_free() */

/* ---- the correction pass of `fermi correct`: ec_fix1 / ec_fix (correct.c:121-256) ------------------------------
 * The table is what fmd_kmer_collect* returns (one triple per solid k-mer, any order), loaded into a device hash
 * table: a look-up kh_get(solid[x & (SUF_NUM-1)], x >> 2*SUF_LEN << 2) (correct.c:156-157) is one 8-byte load.
 * fmd_ecfix_*: read i = bytes [off[i], off[i+1]) of seqs (nt6 codes) and quals (phred + 33).  Both are rewritten in
 * place exactly as ec_fix leaves str.s and qual[i] after its two ec_fix1 passes (correct.c:237-243); info[i] = the
 * value ec_fix holds at correct.c:246, before the lower-case count (correct.c:247-252, host).  A lane keeps its best-
 * first queue (<= 256 paths, correct.c:114) and its trace in d_work; a read whose trace exceeds trace_cap gets
 * info = FMD_ECFIX_TRACE_FULL and must be run again FROM ITS ORIGINAL BYTES with a larger cap (the host form does). */
typedef struct fmd_ectab fmd_ectab_t;
#define FMD_ECFIX_TRACE_FULL ((int32_t)0x80000000)
int fmd_ectab_build_dev(int device, void *stream, int w, int suf_len, uint64_t n, const uint32_t *d_bucket, const uint32_t *d_key,
                        const uint8_t *d_val, fmd_ectab_t **out);
int fmd_ectab_build(int device, int w, int suf_len, uint64_t n, const uint32_t *bucket, const uint32_t *key, const uint8_t *val, fmd_ectab_t **out);
void fmd_ectab_free(fmd_ectab_t *t);
size_t fmd_ecfix_work_bytes(const fmd_ectab_t *t, size_t n, uint32_t trace_cap);
int fmd_ecfix_dev(fmd_ectab_t *t, void *stream, size_t n, uint8_t *d_seqs, uint8_t *d_quals, const uint64_t *d_off, int step, uint32_t trace_cap,
                  int32_t *d_info, void *d_work, size_t work_bytes);
int fmd_ecfix_batch(fmd_ectab_t *t, size_t n, uint8_t *seqs, uint8_t *quals, const uint64_t *off, int step, int32_t *info);
/* what the kernels launched on the table requested since the last reset: {table slots probed (8 B each), queue entries moved (16 B), trace entries
 * moved (8 B)} -- counted by the instrumented build only (*counting = 1; the shipped library answers zeros and *counting = 0), like fmd_dev_line_count */
int fmd_ectab_line_count(fmd_ectab_t *t, uint64_t counts[3], int reset, int *counting);

/* ---- fm6_retrieve (exact.c:100-127) in bulk: rank, `$read$` bi-interval and containment of each
 * sequence id (rec.rank, rec.k[], rec.status = -3 when contained, rec.len); no length threshold,
 * no overlap search.  This is what fm6_seqsort (seqsort.c:12-35) consumes. */
int fmd_seqinfo_dev(fmd_dev_t *h, void *stream, size_t n, const uint64_t *d_ids, uint32_t max_len, fmd_ovlp_rec_t *d_rec,
                    uint8_t *d_seq, uint32_t seq_stride, void *d_work, size_t work_bytes); /* work: fmd_ovlp_work_bytes(n, max_len, max_len-1) */
int fmd_seqinfo_batch(fmd_dev_t *h, size_t n, const uint64_t *ids, uint32_t max_len, fmd_ovlp_rec_t *rec);

/* ---- index construction: the BWT `fermi build` computes (cmd.c:378-484, build.c:11-50) ------
 * reads: nt6 bases of all reads back to back, NO sentinels; read i = reads[off[i], off[i+1]).
 * The text indexed is  read $ revcomp(read) $  per read in input order, sentinels ordered by
 * sequence id (ksa.c:54).  Palindrome trimming (cmd.c:457-463) is the caller's job
 * (fermi_amd/host).  bwt (host) must hold 2*(off[n]+n) bytes.  The _dev form returns a device
 * buffer the caller releases with fmd_dev_free(); uniform_len != 0 asserts all reads have
 * max_len bases. */
int fmd_build_bwt(int device, size_t n_reads, const uint8_t *reads, const uint64_t *off, uint8_t *bwt, uint64_t *n_sym);
int fmd_build_bwt_dev(int device, void *stream, size_t n_reads, const uint8_t *d_reads, const uint64_t *d_off,
                      uint64_t total_bases, uint32_t max_len, int uniform_len, uint8_t **d_bwt, uint64_t *n_sym);
void fmd_dev_free(void *d_ptr);
/* The same index without the byte BWT, for read sets whose text + BWT do not fit next to the index (7*10^8 x 100 bp:
 * 1.4*10^11 symbols): reads of ONE length appended in any number of calls (n x read_len nt6 bytes on the device, no
 * sentinels; they need not stay resident), text kept 4 bits per symbol, BWT slices written straight into the device
 * layout.  fmd_builder_finish releases the builder and returns the index fmd_dev_open_bwt_dev(fmd_build_bwt_dev(..))
 * would return. */
typedef struct fmd_builder fmd_builder_t;
int fmd_builder_new(int device, uint64_t n_reads, uint32_t read_len, fmd_builder_t **out);
int fmd_builder_add_dev(fmd_builder_t *b, void *stream, uint64_t n, const uint8_t *d_reads);
int fmd_builder_finish(fmd_builder_t *b, fmd_dev_t **out);
void fmd_builder_free(fmd_builder_t *b);
/* device memory for C hosts (the reference has no device; these are what a cgo/C caller uses to
 * stage batches): plain hipMalloc / hipMemcpyAsync behind the ABI. */
int fmd_dev_malloc(int device, size_t bytes, void **d_ptr);
int fmd_memcpy_h2d(void *d_dst, const void *h_src, size_t bytes, void *stream);
int fmd_memcpy_d2h(void *h_dst, const void *d_src, size_t bytes, void *stream);
/* device BWT -> host RLE\6 byte stream (`len<<3|sym`, ropebwt.c:132-136); *h_rle6 is malloc'ed,
 * release with fmd_host_free().  Prefix it with "RLE\6" and it is a .fmd the reference loads. */
int fmd_bwt_to_rle6(int device, const uint8_t *d_bwt, uint64_t n, uint8_t **h_rle6, uint64_t *n_bytes);
void fmd_host_free(void *p);

/* ---- diagnostics: random-gather ceiling of this GPU (DESIGN.md "practical roofline") -----
 * Reads n_access random aligned lines of `line_bytes` (64/128/256) from a working set of
 * ws_bytes with the same LDS-DMA gather the rank kernels use; returns milliseconds. */
int fmd_probe_gather(int device, uint64_t ws_bytes, uint32_t line_bytes, uint64_t n_access, int iters, float *ms);

#ifdef __cplusplus
}
#endif
#endif
