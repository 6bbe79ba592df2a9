/* build_cmd.c -- `fermi build [-f] [-o out.fmd] [-l maxlen] [-O] <in.fa>` (cmd.c:378-484): read the
 * sequences, convert to nt6, trim even-length self-reverse-complement reads by one base unless -O
 * (cmd.c:457-463), build the BWT of  read $ revcomp $ ...  on the GPU (fmd_build_bwt) and write the
 * RLD\2 container the reference writes (rld_writer.c).  The output file is byte-identical to
 * `fermi build`'s; -b (block size) other than 3 and -i (append to an index) are not supported. */
#define _GNU_SOURCE
#include <limits.h>
#include <time.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "fmd_host.h"

static double now_s(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec; }

int fmdh_build(const char *fa_path, const char *out_path, int device, int max_len, int no_fr)
{
    const int timing = getenv("FMD_TIMING") != 0;
    const double t0 = now_s();
    fmdh_seqio_t *io = fmdh_seq_open(fa_path);
    if (!io) { fprintf(stderr, "[E::%s] Fail to open the input file.\n", __func__); return 1; }
    size_t cap = 1 << 20, tot = 0, n = 0, ncap = 1 << 16;
    uint8_t *bases = (uint8_t *)malloc(cap);
    uint64_t *off = (uint64_t *)malloc((ncap + 1) * 8);
    int l, rc = 0;
    off[0] = 0;
    while ((l = fmdh_seq_read(io)) >= 0) {
        const char *s = fmdh_seq_bases(io);
        if (l > max_len) l = max_len;
        if (l == 0) continue; /* an empty record contributes nothing to the reference's index either */
        if (tot + (size_t)l + 8 > cap) { while (tot + (size_t)l + 8 > cap) cap <<= 1; bases = (uint8_t *)realloc(bases, cap); }
        for (int i = 0; i < l; ++i) bases[tot + i] = fmdh_nt6[(unsigned char)s[i]];
        if (no_fr) l = (int)fmdh_trim_palindrome(bases + tot, (uint32_t)l);
        if (n == ncap) { ncap <<= 1; off = (uint64_t *)realloc(off, (ncap + 1) * 8); }
        tot += (size_t)l; off[++n] = tot;
    }
    fmdh_seq_close(io);
    if (n == 0) { fprintf(stderr, "[E::%s] no sequences\n", __func__); free(bases); free(off); return 1; }
    uint64_t n_sym = 0;
    const double t1 = now_s();
    uint8_t *bwt = (uint8_t *)malloc(2 * (tot + n) + 64);
    rc = fmd_build_bwt(device, n, bases, off, bwt, &n_sym);
    const double t2 = now_s();
    if (rc) fprintf(stderr, "[E::%s] BWT construction failed: %s\n", __func__, fmd_strerror(rc));
    else {
        rc = fmdh_write_rld_from_bwt(bwt, n_sym, out_path);
        if (rc) fprintf(stderr, "[E::%s] cannot write `%s'\n", __func__, out_path);
    }
    if (timing) fprintf(stderr, "[M::%s] %zu sequences, %llu symbols: read + encode %.3f s, BWT on the GPU (incl. copies) %.3f s, .fmd %.3f s\n", __func__, n,
                        (unsigned long long)n_sym, t1 - t0, t2 - t1, now_s() - t2);
    free(bwt); free(bases); free(off);
    return rc ? 1 : 0;
}
