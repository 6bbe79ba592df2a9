/* oracle/batch_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 * Array-at-a-time wrappers over the oracle so that pytest / bench.py can check large batches
 * without a Python loop.  Same record layouts as include/fmd_hip.h so results compare bytewise.
 */
#include <stdlib.h>
#include <string.h>
#include <pthread.h>
#include "fmd_oracle.h"

void orc_rank1a_batch(const orc_rld_t *e, size_t n, const uint64_t *k, uint64_t *ok, int8_t *sym)
{
    size_t i;
    for (i = 0; i < n; ++i) {
        int c = orc_rank1a(e, k[i], ok + 6 * i);
        if (sym) sym[i] = (int8_t)c;
    }
}

void orc_rank2a_batch(const orc_rld_t *e, size_t n, const uint64_t *k, const uint64_t *l, uint64_t *ok, uint64_t *ol)
{
    size_t i;
    for (i = 0; i < n; ++i) orc_rank2a(e, k[i], l[i], ok + 6 * i, ol + 6 * i);
}

void orc_extend_batch(const orc_rld_t *e, size_t n, const orc_intv_t *ik, const uint8_t *is_back, orc_intv_t *ok)
{
    size_t i;
    for (i = 0; i < n; ++i) {
        int c;
        orc_extend(e, &ik[i], ok + 6 * i, is_back[i]);
        for (c = 0; c < 6; ++c) ok[6 * i + c].info = 0; /* fm6_extend leaves info untouched (exact.c:72-88) */
    }
}

/* retrieve: seqs is n rows of `stride` bytes (read order, zero padded), len[i], rank[i] */
void orc_retrieve_batch(const orc_rld_t *e, size_t n, const uint64_t *x, uint8_t *seqs, int stride,
                        int32_t *len, uint64_t *rank)
{
    size_t i;
    for (i = 0; i < n; ++i) {
        int l = 0;
        uint8_t *s = seqs + i * (size_t)stride;
        rank[i] = (uint64_t)orc_retrieve(e, x[i], s, stride, &l);
        len[i] = l;
        orc_reverse(l < stride ? l : stride, s);
    }
}
