/* oracle/fmd_ops_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of the FMD operations built on rank: bi-interval extension, backward
 * search, LF-walk retrieval, SMEM, overlap discovery and the `correct` k-mer harvest.
 * Restates exact.c, smem.c:13-80, unitig.c:38-179 and correct.c:35-87 of the reference.
 * Parity: pinned against oracle/_ref (the compiled reference) and tests/golden/ vectors.
 */
#include <stdlib.h>
#include <string.h>
#include <pthread.h>
#include "fmd_oracle.h"

/* ---- small helpers -------------------------------------------------------------------- */

const unsigned char orc_nt6_table[128] = { /* seq.c:12-21: $=0 A=1 C=2 G=3 T=4 other=5 */
    0, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5,
    5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5,
    5, 1, 5, 2, 5, 5, 5, 3, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 4, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5,
    5, 1, 5, 2, 5, 5, 5, 3, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 4, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5};

static inline int comp6(int c) { return (c >= 1 && c <= 4) ? 5 - c : c; } /* fermi.h:52 */

void orc_reverse(int l, uint8_t *s) /* seq.c:30-37 */
{
    int i, j;
    for (i = 0, j = l - 1; i < j; ++i, --j) { uint8_t t = s[i]; s[i] = s[j]; s[j] = t; }
}

void orc_revcomp6(int l, uint8_t *s) /* seq.c:46-56 */
{
    int i;
    orc_reverse(l, s);
    for (i = 0; i < l; ++i) s[i] = (uint8_t)comp6(s[i]);
}

static inline void vec_push(orc_intv_v *v, const orc_intv_t *x)
{
    if (v->n == v->m) {
        v->m = v->m ? v->m << 1 : 16;
        v->a = (orc_intv_t *)realloc(v->a, v->m * sizeof(orc_intv_t));
    }
    v->a[v->n++] = *x;
}

static inline void vec_reverse(orc_intv_v *v) /* exact.c:129-139 */
{
    size_t i, j;
    if (v->n < 2) return;
    for (i = 0, j = v->n - 1; i < j; ++i, --j) { orc_intv_t t = v->a[i]; v->a[i] = v->a[j]; v->a[j] = t; }
}

static inline void str_push(orc_str_t *s, int c)
{
    if (s->n + 1 >= s->m) {
        s->m = s->m ? s->m << 1 : 256;
        s->s = (uint8_t *)realloc(s->s, s->m);
    }
    s->s[s->n++] = (uint8_t)c;
    s->s[s->n] = 0;
}

/* ---- exact.c ---------------------------------------------------------------------------- */

void orc_set_intv(const orc_rld_t *e, int c, orc_intv_t *ik) /* fermi.h:53 */
{
    ik->x[0] = e->cnt[c];
    ik->x[1] = e->cnt[comp6(c)];
    ik->x[2] = e->cnt[c + 1] - e->cnt[c];
    ik->info = 0;
}

void orc_extend(const orc_rld_t *e, const orc_intv_t *ik, orc_intv_t ok[6], int is_back) /* exact.c:72-88 */
{
    static const int order[6] = {0, 4, 3, 2, 1, 5}; /* running-sum order of the other strand */
    uint64_t tk[6], tl[6], acc;
    int d = is_back ? 1 : 0, o = !d, i;
    orc_rank2a(e, ik->x[o] - 1, ik->x[o] - 1 + ik->x[2], tk, tl);
    for (i = 0; i < 6; ++i) {
        ok[i].x[o] = e->cnt[i] + tk[i];
        ok[i].x[2] = tl[i] - tk[i];
    }
    acc = ik->x[d];
    for (i = 0; i < 6; ++i) {
        ok[order[i]].x[d] = acc;
        acc += ok[order[i]].x[2];
    }
}

void orc_extend0(const orc_rld_t *e, const orc_intv_t *ik, orc_intv_t *ok0, int is_back) /* exact.c:90-98 */
{
    uint64_t tk[6], tl[6];
    int d = is_back ? 1 : 0, o = !d;
    orc_rank2a(e, ik->x[o] - 1, ik->x[o] - 1 + ik->x[2], tk, tl);
    ok0->x[o] = tk[0];
    ok0->x[d] = ik->x[d];
    ok0->x[2] = tl[0] - tk[0];
}

uint64_t orc_backward_search(const orc_rld_t *e, int len, const uint8_t *str,
                             uint64_t *sa_beg, uint64_t *sa_end) /* exact.c:7-23 */
{
    uint64_t k, l, t[6];
    int i, c = str[len - 1];
    k = e->cnt[c]; l = e->cnt[c + 1] - 1;
    for (i = len - 2; i >= 0; --i) {
        uint64_t ok, ol;
        c = str[i];
        /* rld_rank21 = two independent rank11 (rld.c:418-422) */
        orc_rank1a(e, k - 1, t); ok = t[c];
        orc_rank1a(e, l, t);     ol = t[c];
        k = e->cnt[c] + ok;
        l = e->cnt[c] + ol - 1;
        if (k > l) break;
    }
    if (k > l) return 0;
    *sa_beg = k; *sa_end = l;
    return l - k + 1;
}

int64_t orc_retrieve(const orc_rld_t *e, uint64_t x, uint8_t *s, int cap, int *len) /* exact.c:59-70 */
{
    uint64_t k = x, ok[6];
    int n = 0;
    for (;;) {
        int c = orc_rank1a(e, k, ok);
        k = e->cnt[c] + ok[c] - 1;
        if (c == 0) { *len = n; return (int64_t)k; }
        if (n < cap) s[n] = (uint8_t)c;
        ++n;
    }
}

orc_intv_t *orc_traverse(const orc_rld_t *e, int depth) /* exact.c:141-171 */
{
    orc_intv_t *rst = (orc_intv_t *)calloc((size_t)1 << (depth * 2), sizeof(orc_intv_t));
    orc_intv_v stack = {0, 0, 0};
    orc_intv_t ik, ok[6];
    ik.x[0] = ik.x[1] = 0; ik.x[2] = e->mcnt[0]; ik.info = 0;
    vec_push(&stack, &ik);
    while (stack.n) {
        int c, d;
        ik = stack.a[--stack.n];
        d = (int)(ik.info & 0xffffffffu);
        if (d == depth) { rst[ik.info >> 32] = ik; continue; }
        if (ik.x[2] == e->mcnt[0]) { for (c = 1; c < 5; ++c) orc_set_intv(e, c, &ok[c]); }
        else orc_extend(e, &ik, ok, 1);
        for (c = 1; c < 5; ++c) {
            if (ok[c].x[2] == 0) continue;
            ok[c].info = (ik.info + 1) | (uint64_t)(c - 1) << (32 + d * 2);
            vec_push(&stack, &ok[c]);
        }
    }
    free(stack.a);
    return rst;
}

/* ---- smem.c:13-80 ----------------------------------------------------------------------- */

#define MASK30 0x3fffffffull

static int smem1_core(const orc_rld_t *e, int len, const uint8_t *q, int x, orc_intv_v *mem,
                      int self_match, orc_intv_v *prev, orc_intv_v *curr)
{
    orc_intv_t ik, ok[6];
    orc_intv_v *t;
    int i, ret;
    size_t j;

    orc_set_intv(e, q[x], &ik);
    ik.info = (uint64_t)(x + 1);
    for (i = x + 1; i < len; ++i) { /* forward sweep */
        int c = comp6(q[i]);
        orc_extend(e, &ik, ok, 0);
        if (ok[c].x[2] != ik.x[2]) {
            if (ik.x[2] != ok[0].x[2]) vec_push(curr, &ik);
            if (!self_match && ok[0].x[2]) { ok[0].info = (uint64_t)i; vec_push(curr, &ok[0]); }
        }
        if (self_match ? ok[c].x[2] < 2 : ok[c].x[2] == 0) break;
        ik = ok[c]; ik.info = (uint64_t)(i + 1);
    }
    if (i == len) {
        vec_push(curr, &ik);
        if (!self_match) {
            orc_extend(e, &ik, ok, 0);
            if (ok[0].x[2]) { ok[0].info = (uint64_t)len; vec_push(curr, &ok[0]); }
        }
    }
    vec_reverse(curr);
    ret = (int)curr->a[0].info;
    t = curr; curr = prev; prev = t;

    mem->n = 0;
    for (i = x - 1; i >= -1; --i) { /* backward sweep over the whole list */
        int c = i < 0 ? 0 : q[i];
        for (j = 0, curr->n = 0; j < prev->n; ++j) {
            orc_intv_t *p = &prev->a[j];
            int cont, fl_match;
            orc_extend(e, p, ok, 1);
            fl_match = ok[0].x[2] && p->x[1] < e->mcnt[1];
            cont = self_match ? ok[c].x[2] > 1 : ok[c].x[2] != 0;
            if (!cont || fl_match || i == -1) {
                if (curr->n == 0 || fl_match) {
                    if (fl_match || mem->n == 0 || (uint64_t)(i + 1) < (mem->a[mem->n - 1].info >> 32 & MASK30)) {
                        ik = *p;
                        ik.info |= (uint64_t)(ok[0].x[2] != 0) << 63 | (uint64_t)(i + 1) << 32;
                        vec_push(mem, &ik);
                    }
                }
            }
            if (cont && (p->x[1] < e->mcnt[1] || curr->n == 0 || ok[c].x[2] != curr->a[curr->n - 1].x[2])) {
                ok[c].info = p->info;
                vec_push(curr, &ok[c]);
            }
        }
        if (curr->n == 0) break;
        t = curr; curr = prev; prev = t;
    }
    vec_reverse(mem);
    return ret;
}

int orc_smem1(const orc_rld_t *e, int len, const uint8_t *q, int x, orc_intv_v *mem, int self_match) /* smem.c:104 */
{
    orc_intv_v a = {0, 0, 0}, b = {0, 0, 0};
    int ret = smem1_core(e, len, q, x, mem, self_match, &a, &b);
    free(a.a); free(b.a);
    return ret;
}

int orc_smem(const orc_rld_t *e, int len, const uint8_t *q, orc_intv_v *mem, int self_match) /* smem.c:397-410 */
{
    orc_intv_v tmp = {0, 0, 0};
    int x = 0;
    size_t i;
    mem->n = 0;
    do {
        x = orc_smem1(e, len, q, x, &tmp, self_match);
        for (i = 0; i < tmp.n; ++i) vec_push(mem, &tmp.a[i]);
    } while (x < len);
    free(tmp.a);
    return (int)mem->n;
}

/* ---- unitig.c:38-179 -------------------------------------------------------------------- */

/* unitig.c:38-64.  seq[j] must match the end of a read. */
static orc_intv_t overlap_intv(const orc_rld_t *e, int len, const uint8_t *seq, int min, int j, int at5,
                               orc_intv_v *p, int inc_sentinel)
{
    orc_intv_t ik, ok[6];
    int dir = at5 ? 1 : -1, end = at5 ? len : -1, depth, c;
    p->n = 0;
    orc_set_intv(e, seq[j], &ik);
    for (depth = 1, j += dir; j != end; j += dir, ++depth) {
        c = at5 ? comp6(seq[j]) : seq[j];
        orc_extend(e, &ik, ok, !at5);
        if (ok[c].x[2] == 0) break;
        if (depth >= min && ok[0].x[2]) {
            if (inc_sentinel) { ok[0].info = (uint64_t)(int64_t)(j - dir); vec_push(p, &ok[0]); }
            else { ik.info = (uint64_t)(int64_t)(j - dir); vec_push(p, &ik); }
        }
        ik = ok[c];
    }
    vec_reverse(p);
    return ik;
}

int orc_is_contained(const orc_rld_t *e, int min_match, const uint8_t *s, int len,
                     orc_intv_t *intv, orc_intv_v *ovlp) /* unitig.c:77-91 */
{
    orc_intv_t ik, ok[6];
    int ret = 0;
    ovlp->n = 0;
    ik = overlap_intv(e, len, s, min_match, len - 1, 0, ovlp, 0);
    orc_extend(e, &ik, ok, 1);
    if (ik.x[2] != ok[0].x[2]) ret = -1;
    ik = ok[0];
    orc_extend(e, &ik, ok, 0);
    if (ik.x[2] != ok[0].x[2]) ret = -1;
    *intv = ok[0];
    return ret;
}

static int cmp_info(const void *a, const void *b)
{
    uint64_t x = ((const orc_intv_t *)a)->info, y = ((const orc_intv_t *)b)->info;
    return x < y ? -1 : x > y;
}

int orc_get_nei(const orc_rld_t *e, int min_match, int beg, orc_str_t *s, orc_intv_v *nei,
                orc_intv_v *prev, orc_intv_v *curr) /* unitig.c:93-179 with used=sorted=NULL */
{
    int ori_l = (int)s->n, i, c, rbeg, is_forked = 0;
    size_t j, ncat = 0;
    int *cat = 0;
    orc_intv_v *t;
    orc_intv_t ok[6], ok0;

    curr->n = nei->n = 0;
    if (prev->n == 0) {
        overlap_intv(e, (int)s->n - beg, s->s + beg, min_match, (int)s->n - beg - 1, 0, prev, 0);
        if (prev->n == 0) return -1;
        for (j = 0; j < prev->n; ++j) prev->a[j].info += (uint64_t)beg;
    }
    ncat = prev->n + 16;
    cat = (int *)malloc(ncat * sizeof(int));
    for (j = 0; j < prev->n; ++j) cat[j] = 0;
    while (prev->n) {
        for (j = 0, curr->n = 0; j < prev->n; ++j) {
            orc_intv_t *p = &prev->a[j];
            if (cat[j] < 0) continue;
            orc_extend(e, p, ok, 0);
            if (ok[0].x[2] && ori_l != (int)s->n) {
                orc_extend0(e, &ok[0], &ok0, 1);
                if (ok0.x[2]) {
                    if (ok[0].x[2] == p->x[2] && p->x[2] == ok0.x[2]) {
                        int cat0 = cat[j];
                        size_t ii;
                        ok0.info = (uint64_t)ori_l - (p->info & 0xffffffffu);
                        for (ii = j; ii < prev->n && cat[ii] == cat0; ++ii) cat[ii] = -1;
                        vec_push(nei, &ok0);
                        continue;
                    } /* else: a contained read; `used` is NULL here (unitig.c:124) */
                }
            }
            if (cat[j] < 0) continue;
            for (c = 1; c < 5; ++c)
                if (ok[c].x[2]) {
                    orc_extend0(e, &ok[c], &ok0, 1);
                    if (ok0.x[2]) {
                        ok[c].info = (p->info & 0xfffffff0ffffffffull) | (uint64_t)c << 32;
                        vec_push(curr, &ok[c]);
                    }
                }
        }
        if (curr->n) {
            uint32_t last, cat0;
            if (curr->n > ncat) { ncat = curr->n * 2; cat = (int *)realloc(cat, ncat * sizeof(int)); }
            c = (int)(curr->a[0].info >> 32 & 0xf);
            str_push(s, comp6(c));
            qsort(curr->a, curr->n, sizeof(orc_intv_t), cmp_info); /* keys are unique (unitig.c:132) */
            last = (uint32_t)(curr->a[0].info >> 32);
            cat[0] = 0;
            curr->a[0].info &= 0xffffffffu;
            for (j = 1, cat0 = 0; j < curr->n; ++j) {
                if ((uint32_t)(curr->a[j].info >> 32) != last) last = (uint32_t)(curr->a[j].info >> 32), cat0 = (uint32_t)j;
                cat[j] = (int)cat0;
                curr->a[j].info = (curr->a[j].info & 0xffffffffu) | (uint64_t)cat0 << 36;
            }
            if (cat0 != 0) is_forked = 1;
        }
        t = curr; curr = prev; prev = t;
    }
    free(cat);
    if (nei->n == 0) return -1;
    rbeg = ori_l - (int)(uint32_t)nei->a[0].info;
    if (nei->n == 1 && is_forked) { /* unitig.c:158-176 */
        orc_set_intv(e, 0, &ok0);
        for (i = rbeg; i < ori_l; ++i) {
            orc_extend(e, &ok0, ok, 0);
            ok0 = ok[comp6(s->s[i])];
        }
        for (i = ori_l; i < (int)s->n; ++i) {
            int c0 = -1, cnt = 0;
            orc_extend(e, &ok0, ok, 0);
            for (c = 1; c < 5; ++c)
                if (ok[c].x[2] && ok[c].x[0] <= nei->a[0].x[0] && ok[c].x[0] + ok[c].x[2] >= nei->a[0].x[0] + nei->a[0].x[2])
                    ++cnt, c0 = c;
            if (cnt == 0 && ok[0].x[2]) break;
            s->s[i] = (uint8_t)comp6(c0);
            ok0 = ok[c0];
        }
        s->n = (size_t)i; s->s[s->n] = 0;
    }
    if (nei->n > 1) { s->n = (size_t)ori_l; s->s[s->n] = 0; }
    return rbeg;
}


/* unitig.c:186-204 (check_left_simple) with used == NULL: 0, or -1 on a potential backward
 * bifurcation.  s = unitig so far (len bases), the terminal read starts at beg, its neighbour at rbeg. */
int orc_check_left_simple(const orc_rld_t *e, int min_match, int beg, int rbeg, const uint8_t *s, int len)
{
    orc_intv_v a = {0, 0, 0}, b = {0, 0, 0}, *prev = &a, *curr = &b, *t;
    orc_intv_t ok[6];
    int i, ret = 0;
    size_t j;
    overlap_intv(e, len, s, min_match, rbeg, 1, prev, 1);
    for (i = rbeg - 1; i >= beg && ret == 0; --i) {
        for (j = 0, curr->n = 0; j < prev->n; ++j) {
            orc_intv_t *p = &prev->a[j];
            orc_extend(e, p, ok, 1);
            if (ok[0].x[2] + ok[s[i]].x[2] != p->x[2]) { ret = -1; break; }
            vec_push(curr, &ok[s[i]]);
        }
        t = curr; curr = prev; prev = t;
    }
    free(a.a); free(b.a);
    return ret;
}

/* The lfork field of include/fmd_hip.h, EXACT: what check_left_simple (unitig.c:186-204) would meet on any edge whose
 * neighbour is N = the reverse complement of the strand x (len bases), told round by round without knowing the other
 * read.  The list is check_left_simple's own first step -- overlap_intv(N, from its first base, at5 = 1, sentinel
 * children) -- and each round is its inner loop with the base taken from the reads instead of from s[i]:
 *   round r passes iff every read of every interval either starts here (ok[0]) or goes on with one common base.
 * Returns D << 15 | R: rounds 0..R-1 pass; D = 1: round R has two bases; R = 0x7fff, D = 0: every read ended. */
unsigned orc_left_fork(const orc_rld_t *e, int min_match, const uint8_t *x, int len)
{
    orc_intv_v a = {0, 0, 0}, b = {0, 0, 0}, *prev = &a, *curr = &b, *t;
    orc_intv_t ok[6];
    uint8_t *n = (uint8_t *)malloc((size_t)len + 1);
    unsigned res = 0x7fffu;
    int i, r;
    size_t j;
    for (i = 0; i < len; ++i) { const int c = x[len - 1 - i]; n[i] = (uint8_t)((c >= 1 && c <= 4) ? 5 - c : c); }
    overlap_intv(e, len, n, min_match, 0, 1, prev, 1);
    for (r = 0; prev->n; ++r) {
        unsigned u = 0;
        int c, the = 0;
        for (j = 0; j < prev->n; ++j) {
            orc_extend(e, &prev->a[j], ok, 1);
            for (c = 1; c <= 5; ++c) if (ok[c].x[2]) u |= 1u << c;
        }
        if (u & (u - 1)) { res = 0x8000u | (unsigned)r; break; }   /* two different bases */
        if (u == 0) break;                                          /* every read ended: nothing left to disagree */
        for (c = 1; c <= 5; ++c) if (u >> c & 1) the = c;
        for (j = 0, curr->n = 0; j < prev->n; ++j) {
            orc_extend(e, &prev->a[j], ok, 1);
            if (ok[the].x[2]) vec_push(curr, &ok[the]);
        }
        t = curr; curr = prev; prev = t;
    }
    free(a.a); free(b.a); free(n);
    return res;
}

/* ---- correct.c:35-87 -------------------------------------------------------------------- */

void orc_ec_collect(const orc_rld_t *e, int w, int min_occ, int suf_len, const orc_intv_t *suf_intv,
                    orc_solid_t *out)
{
    int shift = (w - suf_len - 1) * 2, i;
    uint8_t *str;
    int str_l;
    orc_intv_v stack = {0, 0, 0};
    orc_intv_t ok[6], ik;
    if (suf_intv->x[2] == 0) return;
    str = (uint8_t *)calloc((size_t)w + 1, 1);
    ik = *suf_intv; ik.info = (uint64_t)suf_len << 4;
    vec_push(&stack, &ik);
    while (stack.n) {
        int c;
        ik = stack.a[--stack.n];
        orc_extend(e, &ik, ok, 1);
        str_l = (int)(ik.info >> 4) - suf_len;
        if (str_l) str[str_l - 1] = (uint8_t)(ik.info & 0xf);
        if ((int)(ik.info >> 4) == w) {
            uint32_t key = 0;
            int max_c = 6;
            uint64_t max = 0, rest;
            double r;
            for (c = 1; c <= 4; ++c)
                if (ok[c].x[2] > max) max = ok[c].x[2], max_c = c;
            if (max < (uint64_t)min_occ) continue;
            ++out->cnt[0];
            rest = ik.x[2] - max - ok[0].x[2] - ok[5].x[2];
            r = rest == 0 ? (double)max : (double)max / (double)rest;
            if (r > 31.) r = 31.;
            if (rest <= 7 && r >= min_occ) ++out->cnt[1];
            for (i = 0; i < str_l; ++i) key = (uint32_t)str[i] << shift | key >> 2;
            key = key << 2 | (uint32_t)(max_c - 1);
            if (out->n == out->m) {
                out->m = out->m ? out->m << 1 : 1024;
                out->key = (uint32_t *)realloc(out->key, out->m * 4);
                out->val = (uint8_t *)realloc(out->val, out->m);
            }
            out->key[out->n] = key;
            out->val[out->n++] = (uint8_t)((int)(r + .499) << 3 | (rest < 7 ? rest : 7));
        } else {
            for (c = 4; c >= 1; --c)
                if (ok[c].x[2] >= (uint64_t)min_occ) {
                    ok[c].info = ((ik.info >> 4) + 1) << 4 | (uint64_t)(c - 1);
                    vec_push(&stack, &ok[c]);
                }
        }
    }
    free(stack.a); free(str);
}

/* ---- batch driver: static start/step interleave over pthreads (unitig.c:394-404) -------- */

typedef struct {
    const orc_rld_t *e; size_t n; int len; const uint8_t *seqs;
    uint64_t *cnt, *beg, *end; int start, step;
} bs_job_t;

static void *bs_worker(void *d)
{
    bs_job_t *j = (bs_job_t *)d;
    size_t i;
    for (i = (size_t)j->start; i < j->n; i += (size_t)j->step) {
        uint64_t b = 0, en = 0;
        j->cnt[i] = orc_backward_search(j->e, j->len, j->seqs + i * (size_t)j->len, &b, &en);
        j->beg[i] = b; j->end[i] = en;
    }
    orc_counters_flush();
    return 0;
}

void orc_backward_search_batch(const orc_rld_t *e, size_t n, int len, const uint8_t *seqs,
                               uint64_t *cnt, uint64_t *beg, uint64_t *end, int n_threads)
{
    pthread_t *tid = (pthread_t *)calloc((size_t)n_threads, sizeof(pthread_t));
    bs_job_t *job = (bs_job_t *)calloc((size_t)n_threads, sizeof(bs_job_t));
    int t;
    for (t = 0; t < n_threads; ++t) {
        bs_job_t j = {e, n, len, seqs, cnt, beg, end, t, n_threads};
        job[t] = j;
        pthread_create(&tid[t], 0, bs_worker, &job[t]);
    }
    for (t = 0; t < n_threads; ++t) pthread_join(tid[t], 0);
    free(tid); free(job);
}
