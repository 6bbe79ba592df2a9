/* oracle/ref_ec_harness.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * ec_collect (correct.c:35-87) is `static` in the reference, so the only way to pin the oracle's
 * restatement of it is to compile the reference's own correct.c into a harness: this file
 * #includes it FROM /root/reference at build time (make -C oracle ref; nothing is copied) and
 * dumps the content of the per-bucket `solid` hash tables that fm6_ec_correct's first phase
 * builds (correct.c:341-356).  Output: (bucket, key, val) triples in bucket order.
 */
#include "correct.c"

int refec_collect(const char *fn, int w, int min_occ, int suf_len, uint32_t **o_bucket, uint32_t **o_key, uint8_t **o_val,
                  uint64_t *o_n, int64_t cnt[2])
{
    rld_t *e = rld_restore(fn);
    fmecopt_t opt;
    fmintv_t *top;
    uint64_t n = 0, m = 1024;
    uint32_t *B = malloc(m * 4), *K = malloc(m * 4);
    uint8_t *V = malloc(m);
    int b;
    if (e == 0) return -1;
    memset(&opt, 0, sizeof(opt));
    opt.w = w; opt.min_occ = min_occ;
    compute_SUF(suf_len);
    top = fm6_traverse(e, SUF_LEN);
    cnt[0] = cnt[1] = 0;
    for (b = 0; b < SUF_NUM; ++b) {
        shash_t *h = kh_init(solid);
        khint_t k;
        ec_collect(e, &opt, SUF_LEN, &top[b], h, cnt);
        for (k = kh_begin(h); k != kh_end(h); ++k) {
            if (!kh_exist(h, k)) continue;
            if (n == m) { m <<= 1; B = realloc(B, m * 4); K = realloc(K, m * 4); V = realloc(V, m); }
            B[n] = (uint32_t)b; K[n] = kh_key(h, k); V[n] = kh_val(h, k); ++n;
        }
        kh_destroy(solid, h);
    }
    free(top);
    rld_destroy(e);
    *o_bucket = B; *o_key = K; *o_val = V; *o_n = n;
    return 0;
}
