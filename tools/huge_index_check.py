#!/usr/bin/env python3
"""Scale check of the device layout and the search arithmetic at the size of the 35x human index
(1.4*10^11 symbols; positions beyond 2^37, symbol counts beyond 2^32, 1.5*10^9 rank
blocks) on ONE GPU, without needing such an index: the "BWT" is a synthetic periodic symbol string whose
rank has a closed form, generated in HBM, transcoded by the product (fmd_dev_open_bwt_dev), then
  * rank1a / rank2a at random and at extreme positions against the closed form,
  * backward search (prefix table + stepping) against the same recurrence evaluated with the closed form.
The string is not the BWT of any text -- rank and the search recurrence do not care.

    python tools/huge_index_check.py [n_symbols=1.4e11]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from fermi_amd import api  # noqa: E402

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 140_000_000_000
P = 1000     # period of the sentinel / N pattern (a multiple of 4)


def rank_closed(k):
    """counts of $,A,C,G,T,N in s[0..k] for s[p] = '$' if p % P == 0, 'N' if p % P == 1, else 1 + p % 4"""
    k = np.asarray(k, dtype=np.int64)
    out = np.zeros(k.shape + (6,), dtype=np.int64)
    neg = k < 0
    kk = np.where(neg, 0, k)
    out[..., 0] = kk // P + 1
    out[..., 5] = np.where(kk >= 1, (kk - 1) // P + 1, 0)
    for c in range(1, 5):
        r = c - 1
        out[..., c] = np.where(kk >= r, (kk - r) // 4 + 1, 0)
    out[..., 1] -= out[..., 0]      # p % P == 0 has p % 4 == 0: those would have been 'A'
    out[..., 2] -= out[..., 5]      # p % P == 1 has p % 4 == 1: those would have been 'C'
    out[neg] = 0
    return out


dev = torch.device("cuda", 0)
t0 = time.time()
bwt = torch.empty(n, dtype=torch.uint8, device=dev)
step = 1 << 30
for o in range(0, n, step):
    c = min(step, n - o)
    p = torch.arange(o, o + c, dtype=torch.int64, device=dev)
    s = (1 + p % 4).to(torch.uint8)
    m = p % P
    s[m == 0] = 0
    s[m == 1] = 5
    bwt[o:o + c] = s
    del p, s, m
torch.cuda.synchronize()
torch.cuda.empty_cache()
print("synthetic string of %d symbols in HBM: %.1f s" % (n, time.time() - t0), flush=True)
t0 = time.time()
d = api.DevIndex.from_bwt_dev(bwt.data_ptr(), n, 0)
print("transcoded: %.1f s, %.1f GB in HBM, mcnt = %s" % (time.time() - t0, d.hbm_bytes / 1e9, list(map(int, d.mcnt))), flush=True)
del bwt
torch.cuda.empty_cache()
want_m = rank_closed(np.array([n - 1]))[0]
assert list(map(int, d.mcnt[1:])) == list(map(int, want_m)), (d.mcnt, want_m)

rng = np.random.default_rng(11)
k = np.concatenate([rng.integers(0, n, 2_000_000), n - 1 - rng.integers(0, 10_000_000, 200_000), np.arange(0, 3000), [n - 1, n - 2, (1 << 37) - 1, 1 << 37, (1 << 37) + 1][:5 if n > (1 << 37) + 1 else 2]]).astype(np.uint64)
k = np.minimum(k, np.uint64(n - 1))
ok, sym = d.rank1a(k)
want = rank_closed(k.astype(np.int64))
assert np.array_equal(ok.astype(np.int64), want), "rank1a"
l = np.minimum(k + rng.integers(0, 5000, len(k)).astype(np.uint64), np.uint64(n - 1))
gk, gl = d.rank2a(k, l)
assert np.array_equal(gk.astype(np.int64), want) and np.array_equal(gl.astype(np.int64), rank_closed(l.astype(np.int64))), "rank2a"
print("rank1a / rank2a: %d positions up to %d match the closed form (max count %d)" % (len(k), int(k.max()), int(want.max())), flush=True)

# backward search recurrence with the closed-form rank (short patterns: on a periodic string the recurrence
# narrows by ~4x per base, so 14-16 bases already run from the whole range down to empty intervals)
cnt = np.concatenate([[0], np.cumsum(want_m)]).astype(np.int64)
m = 200_000
for plen in (8, 13, 14, 16, 20):
    pat = rng.integers(1, 5, (m, plen)).astype(np.uint8)
    ks = cnt[pat[:, -1]].copy(); ls = cnt[pat[:, -1] + 1] - 1
    alive = np.ones(m, dtype=bool)
    for j in range(plen - 2, -1, -1):
        c = pat[:, j].astype(np.int64)
        rk = rank_closed(ks - 1)[np.arange(m), c]; rl = rank_closed(ls)[np.arange(m), c]
        ks = np.where(alive, cnt[c] + rk, ks); ls = np.where(alive, cnt[c] + rl - 1, ls)
        alive = alive & (ks <= ls)
    gc, gb, ge = d.backward_search(pat)
    assert np.array_equal(gc > 0, alive), ("hits", plen)
    assert np.array_equal(gb[alive].astype(np.int64), ks[alive]) and np.array_equal(ge[alive].astype(np.int64), ls[alive]), ("intervals", plen)
    assert np.array_equal(gc[alive].astype(np.int64), (ls - ks + 1)[alive])
    print("backward search: %d patterns of %d bases, %d survive, intervals equal the recurrence" % (m, plen, int(alive.sum())), flush=True)
import ctypes as C  # noqa: E402
t0 = time.time()
bad, first = C.c_uint64(), C.c_uint64()
api.check(api.lib().fmd_dev_check_rank(d.h, C.byref(bad), C.byref(first)))
assert bad.value == 0, (bad.value, first.value)
print("rank self-check over all %d positions: consistent (%.1f s)" % (n, time.time() - t0), flush=True)
d.close()
print("OK")
