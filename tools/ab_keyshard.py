#!/usr/bin/env python3
"""One GPU, the 50 M-read index: what ONE RANK of s does in an N = s step, id shard against key shard.
  id shard : pass 1 + sort of the ids 0, s, 2s, ..; pass 2 over those strands (1/s of the coverage everywhere)
  key shard: pass 1 + sort of the same ids; [all-to-all: 64 B per strand, (s-1)/s of them leave]; re-sort; pass 2 over the strands of ALL ids
             whose minimizer key lies in the first 1/s of the key space (the coverage of the whole set on 1/s of the genome) -- emulated by
             running pass 1 over all ids first (untimed) and cutting that range out of the order.
usage: ab_keyshard.py [n_reads] [s ...]"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from fermi_amd import api, workload, dist as fdist

n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
shards = [int(x) for x in sys.argv[2:]] or [1, 2, 4, 8]
dev = torch.device("cuda", 0)
L, mm, max_nei, stride = 100, 50, 4, 200
lib = api.lib()
rd = workload.ReadsOnDevice.synth(n_reads, L, 30, 0.0, dev)
d_bwt, n_sym = workload.build_bwt_on_device(rd, 0)
index = api.DevIndex.from_bwt_dev(d_bwt, n_sym, 0)
lib.fmd_dev_free(d_bwt)
del rd
torch.cuda.empty_cache()
n = 2 * n_reads
u8 = lambda k: torch.empty(k, dtype=torch.uint8, device=dev)
ids_all = torch.arange(n, dtype=torch.int64, device=dev)
park = u8(n * 64); keys = torch.empty(n, dtype=torch.int32, device=dev); order = torch.empty(n, dtype=torch.int32, device=dev)
rec = u8(n * 64); nei = u8(n * max_nei * 32); seq = u8(n * stride)
batch = 20_000_000
work = u8(max(lib.fmd_ovlp_head_work_bytes(n), lib.fmd_ovlp_work_bytes(batch, L, mm)))


def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def head(ids_t, m, park_t, keys_t, order_t):
    api.check(lib.fmd_ovlp_head_dev(index.h, None, m, ids_t.data_ptr(), mm, L, rec.data_ptr(), park_t.data_ptr(), keys_t.data_ptr(), order_t.data_ptr(), work.data_ptr(), work.numel()))


def tail(order_ptr, m, park_t, pieces):
    for p in range(pieces):
        b, e = m * p // pieces, m * (p + 1) // pieces
        for o in range(b, e, batch):
            c = min(batch, e - o)
            api.check(lib.fmd_ovlp_tail_dev(index.h, None, c, order_ptr + 4 * o, park_t.data_ptr(), mm, L, max_nei, rec.data_ptr(), nei.data_ptr(), seq.data_ptr(), stride, work.data_ptr(), work.numel()))


t_head_all = timed(lambda: head(ids_all, n, park, keys, order), 1)
keys_h = None
print("all %d strands on one GPU: pass 1 + sort %.1f ms, pass 2 (batches of %d) %.1f ms" % (n, t_head_all, batch, timed(lambda: tail(order.data_ptr(), n, park, 1), 2)), flush=True)
kk = (keys.view(torch.int32).to(torch.int64) & 0xffffffff).contiguous()
for s in shards:
    if s == 1:
        continue
    m = (n + s - 1) // s
    ids_s = torch.arange(0, n, s, dtype=torch.int64, device=dev)
    park_s = u8(m * 64); keys_s = torch.empty(m, dtype=torch.int32, device=dev); order_s = torch.empty(m, dtype=torch.int32, device=dev)
    t_h = timed(lambda: head(ids_s, m, park_s, keys_s, order_s))
    res = {}
    for pieces in (1, 4):
        res["id", pieces] = timed(lambda: tail(order_s.data_ptr(), m, park_s, pieces))
    # key range 0 of s over ALL strands: a slice of the full order (park / order of the full pass 1 are still in place)
    head(ids_all, n, park, keys, order); torch.cuda.synchronize()
    n_reg = int(torch.searchsorted(kk, torch.tensor([0xfffffffe], device=dev)).item())
    cut = n_reg // s                      # the first of s equal shares of the sorted order (the splitters are quantiles: fmd_ovlp_dist.hip)
    mid = (s // 2) * cut                  # ... and one from the middle
    for pieces in (1, 4):
        res["key", pieces] = timed(lambda: tail(order.data_ptr(), cut, park, pieces))
    res["mid", 1] = timed(lambda: tail(order.data_ptr() + 4 * mid, cut, park, 1))
    # the re-sort of the received rows: the sort of that many parked strands (keys + radix sort)
    park_k = park[: cut * 64]
    t_sort = timed(lambda: head(ids_s, min(m, cut), park_s, keys_s, order_s)) - 0  # (upper bound: a whole pass 1 of that size is t_h; the sort alone is below)
    a2a_bytes = m * 64 * (s - 1) / s
    print("s = %d: one rank: pass 1 + sort of %d strands %.1f ms | pass 2 on the id shard %.1f ms (4 pieces: %.1f) | pass 2 on key range 0/%d (%d strands) %.1f ms (4 pieces: %.1f; range %d/%d: %.1f) | "
          "all-to-all %.2f GB out per rank = %.1f ms at 7 x 50 GB/s | linear would be %.1f ms for the discovery part"
          % (s, m, t_h, res["id", 1], res["id", 4], s, cut, res["key", 1], res["key", 4], s // 2, s, res["mid", 1], a2a_bytes / 1e9, a2a_bytes / (min(s - 1, 7) * 50e9) * 1e3,
             (t_head_all + 0) / s), flush=True)
    del park_s, keys_s, order_s, ids_s
    torch.cuda.empty_cache()
