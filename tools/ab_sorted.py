#!/usr/bin/env python3
"""The sorted overlap job (fmd_ovlp_sorted_dev: park after 32 bases, minimizer sort, the rest in that order) against the same ids
batch by batch in id order (fmd_ovlp_dev): HIP-event time of each over all 2N strands, and the records, neighbours and
sequences compared byte for byte (neighbours up to n_nei, sequences up to len + ext_len).
Usage: python tools/ab_sorted.py [n_reads=50000000] [err=0.0] [batch=20000000] [steps=2] [stride=1] [ENV=VAL ...]
(stride s: the ids 0, s, 2s, ...: the shard of one rank of s)"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from fermi_amd import api, workload

pos = [a for a in sys.argv[1:] if "=" not in a]
for a in sys.argv[1:]:
    if "=" in a:
        k, v = a.split("=", 1); os.environ[k] = v
n_reads = int(pos[0]) if len(pos) > 0 else 50_000_000
err = float(pos[1]) if len(pos) > 1 else 0.0
batch = int(pos[2]) if len(pos) > 2 else 20_000_000
steps = int(pos[3]) if len(pos) > 3 else 2
id_stride = int(pos[4]) if len(pos) > 4 else 1
L, min_match, max_nei = 100, 50, 4
stride = 2 * L
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
lib = api.lib()
rd = workload.ReadsOnDevice.synth(n_reads, L, 30, err, dev)
d_bwt, n_sym = workload.build_bwt_on_device(rd, 0)
del rd
index = api.DevIndex.from_bwt_dev(d_bwt, n_sym, 0)
lib.fmd_dev_free(d_bwt)
torch.cuda.empty_cache()
n = (2 * n_reads + id_stride - 1) // id_stride
batch = min(batch, n)
print("index: %d reads (e = %g), %d symbols, %.2f GB; %d strands, batches of %d" % (n_reads, err, n_sym, index.hbm_bytes / 1e9, n, batch), flush=True)
ids = torch.arange(n, dtype=torch.int64, device=dev) * id_stride
st = torch.cuda.current_stream()
sh = C.c_void_p(st.cuda_stream)
wb = lib.fmd_ovlp_sorted_work_bytes(n, batch, L, min_match)
work = torch.empty(wb, dtype=torch.uint8, device=dev)
print("work area %.1f GB (one batch of fmd_ovlp_dev: %.1f GB)" % (wb / 1e9, lib.fmd_ovlp_work_bytes(batch, L, min_match) / 1e9), flush=True)


def buffers():
    return (torch.zeros(n * 64, dtype=torch.uint8, device=dev), torch.zeros(n * max_nei * 32, dtype=torch.uint8, device=dev),
            torch.zeros(n * stride, dtype=torch.uint8, device=dev))


def run_ids(rec, nei, seq):
    for o in range(0, n, batch):
        c = min(batch, n - o)
        api.check(lib.fmd_ovlp_dev(index.h, sh, c, ids.data_ptr() + 8 * o, min_match, L, max_nei, rec.data_ptr() + 64 * o, nei.data_ptr() + o * max_nei * 32,
                                   seq.data_ptr() + o * stride, stride, work.data_ptr(), wb))


def run_sorted(rec, nei, seq):
    api.check(lib.fmd_ovlp_sorted_dev(index.h, sh, n, ids.data_ptr(), min_match, L, max_nei, rec.data_ptr(), nei.data_ptr(), seq.data_ptr(), stride,
                                      work.data_ptr(), wb, batch))


def timed(fn, bufs):
    fn(*bufs); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(steps):
        fn(*bufs)
    e1.record(st)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps


A, B = buffers(), buffers()
ta = timed(run_ids, A)
print("id order, batch by batch (fmd_ovlp_dev)     %8.1f ms per pass over %d strands = %.3e reads/s" % (ta, n, n / 2 / ta * 1e3), flush=True)
tb = timed(run_sorted, B)
print("sorted job (fmd_ovlp_sorted_dev)            %8.1f ms per pass over %d strands = %.3e reads/s   (%.1f %% of the id-order time)" % (tb, n, n / 2 / tb * 1e3, 100 * tb / ta), flush=True)
# ---- same bytes?
bad_rec = bad_nei = bad_seq = 0
first = []
for o in range(0, n, 1 << 22):
    e = min(n, o + (1 << 22))
    ga, gb = A[0].view(torch.int32).view(n, 16)[o:e], B[0].view(torch.int32).view(n, 16)[o:e]
    br = (ga != gb).any(dim=1)
    nn = ga[:, 13].clamp(0, max_nei)
    km = (torch.arange(max_nei, device=dev)[None, :] < nn[:, None])[:, :, None]
    na, nb = A[1].view(torch.int64).view(n, max_nei, 4)[o:e], B[1].view(torch.int64).view(n, max_nei, 4)[o:e]
    bn = ((na != nb) & km).any(dim=2).any(dim=1)
    used = (ga[:, 8] + ga[:, 12].clamp(min=0)).clamp(0, stride)
    sm = torch.arange(stride, device=dev)[None, :] < used[:, None]
    bs = ((A[2].view(n, stride)[o:e] != B[2].view(n, stride)[o:e]) & sm).any(dim=1)
    bad_rec += int(br.sum()); bad_nei += int(bn.sum()); bad_seq += int(bs.sum())
    if len(first) < 4:
        first += (torch.nonzero(br | bn | bs).flatten()[:4] + o).tolist()
print("strands whose record differs: %d, neighbours: %d, sequence + appended bases: %d   -> %s" % (bad_rec, bad_nei, bad_seq, "SAME BYTES" if bad_rec + bad_nei + bad_seq == 0 else "DIFFERENT"), flush=True)
for i in first[:4]:
    print("  strand %d\n    id order: %s\n    sorted  : %s" % (i, A[0].view(torch.int32).view(n, 16)[i].tolist(), B[0].view(torch.int32).view(n, 16)[i].tolist()))
index.close()
sys.exit(0 if bad_rec + bad_nei + bad_seq == 0 else 1)
