#!/usr/bin/env python3
"""How much would the walk gain if strands that lie next to each other on the genome were in flight together?
The synthetic reads know where they came from (synth.py: position and strand of read r), so the ids of one batch can be handed
to fmd_ovlp_dev / fmd_seqinfo_dev in ANY order: identity (the bench), sorted by the genome coordinate of the strand's last base
(an upper bound nobody can have without the layout), and in groups of g coordinate-neighbours with the groups shuffled (what a
minimizer sort of the strands' tails would give).  Same strands, same results, only the order in the batch changes.
Usage: python tools/locality_probe.py [n_reads=50000000] [batch=20000000] [steps=2]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from fermi_amd import api, synth, workload
import bench

n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 20_000_000
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
L = 100
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
lib = api.lib()
rd = workload.ReadsOnDevice.synth(n_reads, L, 30, 0.0, dev)
d_bwt, n_sym = workload.build_bwt_on_device(rd, 0)
del rd
index = api.DevIndex.from_bwt_dev(d_bwt, n_sym, 0)
lib.fmd_dev_free(d_bwt)
torch.cuda.empty_cache()
print("index: %d reads, %d symbols, %.2f GB" % (n_reads, n_sym, index.hbm_bytes / 1e9), flush=True)

# genome coordinate of every strand's LAST base, per orientation (strand 2r = read r as given, 2r + 1 = its reverse complement)
G = max(n_reads * L // 30, L)
r = torch.arange(n_reads, dtype=torch.int64, device=dev)
pos = synth._umod(synth.rnd_torch(synth.DEFAULT_SEED, 2, r), G - L + 1)
rev = synth._lsr(synth.rnd_torch(synth.DEFAULT_SEED, 3, r), 63)          # 1: the read is the reverse complement of the window
key = torch.empty(2 * n_reads, dtype=torch.int64, device=dev)
# orientation of strand s = rev ^ (s & 1); '+' strands are walked from pos + L backward, '-' strands from pos forward
key[0::2] = torch.where(rev == 0, pos + L, pos + (1 << 40))
key[1::2] = torch.where(rev == 1, pos + L, pos + (1 << 40))
order = torch.argsort(key)
del key, pos, rev, r
n = min(batch, 2 * n_reads)
gen = torch.Generator(device="cpu"); gen.manual_seed(5)


def grouped(g):
    """the first n of the sorted order in groups of g neighbours, groups shuffled"""
    m = n // g * g
    o = order[:m].view(-1, g)
    p = torch.randperm(o.shape[0], generator=gen).to(dev)
    return torch.cat([o[p].reshape(-1), order[m:n]])


orders = [("identity (ids 0..n-1)", torch.arange(n, dtype=torch.int64, device=dev)),
          ("the sorted strands, shuffled one by one", order[:n][torch.randperm(n, generator=gen).to(dev)]),
          ("sorted by genome coordinate", order[:n].clone())]
for g in (4, 8, 16, 32, 64, 256):
    orders.append(("groups of %d neighbours, shuffled" % g, grouped(g)))

max_nei, stride = 4, 2 * L
rec = torch.zeros(n * 64, dtype=torch.uint8, device=dev)
nei = torch.zeros(n * max_nei * 32, dtype=torch.uint8, device=dev)
seq = torch.zeros(n * stride, dtype=torch.uint8, device=dev)
wb = lib.fmd_ovlp_work_bytes(n, L, 50)
work = torch.empty(wb, dtype=torch.uint8, device=dev)
st = torch.cuda.current_stream()
sh = C.c_void_p(st.cuda_stream)


def timed(fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(steps):
        fn()
    e1.record(st)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps


for name, ids in orders:
    ids = ids.contiguous()
    full = timed(lambda: api.check(lib.fmd_ovlp_dev(index.h, sh, n, ids.data_ptr(), 50, L, max_nei, rec.data_ptr(), nei.data_ptr(), seq.data_ptr(), stride, work.data_ptr(), wb)))
    info = timed(lambda: api.check(lib.fmd_seqinfo_dev(index.h, sh, n, ids.data_ptr(), L, rec.data_ptr(), seq.data_ptr(), stride, work.data_ptr(), wb)))
    print("%-44s discovery %7.1f ms   walk alone (info_only: no candidate lists) %7.1f ms   per %d strands" % (name, full, info, n), flush=True)
index.close()
