#!/usr/bin/env python3
"""Does the order of the READS matter to fm6_smem (fmd_smem_dev) and fm_backward_search (fmd_bsearch_dev) as it does to the walk?
The reads are the input here, so a minimizer of their own first bases costs no first pass: N reads with 1 % substitutions against
their own index, K launches each in input order, sorted by the minimizer of the first / last 32 bases, and in true genome order.
Usage: python tools/smem_locality_probe.py [n_reads=50000000] [steps=2]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from fermi_amd import api, synth, workload

n = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
L, max_mem, err, seed = 100, 8, 0.01, synth.DEFAULT_SEED
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
lib = api.lib()
rd = workload.ReadsOnDevice.synth(n, L, 30, err, dev)
d_bwt, n_sym = workload.build_bwt_on_device(rd, 0)
index = api.DevIndex.from_bwt_dev(d_bwt, n_sym, 0)
lib.fmd_dev_free(d_bwt)
torch.cuda.empty_cache()
reads = rd.flat[: n * L].view(n, L)
print("index: %d reads (e = %g), %.2f GB" % (n, err, index.hbm_bytes / 1e9), flush=True)


def minimizer(win, k=16):
    key = torch.empty(n, dtype=torch.int64, device=dev)
    for s in range(0, n, 4_000_000):
        t = (win[s:s + 4_000_000].to(torch.int64) - 1) & 3
        nk = t.shape[1] - k + 1
        v = torch.zeros((t.shape[0], nk), dtype=torch.int64, device=dev)
        for j in range(k):
            v = v * 4 + t[:, j:j + nk]
        h = (v * synth._i64(0x9E3779B97F4A7C15) + 0x7F4A7C15) & 0x7FFFFFFFFFFFFFFF
        h = ((h ^ (h >> 29)) * synth._i64(0xBF58476D1CE4E5B9)) & 0x7FFFFFFFFFFFFFFF
        key[s:s + 4_000_000] = (h >> 20).min(dim=1).values
    return key


G = max(n * L // 30, L)
r = torch.arange(n, dtype=torch.int64, device=dev)
pos = synth._umod(synth.rnd_torch(seed, 2, r), G - L + 1)
rev = synth._lsr(synth.rnd_torch(seed, 3, r), 63)
orders = [("input order", None), ("true genome order (per strand of the genome)", torch.argsort(pos + rev * (1 << 40))),
          ("minimizer of the first 32 bases", torch.argsort(minimizer(reads[:, :32]))), ("minimizer of the last 32 bases", torch.argsort(minimizer(reads[:, L - 32:])))]
del pos, rev, r
mem = torch.zeros(n * max_mem * 32, dtype=torch.uint8, device=dev)
n_mem = torch.zeros(n, dtype=torch.int32, device=dev)
wb = lib.fmd_smem_work_bytes(n, L)
work = torch.empty(wb, dtype=torch.uint8, device=dev)
cnt = torch.zeros(n, dtype=torch.int64, device=dev); beg = torch.zeros_like(cnt); end = torch.zeros_like(cnt)
st = torch.cuda.current_stream()
sh = C.c_void_p(st.cuda_stream)
off = rd.off


def timed(fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(steps):
        fn()
    e1.record(st)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps


for name, o in orders:
    flat = rd.flat if o is None else torch.cat([reads[o].reshape(-1), torch.zeros(64, dtype=torch.uint8, device=dev)])
    t_s = timed(lambda: api.check(lib.fmd_smem_dev(index.h, sh, n, flat.data_ptr(), off.data_ptr(), 0, L, max_mem, mem.data_ptr(), n_mem.data_ptr(), work.data_ptr(), wb)))
    t_b = timed(lambda: api.check(lib.fmd_bsearch_dev(index.h, sh, n, flat.data_ptr(), off.data_ptr(), cnt.data_ptr(), beg.data_ptr(), end.data_ptr())))
    print("%-48s fm6_smem %7.1f ms (%d SMEMs)   fm_backward_search %7.1f ms (%d hits)" % (name, t_s, int((n_mem & 0x7fffffff).sum()), t_b, int((cnt > 0).sum())), flush=True)
    del flat
index.close()
