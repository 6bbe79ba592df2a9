#!/usr/bin/env python3
"""Second locality probe: what a sort the product CAN do is worth.  Every strand's last W bases (what a first pass of the walk
would have in hand after W steps) -> the minimizer (smallest hashed k-mer) of that window -> all 2N strands sorted by it ->
overlap discovery in batches taken from the sorted order.  Compared with the same batches in id order and in true genome order.
Usage: python tools/locality_probe2.py [n_reads=50000000] [batch=20000000] [W,k ...]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from fermi_amd import api, synth, workload

n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 20_000_000
combos = [tuple(int(v) for v in a.split(",")) for a in sys.argv[3:]] or [(32, 16), (48, 16), (48, 14), (64, 16)]
L = 100
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
lib = api.lib()
seed = synth.DEFAULT_SEED
rd = workload.ReadsOnDevice.synth(n_reads, L, 30, 0.0, dev)
reads = rd.flat[: n_reads * L].view(n_reads, L)
Wmax = max(w for w, _ in combos)
# last Wmax bases of strand 2r (the read) and of strand 2r + 1 (its reverse complement), as 0..3
tails = torch.empty((2 * n_reads, Wmax), dtype=torch.uint8, device=dev)
tails[0::2] = reads[:, L - Wmax:] - 1
tails[1::2] = (4 - reads[:, :Wmax]).flip(1)
d_bwt, n_sym = workload.build_bwt_on_device(rd, 0)
del rd, reads
index = api.DevIndex.from_bwt_dev(d_bwt, n_sym, 0)
lib.fmd_dev_free(d_bwt)
torch.cuda.empty_cache()
print("index: %d reads, %d symbols, %.2f GB" % (n_reads, n_sym, index.hbm_bytes / 1e9), flush=True)


def minimizer_keys(W, k):
    key = torch.empty(2 * n_reads, dtype=torch.int64, device=dev)
    CH = 4_000_000
    for s in range(0, 2 * n_reads, CH):
        t = tails[s:s + CH, Wmax - W:].to(torch.int64)
        nk = W - k + 1
        v = torch.zeros((t.shape[0], nk), dtype=torch.int64, device=dev)
        for j in range(k):
            v = v * 4 + t[:, j:j + nk]
        h = (v * synth._i64(0x9E3779B97F4A7C15) + 0x7F4A7C15) & 0x7FFFFFFFFFFFFFFF
        h = ((h ^ (h >> 29)) * synth._i64(0xBF58476D1CE4E5B9)) & 0x7FFFFFFFFFFFFFFF
        key[s:s + CH] = (h >> 20).min(dim=1).values
    return key


G = max(n_reads * L // 30, L)
r = torch.arange(n_reads, dtype=torch.int64, device=dev)
pos = synth._umod(synth.rnd_torch(seed, 2, r), G - L + 1)
rev = synth._lsr(synth.rnd_torch(seed, 3, r), 63)
gkey = torch.empty(2 * n_reads, dtype=torch.int64, device=dev)
gkey[0::2] = torch.where(rev == 0, pos + L, pos + (1 << 40))
gkey[1::2] = torch.where(rev == 1, pos + L, pos + (1 << 40))
orders = [("id order", torch.arange(2 * n_reads, dtype=torch.int64, device=dev)), ("true genome order", torch.argsort(gkey))]
del gkey, pos, rev, r
for W, k in combos:
    kk = minimizer_keys(W, k)
    o = torch.argsort(kk, stable=True)
    ks = kk[o]
    runs = int((ks[1:] != ks[:-1]).sum().item()) + 1
    orders.append(("minimizer of the last %d bases, k = %d (%.2f strands per key)" % (W, k, 2 * n_reads / runs), o))
    del kk, ks
del tails
torch.cuda.empty_cache()

n = min(batch, 2 * n_reads)
max_nei, stride = 4, 2 * L
rec = torch.zeros(n * 64, dtype=torch.uint8, device=dev)
nei = torch.zeros(n * max_nei * 32, dtype=torch.uint8, device=dev)
seq = torch.zeros(n * stride, dtype=torch.uint8, device=dev)
wb = lib.fmd_ovlp_work_bytes(n, L, 50)
work = torch.empty(wb, dtype=torch.uint8, device=dev)
st = torch.cuda.current_stream()
sh = C.c_void_p(st.cuda_stream)
for name, o in orders:
    o = o.contiguous()

    def run(fn_full):
        for b in range(0, 2 * n_reads, n):
            c = min(n, 2 * n_reads - b)
            if fn_full:
                api.check(lib.fmd_ovlp_dev(index.h, sh, c, o.data_ptr() + 8 * b, 50, L, max_nei, rec.data_ptr(), nei.data_ptr(), seq.data_ptr(), stride, work.data_ptr(), wb))
            else:
                api.check(lib.fmd_seqinfo_dev(index.h, sh, c, o.data_ptr() + 8 * b, L, rec.data_ptr(), seq.data_ptr(), stride, work.data_ptr(), wb))
    res = []
    for full in (True, False):
        run(full); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st); run(full); e1.record(st)
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1))
    print("%-70s all %d strands: discovery %7.1f ms, walk alone (info_only) %7.1f ms" % (name, 2 * n_reads, res[0], res[1]), flush=True)
index.close()
