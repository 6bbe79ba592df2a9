#!/usr/bin/env python3
"""Indexes beyond 2.6*10^10 symbols on one MI355X (task 7): synthetic N x 100 bp reads (30x, e = 0) generated in
HBM, BWT built by the prefix-bucketed GPU builder, device-wide rank self-check (the `chkbwt -r` equivalent), backward
search of a sample of the reads (every one must hit, interval size = its multiplicity >= 1), and overlap discovery on a
random sample of sequence ids against the REFERENCE (oracle/_ref when it travelled, the oracle otherwise) through the
.fmd the product writes.  Usage: python tools/scale_check.py [n_reads=250000000] [sample=20000]"""
import ctypes as C, os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from fermi_amd import api, workload
import bench

n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 250_000_000
sample = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
L = 100
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
lib = api.lib()
t0 = time.time()
rd = workload.ReadsOnDevice.synth(n_reads, L, 30, 0.0, dev)
torch.cuda.synchronize()
print("reads in HBM: %d x %d bp, %.1f s" % (n_reads, L, time.time() - t0), flush=True)
t0 = time.time()
d_bwt, n_sym = workload.build_bwt_on_device(rd, 0)
torch.cuda.synchronize()
t_build = time.time() - t0
free_b, total_b = torch.cuda.mem_get_info()
print("GPU BWT build: %d symbols in %.1f s (%.2e symbols/s); HBM in use after the build %.1f GB" % (n_sym, t_build, n_sym / t_build, (total_b - free_b) / 1e9), flush=True)
# a sample of the reads before they are released
rng = np.random.default_rng(11)
sel = np.sort(rng.choice(n_reads, min(sample, n_reads), replace=False))
q = rd.flat[: n_reads * L].view(n_reads, L)[torch.from_numpy(sel).to(dev)].cpu().numpy()
del rd
torch.cuda.empty_cache()
fmd_path = os.path.join(tempfile.gettempdir(), "fmd_scale_%d.fmd" % n_reads)
t0 = time.time()
workload.write_fmd_from_device_bwt(d_bwt, n_sym, fmd_path, 0)
print(".fmd written (GPU run-length pass + host RLD encoder): %.1f GB in %.1f s" % (os.path.getsize(fmd_path) / 1e9, time.time() - t0), flush=True)
t0 = time.time()
index = api.DevIndex.from_bwt_dev(d_bwt, n_sym, 0)
lib.fmd_dev_free(d_bwt)
print("device index: %.1f GB in HBM, %.1f s" % (index.hbm_bytes / 1e9, time.time() - t0), flush=True)
t0 = time.time()
bad, first = C.c_uint64(), C.c_uint64()
api.check(lib.fmd_dev_check_rank(index.h, C.byref(bad), C.byref(first)))
print("rank self-check over all %d positions: %d bad (%.1f s)" % (n_sym, bad.value, time.time() - t0), flush=True)
assert bad.value == 0
cnt, beg, end = index.backward_search(q)
assert (cnt >= 1).all() and np.array_equal(end - beg + 1, cnt), "a read of the set does not hit its own index"
print("backward search: all %d sampled reads hit (multiplicities 1..%d)" % (len(q), int(cnt.max())), flush=True)
base, ok = bench.cpu_bsearch(fmd_path, q, cnt, beg, end)
print("backward search vs %s on the sample: %s (%.0f reads/s on %d host threads)" % (base["kind"], "bit-exact" if ok else "MISMATCH", base["value"], base["cores"]), flush=True)
assert ok
ids = np.sort(rng.choice(2 * n_reads, min(sample, 2 * n_reads), replace=False)).astype(np.uint64)
t0 = time.time()
rec, nei, seq = index.overlap(ids, 50, max_len=100, max_nei=4, check_left=False)
t_g = time.time() - t0
base, ok = bench.cpu_overlap(fmd_path, ids, 50, rec, nei)
print("overlap discovery vs %s on %d random sequence ids: %s (GPU host-form call %.2f s; reference %.0f reads/s on %d threads)"
      % (base["kind"], len(ids), "bit-exact" if ok else "MISMATCH", t_g, base["value"], base["cores"]), flush=True)
assert ok
index.close()
os.remove(fmd_path)
print("scale check passed: %d reads, %d symbols" % (n_reads, n_sym))
