#!/usr/bin/env python3
"""The bench legs with nothing around them, for the rocprofv3 --pmc passes (tools/pmc_collect.sh): K steps of each leg at
bench.py's sizes, no warm-up, no checks, no instrumented build -- every launch of a leg's kernels in the profile belongs
to one of its K steps.  PMC_LEGS=overlap_raw runs only overlap discovery on the raw-read index (the same kernels as the headline
leg: it needs a profile of its own).  Usage: python tools/pmc_legs.py [steps=2]"""
import ctypes as C, math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from fermi_amd import api, workload
import bench

K = int(sys.argv[1]) if len(sys.argv) > 1 else 2
n_reads = int(os.environ.get("FMD_BENCH_READS", "50000000"))
n_bs = int(os.environ.get("FMD_BENCH_BSEARCH_READS", "10000000"))
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
L = 100
sh = C.c_void_p(torch.cuda.current_stream().cuda_stream)
lib = api.lib()


def index_of(n, err):
    fmd = os.environ.get("PMC_FMD")          # bench.py's own in-run pass: the .fmd it has just written (no second build)
    if fmd and err == 0.0 and n == n_reads:
        ix = api.DevIndex.open(fmd, 0)
        return None, ix, ix.n
    rd = workload.ReadsOnDevice.synth(n, L, 30, err, dev)
    d_bwt, n_sym = workload.build_bwt_on_device(rd, 0)
    ix = api.DevIndex.from_bwt_dev(d_bwt, n_sym, 0)
    lib.fmd_dev_free(d_bwt)
    return rd, ix, n_sym


if os.environ.get("PMC_LEGS") == "overlap_raw":
    rd, ix, n_sym = index_of(n_reads, 0.01)
    del rd
    job = bench.OverlapJob(torch, api, ix, dev, 2 * n_reads, 0, 1, L, 50)
    for _ in range(K):
        job.compute()
    torch.cuda.synchronize()
    ix.close()
    print("pmc_legs: %d steps of overlap discovery on the raw-read index of %d reads" % (K, n_reads))
    sys.exit(0)
if os.environ.get("PMC_LEGS") == "ecfix":   # the correction pass: harvest (untimed, other kernel names) -> table -> K steps of k_ecfix, each on a fresh copy of the reads
    fmd = os.environ.get("PMC_FMD_RAW")
    rd = workload.ReadsOnDevice.synth(n_reads, L, 30, 0.01, dev)
    if fmd:
        ix = api.DevIndex.open(fmd, 0); n_sym = ix.n
    else:
        d_bwt, n_sym = workload.build_bwt_on_device(rd, 0)
        ix = api.DevIndex.from_bwt_dev(d_bwt, n_sym, 0)
        lib.fmd_dev_free(d_bwt)
    w = min(27, int(math.log(n_sym) / math.log(4) + 8.499)); suf = w - 15 if w > 15 else 1
    cap = max(1 << 22, 1 << int(math.ceil(math.log2(n_sym / 30.0 * 1.5))))
    wb = lib.fmd_kmer_work_bytes(cap); work = torch.empty(wb, dtype=torch.uint8, device=dev)
    ob = torch.empty(cap, dtype=torch.int32, device=dev); ok_ = torch.empty(cap, dtype=torch.int32, device=dev); ov = torch.empty(cap, dtype=torch.uint8, device=dev)
    st = torch.zeros(4, dtype=torch.int64, device=dev)
    api.check(lib.fmd_kmer_collect_dev(ix.h, sh, w, 3, suf, work.data_ptr(), wb, cap, ob.data_ptr(), ok_.data_ptr(), ov.data_ptr(), st.data_ptr()))
    torch.cuda.synchronize()
    n_out = int(st.cpu().numpy().view("u8")[0])
    ix.close(); del work; torch.cuda.empty_cache()
    t = C.c_void_p()
    api.check(lib.fmd_ectab_build_dev(0, sh, w, suf, n_out, ob.data_ptr(), ok_.data_ptr(), ov.data_ptr(), C.byref(t)))
    cap_t = int(os.environ.get("FMD_BENCH_EC_TRACE", "1024"))
    wb = lib.fmd_ecfix_work_bytes(t, n_reads, cap_t); work = torch.empty(wb, dtype=torch.uint8, device=dev)
    info = torch.zeros(n_reads, dtype=torch.int32, device=dev)
    for _ in range(K):
        s = rd.flat.clone(); q = torch.full((n_reads * L + 64,), ord("I"), dtype=torch.uint8, device=dev)
        api.check(lib.fmd_ecfix_dev(t, sh, n_reads, s.data_ptr(), q.data_ptr(), rd.off.data_ptr(), int(os.environ.get("FMD_BENCH_EC_STEP", "5")), cap_t, info.data_ptr(), work.data_ptr(), wb))
        torch.cuda.synchronize()
    lib.fmd_ectab_free(t)
    print("pmc_legs: %d steps of ec_fix over %d reads, table of %d solid k-mers" % (K, n_reads, n_out))
    sys.exit(0)
ONLY = os.environ.get("PMC_LEGS")   # e.g. PMC_LEGS=smem,kmer: a partial re-collection after one leg's kernels changed
want = lambda leg: not ONLY or leg in ONLY.split(",")
if want("overlap") or want("check_left"):   # overlap (+ check_left) at n_reads, e = 0
    rd, ix, n_sym = index_of(n_reads, 0.0)
    del rd
    job = bench.OverlapJob(torch, api, ix, dev, 2 * n_reads, 0, 1, L, 50)
    for _ in range(K):
        job.compute()
    torch.cuda.synchronize()
    job.alloc_link()
    for _ in range(K):
        job.check_left_linked()
    torch.cuda.synchronize()
    del job
    ix.close(); torch.cuda.empty_cache()
if want("k_bsearch"):   # backward search at n_bs, e = 0
    rd, ix, _ = index_of(n_bs, 0.0)
    cnt = torch.zeros(n_bs, dtype=torch.int64, device=dev); beg = torch.zeros_like(cnt); end = torch.zeros_like(cnt)
    for _ in range(K):
        api.check(lib.fmd_bsearch_dev(ix.h, sh, n_bs, rd.flat.data_ptr(), rd.off.data_ptr(), cnt.data_ptr(), beg.data_ptr(), end.data_ptr()))
    torch.cuda.synchronize()
    ix.close(); del rd, cnt, beg, end; torch.cuda.empty_cache()
if not (want("smem") or want("kmer")):
    print("pmc_legs: %d steps of %s" % (K, ONLY))
    sys.exit(0)
# SMEM + k-mer harvest at n_reads, e = 0.01
rd, ix, n_sym = index_of(n_reads, 0.01)
mem = torch.zeros(n_reads * 8 * 32, dtype=torch.uint8, device=dev); n_mem = torch.zeros(n_reads, dtype=torch.int32, device=dev)
wb = lib.fmd_smem_work_bytes(n_reads, L); work = torch.empty(wb, dtype=torch.uint8, device=dev)
for _ in range(K if want("smem") else 0):
    api.check(lib.fmd_smem_dev(ix.h, sh, n_reads, rd.flat.data_ptr(), rd.off.data_ptr(), 0, L, 8, mem.data_ptr(), n_mem.data_ptr(), work.data_ptr(), wb))
torch.cuda.synchronize()
del mem, n_mem, work, rd; torch.cuda.empty_cache()
w = min(27, int(math.log(n_sym) / math.log(4) + 8.499)); suf = w - 15 if w > 15 else 1
cap = max(1 << 22, 1 << int(math.ceil(math.log2(n_sym / 30.0 * 1.5))))
wb = lib.fmd_kmer_work_bytes(cap); work = torch.empty(wb, dtype=torch.uint8, device=dev)
ob = torch.empty(cap, dtype=torch.int32, device=dev); ok_ = torch.empty(cap, dtype=torch.int32, device=dev); ov = torch.empty(cap, dtype=torch.uint8, device=dev)
st = torch.zeros(4, dtype=torch.int64, device=dev)
for _ in range(K if want("kmer") else 0):
    api.check(lib.fmd_kmer_collect_dev(ix.h, sh, w, 3, suf, work.data_ptr(), wb, cap, ob.data_ptr(), ok_.data_ptr(), ov.data_ptr(), st.data_ptr()))
torch.cuda.synchronize()
ix.close()
print("pmc_legs: %d steps per leg at %d / %d reads" % (K, n_reads, n_bs))
