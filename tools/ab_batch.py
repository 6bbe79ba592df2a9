#!/usr/bin/env python3
"""The sorted overlap job over all strands with pass 2 in batches of different sizes (the work area follows the batch): HIP-event time per pass.
Usage: python tools/ab_batch.py [n_reads=50000000] [err=0.0] [batch ...]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from fermi_amd import api, workload
n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
err = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
batches = [int(x) for x in sys.argv[3:]] or [20_000_000, 25_000_000, 34_000_000, 50_000_000]
L, min_match, max_nei = 100, 50, 4
stride = 2 * L
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
lib = api.lib()
rd = workload.ReadsOnDevice.synth(n_reads, L, 30, err, dev)
d_bwt, n_sym = workload.build_bwt_on_device(rd, 0); del rd
index = api.DevIndex.from_bwt_dev(d_bwt, n_sym, 0); lib.fmd_dev_free(d_bwt); torch.cuda.empty_cache()
n = 2 * n_reads
ids = torch.arange(n, dtype=torch.int64, device=dev)
st = torch.cuda.current_stream(); sh = C.c_void_p(st.cuda_stream)
rec = torch.zeros(n * 64, dtype=torch.uint8, device=dev); nei = torch.zeros(n * max_nei * 32, dtype=torch.uint8, device=dev); seq = torch.zeros(n * stride, dtype=torch.uint8, device=dev)
ref = None
for rnd in range(2):
    for batch in batches:
        b = min(batch, n)
        wb = lib.fmd_ovlp_sorted_work_bytes(n, b, L, min_match)
        free_b, _ = torch.cuda.mem_get_info()
        if wb + (4 << 30) > free_b:
            print("batch %d: work area %.1f GB does not fit (%.1f GB free)" % (b, wb / 1e9, free_b / 1e9), flush=True); continue
        work = torch.empty(wb, dtype=torch.uint8, device=dev)
        def run():
            api.check(lib.fmd_ovlp_sorted_dev(index.h, sh, n, ids.data_ptr(), min_match, L, max_nei, rec.data_ptr(), nei.data_ptr(), seq.data_ptr(), stride, work.data_ptr(), wb, b))
        run(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st); run(); run(); e1.record(st); torch.cuda.synchronize()
        s = int(rec.view(torch.int32).view(n, 16)[:, 11:14].to(torch.int64).sum().item())
        if ref is None: ref = s
        print("batch %9d (%d batches): %7.1f ms per pass, work area %.1f GB, %s" % (b, (n + b - 1) // b, e0.elapsed_time(e1) / 2, wb / 1e9, "same sums" if s == ref else "DIFFERENT"), flush=True)
        del work; torch.cuda.empty_cache()
