#!/usr/bin/env python3
"""Times fmd_ovlp_dev + fmd_ovlp_check_left_dev on N error-free reads (both strands): what the unitig table costs on the GPU."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from fermi_amd import api, workload

n = int(os.environ.get("N", "2000000")); K = int(os.environ.get("K", "3")); L = 100; mm = 50
dev = torch.device("cuda", 0)
reads = workload.synth_reads_host(n, L, 30, 0.0)
rd = workload.ReadsOnDevice(reads, dev)
d_bwt, n_sym = workload.build_bwt_on_device(rd, 0)
index = api.DevIndex.from_bwt_dev(d_bwt, n_sym, 0)
api.lib().fmd_dev_free(d_bwt)
n_ids, max_nei, stride = 2 * n, 4, 2 * L
ids = torch.arange(n_ids, dtype=torch.int64, device=dev)
rec = torch.zeros(n_ids * 64, dtype=torch.uint8, device=dev)
nei = torch.zeros(n_ids * max_nei * 32, dtype=torch.uint8, device=dev)
seq = torch.zeros(n_ids * stride, dtype=torch.uint8, device=dev)
wb = api.lib().fmd_ovlp_work_bytes(n_ids, L, mm)
work = torch.empty(wb, dtype=torch.uint8, device=dev)
sh = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for it in range(K):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    api.check(api.lib().fmd_ovlp_dev(index.h, sh, n_ids, ids.data_ptr(), mm, L, max_nei, rec.data_ptr(), nei.data_ptr(), seq.data_ptr(), stride, work.data_ptr(), wb))
    torch.cuda.synchronize(); t1 = time.perf_counter()
    api.check(api.lib().fmd_ovlp_check_left_dev(index.h, sh, n_ids, mm, L, rec.data_ptr(), seq.data_ptr(), stride, work.data_ptr(), wb))
    torch.cuda.synchronize(); t2 = time.perf_counter()
    r = rec.cpu().numpy().view(api.OVLP_DT)
    print("run %d: overlap %.2f ms, check_left %.2f ms for %d strands; reserved histogram %s" % (it, (t1 - t0) * 1e3, (t2 - t1) * 1e3, n_ids, dict(zip(*[x.tolist() for x in __import__('numpy').unique(r['reserved'], return_counts=True)]))))
