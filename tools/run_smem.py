#!/usr/bin/env python3
"""Stand-alone driver of fmd_smem_dev for profiling: N reads with substitutions against their own
index, K launches, prints ms per launch.  (bench.py's smem leg without the CPU side.)"""
import ctypes as C
import os
import sys
import time


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from fermi_amd import api, workload

n = int(os.environ.get("N", "4000000")); K = int(os.environ.get("K", "3")); err = float(os.environ.get("ERR", "0.01"))
sm = int(os.environ.get("SELF", "0"))
L, max_mem = 100, 8
dev = torch.device("cuda", 0)
reads = workload.synth_reads_host(n, L, 30, err)
rd = workload.ReadsOnDevice(reads, dev)
d_bwt, n_sym = workload.build_bwt_on_device(rd, 0)
index = api.DevIndex.from_bwt_dev(d_bwt, n_sym, 0)
api.lib().fmd_dev_free(d_bwt)
mem = torch.zeros(n * max_mem * 32, dtype=torch.uint8, device=dev)
n_mem = torch.zeros(n, dtype=torch.int32, device=dev)
wb = api.lib().fmd_smem_work_bytes(n, L)
work = torch.empty(wb, dtype=torch.uint8, device=dev)
sh = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for it in range(K):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    api.check(api.lib().fmd_smem_dev(index.h, sh, n, rd.flat.data_ptr(), rd.off.data_ptr(), sm, L, max_mem, mem.data_ptr(), n_mem.data_ptr(), work.data_ptr(), wb))
    torch.cuda.synchronize()
    print("smem launch %d: %.2f ms for %d reads (%.3g reads/s), %d SMEMs" % (it, (time.perf_counter() - t0) * 1e3, n, n / (time.perf_counter() - t0), int((n_mem & 0x7fffffff).sum())))
