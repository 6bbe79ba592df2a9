#!/usr/bin/env python3
"""Stand-alone driver of fmd_smem_dev for profiling and A/B: N reads with substitutions against their own index (bench.py's smem
leg without the CPU side), K launches per setting of FMD_SMEM_REFILL, HIP-event ms per launch and a checksum of everything the
launch wrote (equal checksums = equal SMEMs).  FMD_HIP_LIB picks the build.
Usage: [N=50000000] [K=3] [ERR=0.01] [SELF=0] [REFILLS=8,16,32,64] python tools/run_smem.py"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from fermi_amd import api, workload

n = int(os.environ.get("N", "50000000")); K = int(os.environ.get("K", "3")); err = float(os.environ.get("ERR", "0.01"))
sm = int(os.environ.get("SELF", "0"))
refills = [int(v) for v in os.environ.get("REFILLS", "0").split(",")]
L, max_mem = 100, 8
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
lib = api.lib()
rd = workload.ReadsOnDevice.synth(n, L, 30, err, dev)
d_bwt, n_sym = workload.build_bwt_on_device(rd, 0)
index = api.DevIndex.from_bwt_dev(d_bwt, n_sym, 0)
lib.fmd_dev_free(d_bwt)
torch.cuda.empty_cache()
print("%s: index of %d reads (e = %g), %d symbols" % (os.path.basename(api.LIB_PATH), n, err, n_sym), flush=True)
mem = torch.zeros(n * max_mem * 4, dtype=torch.int64, device=dev)
n_mem = torch.zeros(n, dtype=torch.int32, device=dev)
wb = lib.fmd_smem_work_bytes(n, L)
work = torch.empty(wb, dtype=torch.uint8, device=dev)
st = torch.cuda.current_stream()
sh = C.c_void_p(st.cuda_stream)
w = torch.arange(1, max_mem * 4 + 1, dtype=torch.int64, device=dev) * 0x9E3779B1
for rf in refills:
    if rf: os.environ["FMD_SMEM_REFILL"] = str(rf)
    else: os.environ.pop("FMD_SMEM_REFILL", None)     # 0 = the library's own choice (64 for reads of one length, else 8)
    ms = []
    for it in range(K + 1):
        mem.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        api.check(lib.fmd_smem_dev(index.h, sh, n, rd.flat.data_ptr(), rd.off.data_ptr(), sm, L, max_mem, mem.data_ptr(), n_mem.data_ptr(), work.data_ptr(), wb))
        e1.record(st)
        torch.cuda.synchronize()
        if it:
            ms.append(e0.elapsed_time(e1))
    cs = int(((mem.view(n, max_mem * 4) * w[None, :]).sum(dim=1) * (torch.arange(n, device=dev) | 1)).sum().item()) ^ int(n_mem.to(torch.int64).sum().item())
    print("refill >= %2d lanes: %s ms per launch of %d reads (best %.1f = %.3e reads/s), %d SMEMs, checksum %016x" % (rf, " ".join("%.1f" % v for v in ms), n, min(ms), n / min(ms) * 1e3, int((n_mem & 0x7fffffff).sum()), cs & 0xffffffffffffffff), flush=True)
index.close()
