#!/usr/bin/env python3
"""One GPU: fmd_ovlp_dist_step over an RCCL communicator of ONE rank (all ids, every piece packed into the table on the second stream
while the next piece is computed) beside the plain sorted job -- what the pieces cost, and that the pack hides under compute
(gather_exposed_ms = the last piece's pack alone).  usage: dist_one_rank.py [n_reads] [pieces ...]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from fermi_amd import api, workload, dist as fdist

n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
plist = [int(x) for x in sys.argv[2:]] or [1, 4, 8]
dev = torch.device("cuda", 0)
lib = api.lib()
rd = workload.ReadsOnDevice.synth(n_reads, 100, 30, 0.0, dev)
d_bwt, n_sym = workload.build_bwt_on_device(rd, 0)
index = api.DevIndex.from_bwt_dev(d_bwt, n_sym, 0)
lib.fmd_dev_free(d_bwt)
del rd
torch.cuda.empty_cache()
n = 2 * n_reads
comm = fdist.RcclComm(api, None, 0, 1, 0)
sh = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for P in plist:
    job = fdist.DistJob(api, index, comm, n, 50, 100, 4, pieces=P, key_shard=0, root=0, host_table=0)
    job.step(sh); torch.cuda.synchronize()
    acc = []
    for _ in range(3):
        acc.append(job.step(sh).as_dict())
    torch.cuda.synchronize()
    m = lambda k: sum(a[k] for a in acc) / len(acc)
    print("%d pieces (%s): step %.1f ms = pass 1 + sort %.1f + pass 2 %.1f; last piece pack %.2f ms, exposed after the last kernel %.2f ms; check: %s"
          % (P, " : ".join(str(fdist.piece_begin(100, p + 1, P) - fdist.piece_begin(100, p, P)) for p in range(P)) + " %", m("step_ms"), m("head_ms"), m("tail_ms"),
             m("last_piece_pack_send_ms"), m("gather_exposed_ms"), fdist.check_table(torch, api, index, job, n, 50, 100, 4, dev, sample=100000, var_sample=500)[:40]), flush=True)
    job.free()
    torch.cuda.empty_cache()
