#!/usr/bin/env python3
"""The round-2 incident: k_ovl_nei_fast instances that spill a few VGPRs to scratch gave wrong neighbours for a
few strands in 10^7 -- only beside the walk kernel on a second stream (FMD_OVLP_PIPE), never alone.  This tool runs overlap
discovery of all strands of an N-read index in id order (fmd_ovlp_dev batches, the path FMD_OVLP_PIPE applies to) once in the
serial order and `reps` times in the pipelined order under whatever library FMD_HIP_LIB names and whatever runtime environment
the caller sets (AMD_SERIALIZE_KERNEL=3, ...), and counts the strands whose record or neighbours differ from the serial run.
Usage: FMD_HIP_LIB=.../libfmdhip_lb6.so python tools/scratch_incident.py [n_reads=20000000] [reps=3] [pipe=4,8,8,8]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from fermi_amd import api, workload
import bench

n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
pipe = sys.argv[3] if len(sys.argv) > 3 else "4,8,8,8"
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
rd = workload.ReadsOnDevice.synth(n_reads, 100, 30, 0.0, dev)
d_bwt, n_sym = workload.build_bwt_on_device(rd, 0)
del rd
index = api.DevIndex.from_bwt_dev(d_bwt, n_sym, 0)
api.lib().fmd_dev_free(d_bwt)
torch.cuda.empty_cache()
job = bench.OverlapJob(torch, api, index, dev, 2 * n_reads, 0, 1, 100, 50)
print("library %s, %d reads, %d strands in batches of %d; AMD_SERIALIZE_KERNEL=%s" % (os.path.basename(api.LIB_PATH), n_reads, job.n, job.batch, os.environ.get("AMD_SERIALIZE_KERNEL", "-")), flush=True)
os.environ.pop("FMD_OVLP_PIPE", None)
job.compute_in_id_order(); torch.cuda.synchronize()
ref = (job.rec.clone(), job.nei.clone())
g0 = ref[0].view(torch.int32).view(job.n, 16)
nn = g0[:, 13].clamp(0, job.max_nei)


def differing():
    bad = (g0 != job.rec.view(torch.int32).view(job.n, 16)).any(dim=1)
    for o in range(0, job.n, 1 << 22):
        e = min(job.n, o + (1 << 22))
        km = (torch.arange(job.max_nei, device=dev)[None, :] < nn[o:e, None])[:, :, None]
        bad[o:e] |= ((ref[1].view(torch.int64).view(job.n, job.max_nei, 4)[o:e] != job.nei.view(torch.int64).view(job.n, job.max_nei, 4)[o:e]) & km).any(dim=2).any(dim=1)
    return int(bad.sum().item())


job.rec.zero_(); job.nei.zero_(); job.seq.zero_()
job.compute_in_id_order(); torch.cuda.synchronize()
print("serial order again: %d strands differ from the first serial run" % differing(), flush=True)
os.environ["FMD_OVLP_PIPE"] = pipe
for r in range(reps):
    job.rec.zero_(); job.nei.zero_(); job.seq.zero_()
    job.compute_in_id_order(); torch.cuda.synchronize()
    print("pipelined order (FMD_OVLP_PIPE=%s), run %d: %d strands differ from the serial run" % (pipe, r, differing()), flush=True)
index.close()
