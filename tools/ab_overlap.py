#!/usr/bin/env python3
"""A/B of overlap discovery settings on one index (built once): for every setting given as ENV=VAL[,ENV=VAL...] on the command
line (use ';' inside FMD_OVLP_PIPE: FMD_OVLP_PIPE=4;6;10), the HIP-event time of `steps` passes over all strands, and whether the
records, neighbours and sequences are byte-identical to the first setting's.
Usage: python tools/ab_overlap.py [n_reads=20000000] [err=0.0] [steps=3] -- SETTING [SETTING...]     ('-' = defaults)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from fermi_amd import api, workload
import bench

args = sys.argv[1:]
cut = args.index("--") if "--" in args else len(args)
pos, settings = args[:cut], args[cut + 1:] or ["-"]
n_reads = int(pos[0]) if len(pos) > 0 else 20_000_000
err = float(pos[1]) if len(pos) > 1 else 0.0
steps = int(pos[2]) if len(pos) > 2 else 3
L = 100
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
rd = workload.ReadsOnDevice.synth(n_reads, L, 30, err, dev)
d_bwt, n_sym = workload.build_bwt_on_device(rd, 0)
del rd
index = api.DevIndex.from_bwt_dev(d_bwt, n_sym, 0)
api.lib().fmd_dev_free(d_bwt)
torch.cuda.empty_cache()
print("index: %d reads (e = %g), %d symbols, %.2f GB" % (n_reads, err, n_sym, index.hbm_bytes / 1e9), flush=True)
job = bench.OverlapJob(torch, api, index, dev, 2 * n_reads, 0, 1, L, 50)


def wsum(t):
    """position-weighted wrap-around sum of a byte tensor (multiple of 8 bytes) on the device"""
    v = t.reshape(-1).view(torch.int64)
    return int((v * (torch.arange(v.numel(), device=dev, dtype=torch.int64) % 1000003 + 1)).sum().item())


def digest():
    g = job.rec.view(torch.int32).view(job.n, 16)   # fmd_ovlp_rec_t: len = word 8, ext_len = 12, n_nei = 13
    n_nei = g[:, 13].clamp(0, job.max_nei)
    nei = job.nei.view(torch.int64).view(job.n, job.max_nei, 4)
    keep = (torch.arange(job.max_nei, device=dev)[None, :] < n_nei[:, None])
    a = wsum(job.rec)
    b = 0
    for o in range(0, job.n, 1 << 22):   # in pieces: the masks are as large as the arrays
        e = min(job.n, o + (1 << 22))
        b += wsum((nei[o:e] * keep[o:e, :, None]).contiguous().view(torch.uint8))
        used = (g[o:e, 8] + g[o:e, 12].clamp(min=0)).clamp(0, job.stride)
        seq = job.seq.view(job.n, job.stride)[o:e]
        b += wsum((seq * (torch.arange(job.stride, device=dev)[None, :] < used[:, None])).contiguous())
    return "%016x.%016x" % (a & (2**64 - 1), b & (2**64 - 1))


first = None
keep0 = None


def explain():
    """the first strands whose record, neighbours or appended bases differ from the first setting's"""
    r0, n0, s0 = keep0
    g0, g1 = r0.view(torch.int32).view(job.n, 16), job.rec.view(torch.int32).view(job.n, 16)
    bad = (g0 != g1).any(dim=1)
    nn = g0[:, 13].clamp(0, job.max_nei)
    km = (torch.arange(job.max_nei, device=dev)[None, :] < nn[:, None])[:, :, None]
    bad |= ((n0.view(torch.int64).view(job.n, job.max_nei, 4) != job.nei.view(torch.int64).view(job.n, job.max_nei, 4)) & km).any(dim=2).any(dim=1)
    ii = torch.nonzero(bad).flatten()
    print("   %d strands differ in record or neighbours; first: %s" % (ii.numel(), ii[:6].tolist()))
    for i in ii[:4].tolist():
        print("   strand %d\n     first : %s nei %s\n     now   : %s nei %s" % (i, g0[i].tolist(), n0.view(torch.int64).view(job.n, job.max_nei, 4)[i, :2].tolist(),
                                                                                  g1[i].tolist(), job.nei.view(torch.int64).view(job.n, job.max_nei, 4)[i, :2].tolist()))


for s in settings:
    saved = {}
    if s != "-":
        for kv in s.split(","):
            k, v = kv.split("=", 1)
            saved[k] = os.environ.get(k)
            os.environ[k] = v.replace(";", ",")
    job.rec.zero_(); job.nei.zero_(); job.seq.zero_()
    job.compute()
    torch.cuda.synchronize()
    dg = digest()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(job.stream)
    for _ in range(steps):
        job.compute()
    e1.record(job.stream)
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / steps
    if first is None:
        first = dg
        job.compute(); torch.cuda.synchronize()
        keep0 = (job.rec.clone(), job.nei.clone(), job.seq.clone())
    elif dg != first:
        job.rec.zero_(); job.nei.zero_(); job.seq.zero_()
        job.compute(); torch.cuda.synchronize()
        explain()
    print("%-60s %8.1f ms per pass = %.3e strands/s   %s" % (s, ms, job.n / ms * 1e3, ("same bytes" if dg == first else "DIFFERENT (%s vs %s)" % (dg, first)) + ("  digest " + dg if os.environ.get("AB_DIGEST") else "")), flush=True)
    for k, v in saved.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v
index.close()
