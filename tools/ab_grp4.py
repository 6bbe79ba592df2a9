#!/usr/bin/env python3
"""Groups of 4 lanes for strands with at most four candidates (FMD_GRP4, fmd_kernel_common.h) against round 3's classes (8, 12, 16, 21, 32):
the sorted overlap job over all strands of an index of reads with errors, HIP-event time each way, and the records, neighbours and
sequences compared byte for byte.  Usage: python tools/ab_grp4.py [n_reads=50000000] [err=0.01] [steps=2] [only=0|1|-] [switch=FMD_GRP4]
(only: that setting of FMD_GRP4 alone, steps + 1 passes, nothing compared -- the form tools/pmc_grp4.sh profiles)"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from fermi_amd import api, workload

n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
err = float(sys.argv[2]) if len(sys.argv) > 2 else 0.01
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
only = sys.argv[4] if len(sys.argv) > 4 and sys.argv[4] != "-" else None
VAR = sys.argv[5] if len(sys.argv) > 5 else "FMD_GRP4"   # the switch under test (FMD_GRP4, FMD_ROUTE_DOWN): 0 = without, 1 = with
L, min_match, max_nei, batch = 100, 50, 4, 20_000_000
stride = 2 * L
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
lib = api.lib()
rd = workload.ReadsOnDevice.synth(n_reads, L, 30, err, dev)
d_bwt, n_sym = workload.build_bwt_on_device(rd, 0)
del rd
index = api.DevIndex.from_bwt_dev(d_bwt, n_sym, 0)
lib.fmd_dev_free(d_bwt)
torch.cuda.empty_cache()
n = 2 * n_reads
batch = min(batch, n)
ids = torch.arange(n, dtype=torch.int64, device=dev)
st = torch.cuda.current_stream()
sh = C.c_void_p(st.cuda_stream)
wb = lib.fmd_ovlp_sorted_work_bytes(n, batch, L, min_match)
work = torch.empty(wb, dtype=torch.uint8, device=dev)
print("index: %d reads (e = %g), %d strands; work area %.1f GB" % (n_reads, err, n, wb / 1e9), flush=True)


def buffers():
    return (torch.zeros(n * 64, dtype=torch.uint8, device=dev), torch.zeros(n * max_nei * 32, dtype=torch.uint8, device=dev),
            torch.zeros(n * stride, dtype=torch.uint8, device=dev))


def run(rec, nei, seq):
    api.check(lib.fmd_ovlp_sorted_dev(index.h, sh, n, ids.data_ptr(), min_match, L, max_nei, rec.data_ptr(), nei.data_ptr(), seq.data_ptr(), stride,
                                      work.data_ptr(), wb, batch))


def timed(bufs):
    run(*bufs); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(steps):
        run(*bufs)
    e1.record(st)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps


if only is not None:
    os.environ[VAR] = only
    print("%s=%s: %.1f ms per pass over %d strands" % (VAR, only, timed(buffers()), n), flush=True)
    index.close()
    sys.exit(0)
A, B = buffers(), buffers()
res = {}
for rnd in range(2):   # (twice each way, alternating: clocks and caches settle)
    for key, val, bufs in ((VAR + "=0", "0", A), (VAR + "=1", "1", B)):
        os.environ[VAR] = val
        res.setdefault(key, []).append(timed(bufs))
for key, v in res.items():
    print("%-14s %s ms per pass over %d strands" % (key, " ".join("%8.1f" % t for t in v), n), flush=True)
bad = [0, 0, 0]
for o in range(0, n, 1 << 22):
    e = min(n, o + (1 << 22))
    ga, gb = A[0].view(torch.int32).view(n, 16)[o:e], B[0].view(torch.int32).view(n, 16)[o:e]
    nn = ga[:, 13].clamp(0, max_nei)
    km = (torch.arange(max_nei, device=dev)[None, :] < nn[:, None])[:, :, None]
    na, nb = A[1].view(torch.int64).view(n, max_nei, 4)[o:e], B[1].view(torch.int64).view(n, max_nei, 4)[o:e]
    used = (ga[:, 8] + ga[:, 12].clamp(min=0)).clamp(0, stride)
    sm = torch.arange(stride, device=dev)[None, :] < used[:, None]
    bad[0] += int((ga != gb).any(dim=1).sum()); bad[1] += int(((na != nb) & km).any(dim=2).any(dim=1).sum())
    bad[2] += int(((A[2].view(n, stride)[o:e] != B[2].view(n, stride)[o:e]) & sm).any(dim=1).sum())
print("strands whose record differs: %d, neighbours: %d, sequence + appended bases: %d   -> %s" % (bad[0], bad[1], bad[2], "SAME BYTES" if sum(bad) == 0 else "DIFFERENT"), flush=True)
index.close()
sys.exit(0 if sum(bad) == 0 else 1)
