#!/usr/bin/env python3
"""k_ovl_seq_out, one thread per 16 output bytes, against one thread per word (FMD_SEQ_OUT_WORDS=1): the sorted job over all strands, HIP-event time
each way, alternating, and the tables compared byte for byte (whole sequence rows).  Usage: python tools/ab_seqout.py [n_reads=50000000] [err=0.0]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from fermi_amd import api, workload
n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
err = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
L, min_match, max_nei, batch = 100, 50, 4, 20_000_000
stride = 2 * L
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
lib = api.lib()
rd = workload.ReadsOnDevice.synth(n_reads, L, 30, err, dev)
d_bwt, n_sym = workload.build_bwt_on_device(rd, 0); del rd
index = api.DevIndex.from_bwt_dev(d_bwt, n_sym, 0); lib.fmd_dev_free(d_bwt); torch.cuda.empty_cache()
n = 2 * n_reads; batch = min(batch, n)
ids = torch.arange(n, dtype=torch.int64, device=dev)
st = torch.cuda.current_stream(); sh = C.c_void_p(st.cuda_stream)
wb = lib.fmd_ovlp_sorted_work_bytes(n, batch, L, min_match)
work = torch.empty(wb, dtype=torch.uint8, device=dev)
def buffers():
    return (torch.zeros(n * 64, dtype=torch.uint8, device=dev), torch.zeros(n * max_nei * 32, dtype=torch.uint8, device=dev), torch.zeros(n * stride, dtype=torch.uint8, device=dev))
def run(rec, nei, seq):
    api.check(lib.fmd_ovlp_sorted_dev(index.h, sh, n, ids.data_ptr(), min_match, L, max_nei, rec.data_ptr(), nei.data_ptr(), seq.data_ptr(), stride, work.data_ptr(), wb, batch))
def timed(bufs, steps=3):
    run(*bufs); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(steps): run(*bufs)
    e1.record(st); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps
A, B = buffers(), buffers()
res = {}
for rnd in range(2):
    for key, val, bufs in (("one thread per word", "1", A), ("one thread per 16 bytes", None, B)):
        if val: os.environ["FMD_SEQ_OUT_WORDS"] = val
        else: os.environ.pop("FMD_SEQ_OUT_WORDS", None)
        res.setdefault(key, []).append(timed(bufs))
for key, v in res.items():
    print("%-24s %s ms per pass over %d strands" % (key, " ".join("%8.1f" % t for t in v), n), flush=True)
same = all(bool(torch.equal(a, b)) for a, b in zip(A, B))
print("records, neighbours, whole sequence rows:", "SAME BYTES" if same else "DIFFERENT", flush=True)
sys.exit(0 if same else 1)
