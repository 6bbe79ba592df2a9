#!/usr/bin/env python3
"""bench.py -- reads/s through FMD backward search on MI355X (BASELINE.json configs[1]).

One "step" = one pass of fm_backward_search (exact.c:7) over the whole batch of synthetic reads,
all inputs (reads, offsets, index) already resident in HBM.  N GPUs: one process per GPU, the
full index replicated in each GPU's HBM, every rank searches its own batch of the same size
(weak scaling, no data-path collective); value = all reads searched by all ranks / max-over-ranks
wall time.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Environment knobs (defaults reproduce the BASELINE config): FMD_BENCH_READS (10_000_000),
FMD_BENCH_ERR (0 = every read hits), FMD_BENCH_CPU_SAMPLE (1_000_000 reads for the CPU baseline).
"""
import argparse
import ctypes as C
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s HBM3E
BYTES_PER_RANK_QUERY = 128     # one rank block + its counts


def pmc_traffic(key):
    """HBM bytes per launch/step measured by the separate rocprofv3 --pmc passes (profiles/pmc_traffic.json)."""
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json"))).get(key)
        if pmc:
            return (pmc["fetch_kb"] * pmc["fetch_calibration"] + pmc["write_kb"]) * 1024.0, pmc["source"]
    except Exception:
        pass
    return None, None


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def cpu_baseline(fmd_path, reads_host, sample, gpu_cnt, gpu_beg, gpu_end):
    """fm_backward_search on the host cores over a bounded sample, timed beside the GPU.
    Uses the compiled reference (oracle/_ref) when it travelled with the repo, else our C port
    (oracle/).  Also the parity check of the GPU results on that sample."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    cores = os.cpu_count() or 1
    n = min(sample, len(reads_host))
    q = np.ascontiguousarray(reads_host[:n])
    cnt = np.zeros(n, dtype=np.uint64); beg = np.zeros(n, dtype=np.uint64); end = np.zeros(n, dtype=np.uint64)
    drv = os.path.join(ROOT, "oracle", "_ref", "libref_driver.so")
    if os.path.exists(drv) and not os.environ.get("FMD_BENCH_FORCE_PORT"):
        L = C.CDLL(drv)
        L.refdrv_load.restype = C.c_void_p; L.refdrv_load.argtypes = [C.c_char_p]
        L.refdrv_free.argtypes = [C.c_void_p]
        L.refdrv_bsearch.restype = C.c_double
        L.refdrv_bsearch.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        e = L.refdrv_load(fmd_path.encode())
        assert e, "reference rld_restore failed on the .fmd written by the product"
        n1 = min(n, 50_000)
        t1 = L.refdrv_bsearch(e, n1, q.shape[1], q.ctypes.data, cnt.ctypes.data, beg.ctypes.data, end.ctypes.data, 1)
        tall = L.refdrv_bsearch(e, n, q.shape[1], q.ctypes.data, cnt.ctypes.data, beg.ctypes.data, end.ctypes.data, cores)
        L.refdrv_free(e)
        kind = "reference"
        rate1 = n1 / t1
    else:
        import orcbind
        o = orcbind.OrcIndex(fmd_path)
        t0 = time.time()
        cnt, beg, end = o.backward_search(q, n_threads=cores)
        tall = time.time() - t0
        n1 = min(n, 50_000)
        t0 = time.time(); o.backward_search(q[:n1], n_threads=1); rate1 = n1 / (time.time() - t0)
        o.close()
        kind = "port"
    hit = cnt > 0
    parity = bool(np.array_equal(cnt, gpu_cnt[:n]) and np.array_equal(beg[hit], gpu_beg[:n][hit]) and np.array_equal(end[hit], gpu_end[:n][hit]))
    return {"value": n / tall, "unit": "reads/s", "cores": cores, "kind": kind,
            "sample": "first %d of the batch, all %d host threads (1 thread: %.0f reads/s)" % (n, cores, rate1)}, parity


def rank_queries_per_read(reads_host, fmd_path, sample=20000):
    """Algorithmic rank queries per read, counted by the instrumented CPU restatement on a
    sample of the same input.  For hits this is the closed form 2*(len-1)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import orcbind
    o = orcbind.OrcIndex(fmd_path)
    o.counters()
    n = min(sample, len(reads_host))
    o.backward_search(reads_host[:n])
    c = o.counters()
    o.close()
    # rank1a(-1) is free in the reference too (rld.c:428-431) and not counted
    return (c["rank1a"] + c["rank2a"] + c["rank2a_spill"]) / n


REF_OVLP_DT = np.dtype([("rank", "<u8"), ("k0", "<u8"), ("k1", "<u8"), ("len", "<i4"), ("status", "<i4"), ("n_ovlp", "<i4"),
                        ("rbeg", "<i4"), ("ext_len", "<i4"), ("n_nei", "<i4"), ("nei", "<u8", (4, 3))])  # oracle/ref_driver.c


def overlap_cpu_baseline(fmd_path, ids, min_match, g_rec, g_nei):
    """fm_retrieve + fm6_is_contained + fm6_get_nei per sequence id on the host cores (the
    reference itself when oracle/_ref travelled, else our C port), and the parity check."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    cores = os.cpu_count() or 1
    n = len(ids)
    ids = np.ascontiguousarray(ids, dtype=np.uint64)
    drv = os.path.join(ROOT, "oracle", "_ref", "libref_driver.so")
    if os.path.exists(drv) and not os.environ.get("FMD_BENCH_FORCE_PORT"):
        L = C.CDLL(drv)
        L.refdrv_load.restype = C.c_void_p; L.refdrv_load.argtypes = [C.c_char_p]
        L.refdrv_free.argtypes = [C.c_void_p]
        L.refdrv_overlap.restype = C.c_double
        L.refdrv_overlap.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        e = L.refdrv_load(fmd_path.encode())
        assert e
        rec = np.zeros(n, dtype=REF_OVLP_DT)
        n1 = min(n, 20_000)
        t1 = L.refdrv_overlap(e, n1, ids.ctypes.data, min_match, rec.ctypes.data, 1)
        tall = L.refdrv_overlap(e, n, ids.ctypes.data, min_match, rec.ctypes.data, cores)
        L.refdrv_free(e)
        kind = "reference"
        ok = (np.array_equal(rec["rank"], g_rec["rank"]) and np.array_equal(rec["k0"], g_rec["k"][:, 0]) and
              np.array_equal(rec["k1"], g_rec["k"][:, 1]) and np.array_equal(rec["len"], g_rec["len"]) and
              np.array_equal(rec["status"], g_rec["status"]) and np.array_equal(rec["n_ovlp"], g_rec["n_ovlp"]) and
              np.array_equal(rec["rbeg"], g_rec["rbeg"]) and np.array_equal(rec["ext_len"], g_rec["ext_len"]) and
              np.array_equal(rec["n_nei"], g_rec["n_nei"]))
        for j in range(min(4, g_nei.shape[1])):
            m = rec["n_nei"] > j
            ok = ok and np.array_equal(rec["nei"][m, j, 0], g_nei["x"][m, j, 0]) and np.array_equal(rec["nei"][m, j, 1], g_nei["x"][m, j, 1]) \
                and np.array_equal(rec["nei"][m, j, 2], g_nei["info"][m, j])
    else:
        import orcbind
        o = orcbind.OrcIndex(fmd_path)
        n1 = min(n, 20_000)
        t0 = time.time(); o.overlap_batch(ids[:n1], min_match, 100, g_nei.shape[1], 1, check_left=False); t1 = time.time() - t0
        t0 = time.time(); rec, nei, _ = o.overlap_batch(ids, min_match, 100, g_nei.shape[1], cores, check_left=False); tall = time.time() - t0
        o.close()
        kind = "port"
        ok = rec.tobytes() == g_rec.tobytes() and nei.tobytes() == g_nei.tobytes()
    return {"value": n / 2.0 / tall, "unit": "reads/s", "cores": cores, "kind": kind,
            "sample": "sequence ids 0..%d (both strands of %d reads), all %d host threads (1 thread: %.0f reads/s)"
                      % (n - 1, n // 2, cores, n1 / 2.0 / t1)}, bool(ok)


def overlap_rank_queries_per_strand(fmd_path, min_match, sample=4000):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import orcbind
    o = orcbind.OrcIndex(fmd_path)
    o.counters()
    o.overlap_batch(np.arange(sample, dtype=np.uint64), min_match, 100, 4, 1, check_left=False)
    c = o.counters()
    o.close()
    return c, (c["rank1a"] + c["rank2a"] + c["rank2a_spill"]) / sample


def bench_overlap(torch, api, index, dev, n_reads, L, steps, warmup, dist, world, rank, fmd_path):
    """Overlap discovery for ALL 2N sequence ids: one step = retrieve +
    is_contained + get_nei for every strand, in HBM-bounded batches."""
    min_match = int(os.environ.get("FMD_BENCH_MINMATCH", "50"))
    n_ids = 2 * n_reads
    # strands per launch: the HBM work area is 6.4 kB per strand (two candidate lists of 100 entries); 2*10^7
    # strands = 128 GB of the 288 GB, fewer kernel tails than small batches (4 M: +4 % time)
    batch = min(n_ids, int(os.environ.get("FMD_BENCH_OVLP_BATCH", "20000000")))
    max_nei, stride = 4, 2 * L
    ids = torch.arange(n_ids, dtype=torch.int64, device=dev)
    rec = torch.zeros(n_ids * 64, dtype=torch.uint8, device=dev)
    nei = torch.zeros(n_ids * max_nei * 32, dtype=torch.uint8, device=dev)
    seq = torch.zeros(n_ids * stride, dtype=torch.uint8, device=dev)
    wb = api.lib().fmd_ovlp_work_bytes(batch, L, min_match)
    work = torch.empty(wb, dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream()
    sh = C.c_void_p(stream.cuda_stream)

    def step():
        for o in range(0, n_ids, batch):
            c = min(batch, n_ids - o)
            api.check(api.lib().fmd_ovlp_dev(index.h, sh, c, ids.data_ptr() + o * 8, min_match, L, max_nei,
                                             rec.data_ptr() + o * 64, nei.data_ptr() + o * max_nei * 32,
                                             seq.data_ptr() + o * stride, stride, work.data_ptr(), wb))
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    w0 = time.perf_counter()
    for a, b in evs:
        a.record(stream); step(); b.record(stream)
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    wall = time.perf_counter() - w0
    if dist:
        t = torch.tensor([wall], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())
    # ---- N > 1, FMD_BENCH_GATHER_CHECK=1: the one exchange of the real pipeline -- gather of per-id
    # records on rank 0 over RCCL (fermi_amd/dist.py; covered on CPU by tests/test_dist_cpu.py with gloo).
    # Outside the timed region; every rank holds the same table here, so rank 0 can check what it
    # received against its own copy.
    gather_note = None
    if dist and os.environ.get("FMD_BENCH_GATHER_CHECK", "0") == "1":   # opt-in: a peer-to-peer exchange the driver's scaling run does not need
        try:
            from fermi_amd import dist as fdist
            m = min(n_ids, 200000)
            mine = fdist.shard_ids(m, rank, world).astype(np.int64)
            local = rec.view(torch.uint8).reshape(n_ids, 64)[torch.from_numpy(mine).to(dev)].cpu().numpy().view(api.OVLP_DT).reshape(-1)
            got = fdist.gather_rows(local, m, rank, world, dist, device=dev)
            if rank == 0:
                want = rec.view(torch.uint8).reshape(n_ids, 64)[:m].cpu().numpy().view(api.OVLP_DT).reshape(-1)
                gather_note = "ok (%d records from %d ranks, identical to rank 0's table)" % (m, world) if got.tobytes() == want.tobytes() else "MISMATCH"
        except Exception as ex:  # never let the optional check take the benchmark line down
            gather_note = "failed: %r" % (ex,)
    if rank != 0:
        return None
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in evs]))
    # Large batches run get_nei of one part beside the walk of the next on a second stream.
    # Outside the timed region: the same step in serial order into fresh buffers must give the same bytes.
    pipe_note = None
    if batch >= (1 << 21) and "FMD_OVLP_PIPE" not in os.environ and world == 1:
        keep = (rec, nei, seq)
        rec, nei, seq = torch.zeros_like(rec), torch.zeros_like(nei), torch.zeros_like(seq)
        os.environ["FMD_OVLP_PIPE"] = "1"
        try:
            step()
            torch.cuda.synchronize()
        finally:
            del os.environ["FMD_OVLP_PIPE"]
        same = torch.equal(rec, keep[0]) and torch.equal(nei, keep[1]) and torch.equal(seq, keep[2])
        pipe_note = "identical (records, neighbours, sequences of all %d strands)" % n_ids if same else "MISMATCH"
        rec, nei, seq = keep
    g_rec = rec.cpu().numpy().view(api.OVLP_DT)
    out = {"metric": "reads/sec through unitig overlap discovery (retrieve + is_contained + get_nei, both strands)",
           "value": n_reads * world * steps / wall, "unit": "reads/s", "strands_per_s": n_ids * world * steps / wall,
           "ms_per_step": wall / steps * 1e3, "min_match": min_match, "batch_strands": batch,
           "overflow_records": int(((g_rec["flags"] & api.OVLP_F_OVERFLOW) != 0).sum()),
           "contained": int((g_rec["status"] == -3).sum()), "with_neighbour": int((g_rec["n_nei"] > 0).sum())}
    if gather_note:
        out["record_gather_rccl"] = gather_note
    if pipe_note:
        out["pipelined_vs_serial_order"] = pipe_note
    if world == 1:
        cnts, qps = overlap_rank_queries_per_strand(fmd_path, min_match)
        alg = qps * BYTES_PER_RANK_QUERY * n_ids
        ach = alg / (kern_ms * 1e-3) / 1e9
        out["roofline"] = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                           "traffic": None, "kernel": "k_ovl_walk + k_ovl_seq_out + k_ovl_classify + k_ovl_nei_grp<8|12|16|21|32> + k_ovl_nei (one step = %d batches of %d strands)" % ((n_ids + batch - 1) // batch, batch),
                           "kernel_ms": kern_ms, "rank_queries_per_strand": qps, "algorithmic_bytes_per_read": 2 * qps * BYTES_PER_RANK_QUERY,
                           "oracle_counters_on_sample": cnts}
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json"))).get("overlap@%d" % n_reads)
            if pmc and min_match == 50:   # measured with 4 M-strand batches; per-step traffic does not depend on the batch size
                out["roofline"]["traffic"] = (pmc["fetch_kb"] * pmc["fetch_calibration"] + pmc["write_kb"]) * 1024.0
                out["roofline"]["traffic_source"] = pmc["source"]
        except Exception:
            pass
        ns = min(n_ids, int(os.environ.get("FMD_BENCH_CPU_SAMPLE_OVLP", "400000")))
        g_nei = nei.cpu().numpy().view(api.INTV_DT).reshape(n_ids, max_nei)
        base, ok = overlap_cpu_baseline(fmd_path, np.arange(ns, dtype=np.uint64), min_match, g_rec[:ns], g_nei[:ns])
        out["cpu_baseline"] = base
        out["parity_vs_cpu_on_sample"] = "bit-exact" if ok else "MISMATCH"
        out["speedup_vs_cpu_all_cores"] = out["value"] / base["value"]
    return out


def smem_cpu_baseline(fmd_path, reads, max_mem, g_mem, g_nmem):
    """fm6_smem (smem.c:397) per read on the host cores: the compiled reference when oracle/_ref
    travelled, else our C port; also the parity check of the GPU output on that sample."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    cores = os.cpu_count() or 1
    n, L = reads.shape
    q = np.ascontiguousarray(reads)
    INTV = np.dtype([("x", "<u8", (3,)), ("info", "<u8")])
    mem = np.zeros((n, max_mem), dtype=INTV); n_mem = np.zeros(n, dtype=np.uint32)
    drv = os.path.join(ROOT, "oracle", "_ref", "libref_driver.so")
    n1 = min(n, 10_000)
    if os.path.exists(drv) and not os.environ.get("FMD_BENCH_FORCE_PORT"):
        Lb = C.CDLL(drv)
        Lb.refdrv_load.restype = C.c_void_p; Lb.refdrv_load.argtypes = [C.c_char_p]
        Lb.refdrv_free.argtypes = [C.c_void_p]
        Lb.refdrv_smem.restype = C.c_double
        Lb.refdrv_smem.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_int, C.c_uint32, C.c_void_p, C.c_void_p, C.c_int]
        e = Lb.refdrv_load(fmd_path.encode())
        assert e
        t1 = Lb.refdrv_smem(e, n1, L, q.ctypes.data, 0, max_mem, mem.ctypes.data, n_mem.ctypes.data, 1)
        tall = Lb.refdrv_smem(e, n, L, q.ctypes.data, 0, max_mem, mem.ctypes.data, n_mem.ctypes.data, cores)
        Lb.refdrv_free(e)
        kind = "reference"
    else:
        import orcbind
        o = orcbind.OrcIndex(fmd_path)
        t0 = time.time(); o.smem_batch(q[:n1], 0, max_mem, 1); t1 = time.time() - t0
        t0 = time.time(); mem, n_mem = o.smem_batch(q, 0, max_mem, cores); tall = time.time() - t0
        o.close()
        kind = "port"
    ok = np.array_equal(n_mem, g_nmem)
    if ok:
        for j in range(max_mem):
            m = n_mem > j
            ok = ok and mem[m, j].tobytes() == g_mem[m, j].tobytes()
    return {"value": n / tall, "unit": "reads/s", "cores": cores, "kind": kind,
            "sample": "first %d reads of the batch, all %d host threads (1 thread: %.0f reads/s)" % (n, cores, n1 / t1)}, bool(ok)


def bench_raw_reads(torch, api, workload, dev, local_rank, n_reads, L, steps, warmup, dist, world, rank):
    """the index of reads that carry 1 % substitutions (what `fermi exact`
    and `fermi correct` see before error correction).  Built once, used by the SMEM leg and the
    k-mer harvest leg."""
    err = float(os.environ.get("FMD_BENCH_SMEM_ERR", "0.01"))
    t0 = time.time()
    reads = workload.synth_reads_host(n_reads, L, 30, err)
    rd = workload.ReadsOnDevice(reads, dev)
    d_bwt, n_sym = workload.build_bwt_on_device(rd, local_rank)
    torch.cuda.synchronize()
    fmd_path = None
    if rank == 0 and world == 1:
        fmd_path = os.path.join(tempfile.gettempdir(), "fmd_bench_raw_%d_%d.fmd" % (n_reads, os.getpid()))
        workload.write_fmd_from_device_bwt(d_bwt, n_sym, fmd_path, local_rank)
    index = api.DevIndex.from_bwt_dev(d_bwt, n_sym, local_rank)
    api.lib().fmd_dev_free(d_bwt)
    if rank == 0:
        log("raw-read index: %d reads at e=%g, %d symbols, %.1fs" % (n_reads, err, n_sym, time.time() - t0))
    sm = km = None
    try:
        if os.environ.get("FMD_BENCH_SMEM", "1") != "0":
            sm = bench_smem(torch, api, index, rd, reads, err, n_sym, fmd_path, dev, n_reads, L, steps, warmup, dist, world, rank)
        if os.environ.get("FMD_BENCH_KMER", "1") != "0":
            km = bench_kmer(torch, api, index, n_sym, fmd_path, dev, n_reads, steps, warmup, dist, world, rank)
    finally:
        if fmd_path and os.path.exists(fmd_path):
            os.remove(fmd_path)
        index.close()
    return sm, km


def kmer_cpu_baseline(fmd_path, w, min_occ, suf_len, n_buckets, g_trip):
    """fm6_traverse + ec_collect (correct.c:35-87) over the first n_buckets suffix buckets on the host
    cores: the reference's own static function through oracle/_ref/libref_ec.so when it travelled,
    else our C port.  Parity = identical (bucket, key, val) multisets for those buckets."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    cores = os.cpu_count() or 1
    drv = os.path.join(ROOT, "oracle", "_ref", "libref_ec.so")
    n1 = max(1, n_buckets // 128)

    def pack(B, K, V):
        return np.sort(B.astype(np.uint64) << np.uint64(40) | K.astype(np.uint64) << np.uint64(8) | V.astype(np.uint64))
    if os.path.exists(drv) and not os.environ.get("FMD_BENCH_FORCE_PORT"):
        Lb = C.CDLL(drv)
        Lb.refec_range.argtypes = [C.c_char_p] + [C.c_int] * 6 + [C.c_void_p] * 5
        Lb.refec_free.argtypes = [C.c_void_p]

        def run(b1, thr):
            pb, pk, pv, n, secs = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_uint64(), C.c_double()
            rc = Lb.refec_range(fmd_path.encode(), w, min_occ, suf_len, 0, b1, thr, C.byref(pb), C.byref(pk), C.byref(pv), C.byref(n), C.byref(secs))
            assert rc == 0
            m = n.value
            B = np.ctypeslib.as_array(C.cast(pb, C.POINTER(C.c_uint32)), (max(m, 1),))[:m].copy()
            K = np.ctypeslib.as_array(C.cast(pk, C.POINTER(C.c_uint32)), (max(m, 1),))[:m].copy()
            V = np.ctypeslib.as_array(C.cast(pv, C.POINTER(C.c_uint8)), (max(m, 1),))[:m].copy()
            for p_ in (pb, pk, pv):
                Lb.refec_free(p_)
            return B, K, V, secs.value
        kind = "reference"
    else:
        import orcbind
        o = orcbind.OrcIndex(fmd_path)

        def run(b1, thr):
            return o.ec_range(w, min_occ, suf_len, 0, b1, thr)
        kind = "port"
    B1, _, _, t1 = run(n1, 1)
    B, K, V, tall = run(n_buckets, cores)
    ok = np.array_equal(pack(B, K, V), g_trip)
    return {"value": len(B) / tall, "unit": "solid k-mers/s", "cores": cores, "kind": kind,
            "sample": "suffix buckets 0..%d of %d (%d solid k-mers), all %d host threads (1 thread: %.0f k-mers/s)"
                      % (n_buckets - 1, 1 << (2 * suf_len), len(B), cores, len(B1) / t1)}, bool(ok)


def bench_kmer(torch, api, index, n_sym, fmd_path, dev, n_reads, steps, warmup, dist, world, rank):
    """The k-mer harvest of `fermi correct` (fm6_traverse + ec_collect, correct.c:341-356) with the
    reference's automatic k (correct.c:313-319) and -O 3.  One step = the whole index."""
    import math
    w = int(os.environ.get("FMD_BENCH_KMER_W", str(min(27, int(math.log(n_sym) / math.log(4) + 8.499)))))
    min_occ, suf_len = 3, (w - 15 if w > 15 else 1)
    cap = int(os.environ.get("FMD_BENCH_KMER_CAP", str(max(1 << 22, 1 << int(math.ceil(math.log2(n_sym / 30.0 * 1.5)))))))
    lib = api.lib()
    stream = torch.cuda.current_stream()
    sh = C.c_void_p(stream.cuda_stream)
    status = torch.zeros(4, dtype=torch.int64, device=dev)
    while True:
        wb = lib.fmd_kmer_work_bytes(cap)
        work = torch.empty(wb, dtype=torch.uint8, device=dev)
        ob = torch.empty(cap, dtype=torch.int32, device=dev); ok_ = torch.empty(cap, dtype=torch.int32, device=dev)
        ov = torch.empty(cap, dtype=torch.uint8, device=dev)

        def step():
            api.check(lib.fmd_kmer_collect_dev(index.h, sh, w, min_occ, suf_len, work.data_ptr(), wb, cap, ob.data_ptr(), ok_.data_ptr(), ov.data_ptr(), status.data_ptr()))
        step()
        torch.cuda.synchronize()
        st = status.cpu().numpy().view(np.uint64)
        if st[1] == 0:
            break
        del work, ob, ok_, ov
        cap *= 2
        log("k-mer harvest: frontier overflow, retrying with cap %d" % cap)
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    w0 = time.perf_counter()
    for a, b in evs:
        a.record(stream); step(); b.record(stream)
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    wall = time.perf_counter() - w0
    if dist:
        t = torch.tensor([wall], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())
    if rank != 0:
        return None
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in evs]))
    st = status.cpu().numpy().view(np.uint64)
    n_out = int(st[0])
    ctr = work[: 72 * 8].cpu().numpy().view(np.uint64)
    nodes = int(ctr[68])                     # trie nodes expanded = backward extensions (one rank2a each), counted by the kernels
    out = {"metric": "solid k-mers/sec through fm6_traverse + ec_collect (fermi correct, k=%d, -O%d)" % (w, min_occ),
           "value": n_out * world * steps / wall, "unit": "solid k-mers/s", "ms_per_step": wall / steps * 1e3,
           "solid_kmers": n_out, "informative": int(st[3]), "extensions": nodes, "extensions_per_s": nodes * world * steps / wall,
           "k": w, "suf_len": suf_len, "frontier_cap": cap}
    if world == 1:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import orcbind
        o = orcbind.OrcIndex(fmd_path)
        o.counters()
        o.ec_range(w, min_occ, suf_len, 0, 16, 1)
        cn = o.counters()
        o.close()
        spill = cn["rank2a_spill"] / max(cn["rank2a"], 1)
        alg = nodes * (1.0 + spill) * BYTES_PER_RANK_QUERY
        ach = alg / (kern_ms * 1e-3) / 1e9
        tr, src = pmc_traffic("kmer@%d" % n_reads)
        out["roofline"] = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": tr, "traffic_source": src,
                           "kernel": "k_kmer_level x %d + k_kmer_emit" % (w - 1), "kernel_ms": kern_ms,
                           "rank_queries": nodes * (1.0 + spill), "rank2a_spill_rate_on_oracle_sample": spill,
                           "oracle_counters_on_sample": cn}
        nb = min(1 << (2 * suf_len), int(os.environ.get("FMD_BENCH_CPU_SAMPLE_KMER", "8192")))
        gb = ob[:n_out].cpu().numpy().view(np.uint32); gk = ok_[:n_out].cpu().numpy().view(np.uint32); gv = ov[:n_out].cpu().numpy()
        m = gb < nb
        g_trip = np.sort(gb[m].astype(np.uint64) << np.uint64(40) | gk[m].astype(np.uint64) << np.uint64(8) | gv[m].astype(np.uint64))
        base, ok = kmer_cpu_baseline(fmd_path, w, min_occ, suf_len, nb, g_trip)
        out["cpu_baseline"] = base
        out["parity_vs_cpu_on_sample"] = "bit-exact" if ok else "MISMATCH"
        out["speedup_vs_cpu_all_cores"] = out["value"] / base["value"]
    return out


def bench_smem(torch, api, index, rd, reads, err, n_sym, fmd_path, dev, n_reads, L, steps, warmup, dist, world, rank):
    """fm6_smem (what `fermi exact` runs) of every read against the index
    of the same reads, reads carrying 1 % substitutions.  One step = all reads."""
    max_mem = 8
    batch = min(n_reads, int(os.environ.get("FMD_BENCH_SMEM_BATCH", str(n_reads))))
    mem = torch.zeros(n_reads * max_mem * 32, dtype=torch.uint8, device=dev)
    n_mem = torch.zeros(n_reads, dtype=torch.int32, device=dev)
    wb = api.lib().fmd_smem_work_bytes(batch, L)
    work = torch.empty(wb, dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream()
    sh = C.c_void_p(stream.cuda_stream)

    def step():
        for o in range(0, n_reads, batch):
            c = min(batch, n_reads - o)
            api.check(api.lib().fmd_smem_dev(index.h, sh, c, rd.flat.data_ptr(), rd.off.data_ptr() + o * 8, 0, L, max_mem,
                                             mem.data_ptr() + o * max_mem * 32, n_mem.data_ptr() + o * 4, work.data_ptr(), wb))
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    w0 = time.perf_counter()
    for a, b in evs:
        a.record(stream); step(); b.record(stream)
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    wall = time.perf_counter() - w0
    if dist:
        t = torch.tensor([wall], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())
    out = None
    if rank == 0:
        kern_ms = float(np.mean([a.elapsed_time(b) for a, b in evs]))
        g_nmem = n_mem.cpu().numpy().view(np.uint32)
        out = {"metric": "reads/sec through fm6_smem (fermi exact), reads with %g substitutions against their own index" % err,
               "value": n_reads * world * steps / wall, "unit": "reads/s", "ms_per_step": wall / steps * 1e3,
               "smems": int((g_nmem & 0x7fffffff).sum()), "overflow_reads": int((g_nmem >> 31).sum()), "index_symbols": n_sym}
        if world == 1:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import orcbind
            o = orcbind.OrcIndex(fmd_path)
            o.counters()
            ns = 4000
            o.smem_batch(reads[:ns], 0, max_mem, 1)
            cn = o.counters()
            o.close()
            qpr = (cn["rank1a"] + cn["rank2a"] + cn["rank2a_spill"]) / ns
            ach = qpr * BYTES_PER_RANK_QUERY * n_reads / (kern_ms * 1e-3) / 1e9
            tr, src = pmc_traffic("smem@%d" % n_reads) if err == 0.01 else (None, None)
            out["roofline"] = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": tr, "traffic_source": src,
                               "kernel": "k_smem", "kernel_ms": kern_ms, "rank_queries_per_read": qpr,
                               "algorithmic_bytes_per_read": qpr * BYTES_PER_RANK_QUERY, "oracle_counters_on_sample": cn}
            ns = min(n_reads, int(os.environ.get("FMD_BENCH_CPU_SAMPLE_SMEM", "400000")))
            INTV = np.dtype([("x", "<u8", (3,)), ("info", "<u8")])
            g_mem = mem[: ns * max_mem * 32].cpu().numpy().view(INTV).reshape(ns, max_mem)
            base, ok = smem_cpu_baseline(fmd_path, reads[:ns], max_mem, g_mem, g_nmem[:ns])
            out["cpu_baseline"] = base
            out["parity_vs_cpu_on_sample"] = "bit-exact" if ok else "MISMATCH"
            out["speedup_vs_cpu_all_cores"] = out["value"] / base["value"]
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    args = ap.parse_args()

    import torch
    from fermi_amd import api, workload

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        log("bench.py: --gpus %d needs torch.distributed.run with %d ranks (WORLD_SIZE=%d)" % (args.gpus, args.gpus, world))
        sys.exit(2)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    assert api.device_count() > 0, "bench.py needs a GPU: libfmdhip has no CPU fallback"

    n_reads = int(os.environ.get("FMD_BENCH_READS", "10000000"))
    err = float(os.environ.get("FMD_BENCH_ERR", "0"))
    cpu_sample = int(os.environ.get("FMD_BENCH_CPU_SAMPLE", "1000000"))
    L = 100

    # ---- untimed set-up: synthetic reads -> HBM -> GPU index build -> .fmd -> drop-in loader
    t0 = time.time()
    reads_host = workload.synth_reads_host(n_reads, L, 30, 0.0)
    rd = workload.ReadsOnDevice(reads_host, dev)
    t1 = time.time()
    d_bwt, n_sym = workload.build_bwt_on_device(rd, local_rank)
    torch.cuda.synchronize()
    t2 = time.time()
    fmd_path = None
    if rank == 0:
        fmd_path = os.path.join(tempfile.gettempdir(), "fmd_bench_%d_%d.fmd" % (n_reads, os.getpid()))
        workload.write_fmd_from_device_bwt(d_bwt, n_sym, fmd_path, local_rank)
    t3 = time.time()
    if rank == 0:   # the drop-in path: load fermi's own file format
        index = api.DevIndex.open(fmd_path, local_rank)
    else:           # other ranks replicate the same index from their own build (no file shared)
        index = api.DevIndex.from_bwt_dev(d_bwt, n_sym, local_rank)
    api.lib().fmd_dev_free(d_bwt)
    t4 = time.time()
    if rank == 0:
        log("setup: synth+upload %.1fs, GPU BWT build %.2fs (%d symbols), .fmd write %.1fs, index load+transcode %.2fs (%.2f GB in HBM)"
            % (t1 - t0, t2 - t1, n_sym, t3 - t2, t4 - t3, index.hbm_bytes / 1e9))

    # queries: rank r searches the batch rotated by r (weak scaling: same work per GPU)
    if err > 0:
        q_host = workload.synth_reads_host(n_reads, L, 30, err)
    else:
        q_host = reads_host
    if rank:
        q_host = np.roll(q_host, -(rank * (n_reads // max(world, 1))), axis=0)
    qd = workload.ReadsOnDevice(q_host, dev) if (err > 0 or rank) else rd
    cnt = torch.zeros(n_reads, dtype=torch.int64, device=dev)
    beg = torch.zeros(n_reads, dtype=torch.int64, device=dev)
    end = torch.zeros(n_reads, dtype=torch.int64, device=dev)
    stream = torch.cuda.current_stream()
    sh = C.c_void_p(stream.cuda_stream)

    def step():
        api.check(api.lib().fmd_bsearch_dev(index.h, sh, n_reads, qd.flat.data_ptr(), qd.off.data_ptr(),
                                            cnt.data_ptr(), beg.data_ptr(), end.data_ptr()))

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    w0 = time.perf_counter()
    for a, b in evs:
        a.record(stream)
        step()
        b.record(stream)
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    wall = time.perf_counter() - w0
    if dist:
        t = torch.tensor([wall], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in evs]))

    ovl = None
    if os.environ.get("FMD_BENCH_OVERLAP", "1") != "0":
        ovl = bench_overlap(torch, api, index, dev, n_reads, L, max(1, min(args.steps, 2)), min(args.warmup, 1), dist, world, rank, fmd_path)
        torch.cuda.empty_cache()   # the 128 GB work area goes back to HIP: the library allocates outside torch's cache

    sm = km = None
    if os.environ.get("FMD_BENCH_SMEM", "1") != "0" or os.environ.get("FMD_BENCH_KMER", "1") != "0":
        sm, km = bench_raw_reads(torch, api, workload, dev, local_rank, n_reads, L, max(1, min(args.steps, 2)), min(args.warmup, 1), dist, world, rank)

    if rank == 0:
        g_cnt = cnt.cpu().numpy().view(np.uint64); g_beg = beg.cpu().numpy().view(np.uint64); g_end = end.cpu().numpy().view(np.uint64)
        total_reads = n_reads * world * args.steps
        value = total_reads / wall
        out = {
            "metric": "reads/sec through FMD backward-search", "value": value, "unit": "reads/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": wall / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": "configs[1]: %dx%d bp synthetic reads (splitmix64 seed 20260928, 30x, e=%g), "
                                   "fm_backward_search of every read against the FMD index of the same reads, "
                                   "index (%.2f GB) + reads resident in HBM" % (n_reads, L, err, index.hbm_bytes / 1e9),
                       "reads_per_gpu": n_reads, "read_len": L, "index_symbols": n_sym, "parallelism": "replicated index, reads sharded x%d" % world},
            "hits": int((g_cnt > 0).sum()),
            # untimed set-up, for the record: GPU suffix-sort construction and drop-in load of fermi's file
            "index_build": {"symbols": n_sym, "gpu_bwt_seconds": t2 - t1, "symbols_per_s": n_sym / max(t2 - t1, 1e-9),
                            "fmd_write_seconds": t3 - t2, "fmd_load_transcode_seconds": t4 - t3, "hbm_bytes": index.hbm_bytes},
        }
        if world == 1:
            qpr = rank_queries_per_read(q_host, fmd_path)
            alg_bytes = qpr * BYTES_PER_RANK_QUERY * n_reads      # per launch (one launch = one step)
            achieved = alg_bytes / (kern_ms * 1e-3) / 1e9
            out["roofline"] = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                               "kernel": "k_bsearch", "kernel_ms": kern_ms, "rank_queries_per_read": qpr,
                               "algorithmic_bytes_per_read": qpr * BYTES_PER_RANK_QUERY}
            try:  # HBM bytes per launch measured by the separate rocprofv3 --pmc passes (profiles/)
                pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json"))).get("k_bsearch@%d" % n_reads)
                if pmc and err == 0:
                    out["roofline"]["traffic"] = (pmc["fetch_kb"] * pmc["fetch_calibration"] + pmc["write_kb"]) * 1024.0
                    out["roofline"]["traffic_source"] = pmc["source"]
            except Exception:
                pass
            try:  # the practical ceiling next to the spec peak: random 64-byte lines through the same gather machinery
                if os.environ.get("FMD_BENCH_PROBE", "1") != "0":
                    nl = 1 << 27
                    pms = api.probe_gather(8 << 30, 64, nl, iters=3, device=local_rank)
                    out["roofline"]["random_gather_probe"] = {"line_bytes": 64, "working_set_GiB": 8, "lines_per_s": nl / (pms * 1e-3),
                                                              "GB_per_s": nl * 64 / (pms * 1e-3) / 1e9,
                                                              "kernel_rank_queries_per_s": qpr * n_reads / (kern_ms * 1e-3)}
            except Exception:
                pass
            base, parity = cpu_baseline(fmd_path, q_host, cpu_sample, g_cnt, g_beg, g_end)
            out["cpu_baseline"] = base
            out["parity_vs_cpu_on_sample"] = "bit-exact" if parity else "MISMATCH"
            out["speedup_vs_cpu_all_cores"] = value / base["value"]
        if ovl:
            out["overlap_discovery"] = ovl
        if sm:
            out["smem"] = sm
        if km:
            out["kmer_harvest"] = km
        # `achieved` counts ALGORITHMIC bytes (the reference's accounting); where the PMC traffic
        # of the same kernels is known, say next to it what actually moved
        for leg in (out, ovl, sm, km):
            r = leg.get("roofline") if leg else None
            if r and r.get("traffic") and r.get("kernel_ms"):
                r["traffic_GBps"] = r["traffic"] / (r["kernel_ms"] * 1e-3) / 1e9
                r["traffic_frac_of_peak"] = r["traffic_GBps"] / r["peak"]
        print(json.dumps(out), flush=True)
        if fmd_path and os.path.exists(fmd_path):
            os.remove(fmd_path)
    index.close()
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
