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
    if os.path.exists(drv):
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
            base, parity = cpu_baseline(fmd_path, q_host, cpu_sample, g_cnt, g_beg, g_end)
            out["cpu_baseline"] = base
            out["parity_vs_cpu_on_sample"] = "bit-exact" if parity else "MISMATCH"
            out["speedup_vs_cpu_all_cores"] = value / base["value"]
        print(json.dumps(out), flush=True)
        if fmd_path and os.path.exists(fmd_path):
            os.remove(fmd_path)
    index.close()
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
