"""bench.py leg: backward search (fm_backward_search, exact.c:7) on configs[1], with the reference's own function on the host cores beside it."""
import ctypes as C
import numpy as np
import os
import sys
import tempfile
import time

from benchlegs.common import BLOCK_BYTES, BYTES_PER_RANK_QUERY, Counter, PROBE, ROOT, baseline_obj, log, ref_driver, roofline, timed, usable_cpus

# ------------------------------------------------------------------------------------------ backward search
def cpu_bsearch(fmd_path, q, gpu_cnt, gpu_beg, gpu_end):
    """fm_backward_search (exact.c:7) on the host cores over a bounded sample + the parity check of the GPU results."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    cores = usable_cpus()
    n = len(q)
    q = np.ascontiguousarray(q)
    cnt = np.zeros(n, dtype=np.uint64); beg = np.zeros(n, dtype=np.uint64); end = np.zeros(n, dtype=np.uint64)
    n1 = min(n, 50_000)
    L = ref_driver()
    if L:
        e = L.refdrv_load(fmd_path.encode())
        assert e, "reference rld_restore failed on the .fmd written by the product"
        t1 = L.refdrv_bsearch(e, n1, q.shape[1], q.ctypes.data, cnt.ctypes.data, beg.ctypes.data, end.ctypes.data, 1)
        tall = L.refdrv_bsearch(e, n, q.shape[1], q.ctypes.data, cnt.ctypes.data, beg.ctypes.data, end.ctypes.data, cores)
        L.refdrv_free(e)
        kind = "reference"
    else:
        import orcbind
        o = orcbind.OrcIndex(fmd_path)
        t0 = time.time(); o.backward_search(q[:n1], n_threads=1); t1 = time.time() - t0
        t0 = time.time(); cnt, beg, end = o.backward_search(q, n_threads=cores); tall = time.time() - t0
        o.close()
        kind = "port"
    hit = cnt > 0
    parity = bool(np.array_equal(cnt, gpu_cnt) and np.array_equal(beg[hit], gpu_beg[hit]) and np.array_equal(end[hit], gpu_end[hit]))
    return baseline_obj(n / tall, "reads/s", cores, kind, "a random sample of %d reads of the batch, %d pinned host threads" % (n, cores), n1 / t1), parity


def bench_bsearch(torch, api, workload, dev, local_rank, steps, warmup):
    """configs[1]: 10 M x 100 bp reads, fm_backward_search of every read against the index of the same reads."""
    n_reads = int(os.environ.get("FMD_BENCH_BSEARCH_READS", "10000000"))
    L = 100
    t0 = time.time()
    rd = workload.ReadsOnDevice.synth(n_reads, L, 30, 0.0, dev)
    d_bwt, n_sym = workload.build_bwt_on_device(rd, local_rank)
    torch.cuda.synchronize()
    t1 = time.time()
    fmd_path = os.path.join(tempfile.gettempdir(), "fmd_bench_bs_%d_%d.fmd" % (n_reads, os.getpid()))
    workload.write_fmd_from_device_bwt(d_bwt, n_sym, fmd_path, local_rank)
    api.lib().fmd_dev_free(d_bwt)
    index = api.DevIndex.open(fmd_path, local_rank)      # the drop-in path: fermi's own file format
    has_pairs = os.environ.get("FMD_PAIR") == "1" and index.build_pairs()     # (the two-base blocks, as the headline's index has them: main())
    log("backward-search index: %d reads, build %.1fs, write+load %.1fs%s" % (n_reads, t1 - t0, time.time() - t1, ", two-base blocks" if has_pairs else ""))
    cnt = torch.zeros(n_reads, dtype=torch.int64, device=dev)
    beg = torch.zeros(n_reads, dtype=torch.int64, device=dev)
    end = torch.zeros(n_reads, dtype=torch.int64, device=dev)
    stream = torch.cuda.current_stream()
    sh = C.c_void_p(stream.cuda_stream)

    def step(Lb=None, h=None):
        Lb = Lb or api.lib()
        api.check(Lb.fmd_bsearch_dev(h or index.h, sh, n_reads, rd.flat.data_ptr(), rd.off.data_ptr(), cnt.data_ptr(), beg.data_ptr(), end.data_ptr()))
    try:
        wall, kern_ms = timed(torch, None, dev, stream, step, steps, warmup)
        g_cnt = cnt.cpu().numpy().view(np.uint64)
        out = {"metric": "reads/sec through FMD backward-search (fm_backward_search, exact.c:7)", "value": n_reads * steps / wall, "unit": "reads/s",
               "ms_per_step": wall / steps * 1e3, "hits": int((g_cnt > 0).sum()),
               "config": {"workload": "configs[1]: %dx%d bp synthetic reads (splitmix64 seed 20260928, 30x, e=0) against the FMD index of the same reads (%.2f GB in HBM)"
                                      % (n_reads, L, index.hbm_bytes / 1e9), "index_symbols": n_sym}}
        if has_pairs:   # the same searches one base per request all the way (FMD_PAIR_USE=0), same box, same run
            keep = cnt.clone()
            os.environ["FMD_PAIR_USE"] = "0"
            try:
                step(); torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                for _ in range(5):
                    step()
                e1.record(stream)
                torch.cuda.synchronize()
                out["without_two_base_blocks"] = {"ms_per_step": e0.elapsed_time(e1) / 5, "same_counts": bool(torch.equal(cnt, keep))}
            finally:
                del os.environ["FMD_PAIR_USE"]
            step(); torch.cuda.synchronize()
        ctr = Counter(api, fmd_path, local_rank, main=index)
        lines = ctr.run(step)
        ctr.close()
        qpr = 2.0 * (L - 1)   # closed form for hits (exact.c:13-19), checked against the instrumented oracle in tests
        io = n_reads * (L + 24) + (2 * 24 * n_reads if has_pairs else 0)      # (+ the hand-over record written and read)
        dev_bytes = None if lines is None else (lines[0] + lines[1]) * BLOCK_BYTES + ctr.pair_lines * 128 + io
        out["roofline"] = roofline("k_bsearch" + (" + k_bsearch_pair" if has_pairs else ""), kern_ms, dev_bytes,
                                   {"rank_blocks": lines and lines[0], "prefix_table_lines": lines and lines[1], "two_base_blocks_128B": ctr.pair_lines, "stream_bytes": io,
                                    "streams": "reads %d B + 3 x 8 B results per read" % L},
                                   qpr * BYTES_PER_RANK_QUERY * n_reads, "k_bsearch@%d" % n_reads,
                                   {"rank_queries_per_read": qpr})
        if PROBE:
            out["roofline"]["random_gather_probe"] = dict(PROBE)
        ns = min(n_reads, int(os.environ.get("FMD_BENCH_CPU_SAMPLE", "1000000")))
        sel = np.sort(np.random.default_rng(1).choice(n_reads, ns, replace=False))
        sel_d = torch.from_numpy(sel).to(dev)
        q = rd.flat[: n_reads * L].view(n_reads, L)[sel_d].cpu().numpy()
        base, parity = cpu_bsearch(fmd_path, q, g_cnt[sel], beg[sel_d].cpu().numpy().view(np.uint64), end[sel_d].cpu().numpy().view(np.uint64))
        out["cpu_baseline"] = base
        out["parity_vs_cpu_on_sample"] = "bit-exact" if parity else "MISMATCH"
        out["speedup_vs_cpu_all_cores"] = out["value"] / base["value"]
        # the reference's signature takes host buffers (exact.c:7): the same reads through fmd_bsearch_batch, host arrays in
        # and out over PCIe, wall clock.  Reported beside `value`, never as `value`.
        if os.environ.get("FMD_BENCH_HOST_API", "1") != "0":
            h_flat = rd.flat[: n_reads * L].cpu().numpy()
            h_off = np.arange(n_reads + 1, dtype=np.uint64) * L
            h_out = [np.zeros(n_reads, dtype=np.uint64) for _ in range(3)]
            best = None
            for _ in range(3):
                t0 = time.time()
                api.check(api.lib().fmd_bsearch_batch(index.h, n_reads, h_flat.ctypes.data, h_off.ctypes.data, h_out[0].ctypes.data, h_out[1].ctypes.data, h_out[2].ctypes.data))
                dt = time.time() - t0
                best = dt if best is None else min(best, dt)
            out["host_buffers_pcie_inclusive"] = {"value": n_reads / best, "unit": "reads/s", "ms": best * 1e3, "bytes_over_pcie": int(n_reads * (L + 8 + 24)),
                                                  "equal_to_resident_results": bool(np.array_equal(h_out[0], g_cnt)),
                                                  "what": "fmd_bsearch_batch: pageable host arrays in (reads + offsets), three host arrays out, best of 3"}
        return out
    finally:
        index.close()
        if os.path.exists(fmd_path):
            os.remove(fmd_path)
