"""bench.py leg: SMEM (fm6_smem, the loop of `fermi exact`) on reads with errors against their own index."""
import ctypes as C
import numpy as np
import os
import sys
import time

from benchlegs.common import BLOCK_BYTES, BYTES_PER_RANK_QUERY, Counter, ROOT, baseline_obj, oracle_counters, ref_driver, roofline, timed, usable_cpus

# ------------------------------------------------------------------------------------------ SMEM + k-mer harvest
def cpu_smem(fmd_path, reads, max_mem, g_mem, g_nmem):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    cores = usable_cpus()
    n, L = reads.shape
    q = np.ascontiguousarray(reads)
    INTV = np.dtype([("x", "<u8", (3,)), ("info", "<u8")])
    mem = np.zeros((n, max_mem), dtype=INTV); n_mem = np.zeros(n, dtype=np.uint32)
    n1 = min(n, 10_000)
    Lb = ref_driver()
    if Lb:
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
    return baseline_obj(n / tall, "reads/s", cores, kind, "a random sample of %d reads of the batch, %d pinned host threads" % (n, cores), n1 / t1), bool(ok)


def bench_smem(torch, api, index, rd, err, n_sym, fmd_path, dev, local_rank, n_reads, L, steps, warmup):
    """fm6_smem (what `fermi exact` runs) of every read against the index of the same reads,
    reads carrying 1 % substitutions."""
    max_mem = 8
    mem = torch.zeros(n_reads * max_mem * 32, dtype=torch.uint8, device=dev)
    n_mem = torch.zeros(n_reads, dtype=torch.int32, device=dev)
    wb = api.lib().fmd_smem_work_bytes(n_reads, L)
    work = torch.empty(wb, dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream()
    sh = C.c_void_p(stream.cuda_stream)

    def step(Lb=None, h=None):
        Lb = Lb or api.lib()
        api.check(Lb.fmd_smem_dev(h or index.h, sh, n_reads, rd.flat.data_ptr(), rd.off.data_ptr(), 0, L, max_mem,
                                  mem.data_ptr(), n_mem.data_ptr(), work.data_ptr(), wb))
    wall, kern_ms = timed(torch, None, dev, stream, step, steps, warmup)
    g_nmem = n_mem.cpu().numpy().view(np.uint32)
    n_out = int((g_nmem & 0x7fffffff).sum())
    out = {"metric": "reads/sec through fm6_smem (fermi exact), reads with %g substitutions against their own index" % err,
           "value": n_reads * steps / wall, "unit": "reads/s", "ms_per_step": wall / steps * 1e3,
           "smems": n_out, "overflow_reads": int((g_nmem >> 31).sum()), "index_symbols": n_sym}
    ctr = Counter(api, fmd_path, local_rank, main=index)
    lines = ctr.run(step)
    ctr.close()
    ns = 4000
    sel = np.sort(np.random.default_rng(3).choice(n_reads, min(n_reads, int(os.environ.get("FMD_BENCH_CPU_SAMPLE_SMEM", "400000"))), replace=False))
    sel_d = torch.from_numpy(sel).to(dev)
    q = rd.flat[: n_reads * L].view(n_reads, L)[sel_d].cpu().numpy()
    cn = oracle_counters(fmd_path, lambda o: o.smem_batch(q[:ns], 0, max_mem, 1))
    qpr = (cn["rank1a"] + cn["rank2a"] + cn["rank2a_spill"]) / float(min(ns, len(q)))
    io = n_reads * (L + 4) + n_out * 32
    # (k_smem's second counter is its candidate lists: 32-byte entries written to / picked from the lane-owned lists in HBM)
    dev_bytes = None if lines is None else lines[0] * BLOCK_BYTES + lines[1] * 32 + io
    out["roofline"] = roofline("k_smem", kern_ms, dev_bytes, {"rank_blocks": lines and lines[0], "list_entries_moved": lines and lines[1], "stream_bytes": io,
                                                              "streams": "reads + SMEM rows out; the candidate lists: 32 B per entry written or picked (counted by the instrumented build, like the rank blocks)"},
                               qpr * BYTES_PER_RANK_QUERY * n_reads, "smem@%d" % n_reads, {"rank_queries_per_read": qpr, "oracle_counters_on_sample": cn})
    INTV = np.dtype([("x", "<u8", (3,)), ("info", "<u8")])
    g_mem = mem.view(n_reads, max_mem * 32)[sel_d].cpu().numpy().view(INTV).reshape(len(sel), max_mem)
    base, ok = cpu_smem(fmd_path, q, max_mem, g_mem, g_nmem[sel])
    out["cpu_baseline"] = base
    out["parity_vs_cpu_on_sample"] = "bit-exact" if ok else "MISMATCH"
    out["speedup_vs_cpu_all_cores"] = out["value"] / base["value"]
    return out
