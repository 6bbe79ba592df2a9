"""bench.py leg: the k-mer harvest of `fermi correct` (fm6_traverse + ec_collect) and the table of the correction pass."""
import ctypes as C
import numpy as np
import os
import sys

from benchlegs.common import BLOCK_BYTES, BYTES_PER_RANK_QUERY, Counter, ROOT, baseline_obj, log, oracle_counters, roofline, timed, usable_cpus

def cpu_kmer(fmd_path, w, min_occ, suf_len, n_buckets, g_trip):
    """fm6_traverse + ec_collect (correct.c:35-87) over the first n_buckets suffix buckets on the host cores: the reference's
    own static function through oracle/_ref/libref_ec.so when it travelled, else our C port."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    cores = usable_cpus()
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
    return baseline_obj(len(B) / tall, "solid k-mers/s", cores, kind,
                        "suffix buckets 0..%d of %d (%d solid k-mers), %d host threads" % (n_buckets - 1, 1 << (2 * suf_len), len(B), cores), len(B1) / t1), bool(ok)


def bench_kmer(torch, api, index, n_sym, fmd_path, dev, local_rank, n_reads, steps, warmup):
    """The k-mer harvest of `fermi correct` (fm6_traverse + ec_collect, correct.c:341-356) with the reference's automatic k
    (correct.c:313-319) and -O 3.  One step = the whole index."""
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

        def step(Lb=None, h=None):
            Lb = Lb or lib
            api.check(Lb.fmd_kmer_collect_dev(h or index.h, sh, w, min_occ, suf_len, work.data_ptr(), wb, cap, ob.data_ptr(), ok_.data_ptr(), ov.data_ptr(), status.data_ptr()))
        step()
        torch.cuda.synchronize()
        st = status.cpu().numpy().view(np.uint64)
        if st[1] == 0:
            break
        del work, ob, ok_, ov
        cap *= 2
        log("k-mer harvest: frontier overflow, retrying with cap %d" % cap)
    wall, kern_ms = timed(torch, None, dev, stream, step, steps, warmup)
    st = status.cpu().numpy().view(np.uint64)
    n_out = int(st[0])
    ctrs = work[: 72 * 8].cpu().numpy().view(np.uint64)
    nodes = int(ctrs[68])                     # trie nodes expanded = backward extensions (one rank2a each), counted by the kernels
    out = {"metric": "solid k-mers/sec through fm6_traverse + ec_collect (fermi correct, k=%d, -O%d)" % (w, min_occ),
           "value": n_out * steps / wall, "unit": "solid k-mers/s", "ms_per_step": wall / steps * 1e3,
           "solid_kmers": n_out, "informative": int(st[3]), "extensions": nodes, "extensions_per_s": nodes * steps / wall,
           "k": w, "suf_len": suf_len, "frontier_cap": cap}
    ctr = Counter(api, fmd_path, local_rank, main=index)
    lines = ctr.run(step)
    ctr.close()
    cn = oracle_counters(fmd_path, lambda o: o.ec_range(w, min_occ, suf_len, 0, 16, 1))
    spill = cn["rank2a_spill"] / max(cn["rank2a"], 1)
    io = nodes * 64 + n_out * 9
    dev_bytes = None if lines is None else (lines[0] + lines[1]) * BLOCK_BYTES + io
    out["roofline"] = roofline("k_kmer_level x %d + k_kmer_emit" % (w - 1), kern_ms, dev_bytes,
                               {"rank_blocks": lines and lines[0], "stream_bytes": io, "streams": "32 B frontier node read + 32 B child written per extension, 9 B per triple"},
                               nodes * (1.0 + spill) * BYTES_PER_RANK_QUERY, "kmer@%d" % n_reads,
                               {"rank_queries": nodes * (1.0 + spill), "rank2a_spill_rate_on_oracle_sample": spill, "oracle_counters_on_sample": cn})
    nb = min(1 << (2 * suf_len), int(os.environ.get("FMD_BENCH_CPU_SAMPLE_KMER", "8192")))
    gb = ob[:n_out].cpu().numpy().view(np.uint32); gk = ok_[:n_out].cpu().numpy().view(np.uint32); gv = ov[:n_out].cpu().numpy()
    m = gb < nb
    g_trip = np.sort(gb[m].astype(np.uint64) << np.uint64(40) | gk[m].astype(np.uint64) << np.uint64(8) | gv[m].astype(np.uint64))
    base, ok = cpu_kmer(fmd_path, w, min_occ, suf_len, nb, g_trip)
    out["cpu_baseline"] = base
    out["parity_vs_cpu_on_sample"] = "bit-exact" if ok else "MISMATCH"
    out["speedup_vs_cpu_all_cores"] = out["value"] / base["value"]
    del work
    return out, {"w": w, "suf_len": suf_len, "n": n_out, "bucket": ob, "key": ok_, "val": ov}    # the table of the correction pass (bench_ecfix), resident
