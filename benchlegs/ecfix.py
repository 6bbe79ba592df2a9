"""bench.py leg: the correction pass of `fermi correct` (ec_fix, correct.c:121-256) over the table the harvest built."""
import ctypes as C
import numpy as np
import os
import sys
import time

from benchlegs.common import ROOT, baseline_obj, roofline, timed, usable_cpus

# ------------------------------------------------------------------------------------------ the correction pass of `fermi correct`
NT6_OF_ASCII = np.full(256, 5, dtype=np.uint8)
for _ch, _v in zip(b"ACGTacgt", [1, 2, 3, 4, 1, 2, 3, 4]):
    NT6_OF_ASCII[_ch] = _v


def mark_corrected(orig_nt6, fixed_nt6, quals, info, max_corr=0.3):
    """What the reference does with a read after its two ec_fix1 passes (correct.c:247-252), on n x L arrays: corrected bases in lower case with
    quality 36, bit 16 of info when more than max_corr of the read changed or the score difference is <= 10.  -> (ASCII text, quals, info)"""
    changed = orig_nt6 != fixed_nt6
    text = np.where(changed, np.frombuffer(b"$acgtn", dtype=np.uint8)[fixed_nt6], np.frombuffer(b"$ACGTN", dtype=np.uint8)[orig_nt6])
    q = np.where(changed, np.uint8(36), quals)
    inf = info.astype(np.int64)
    inf = np.where(changed.sum(axis=1) / float(orig_nt6.shape[1]) > np.float32(max_corr).astype(np.float64), inf | (1 << 16), inf)
    inf = np.where((inf >> 18) <= 10, inf | (1 << 16), inf)
    return text, q, inf.astype(np.int32)


def ref_ec_lib():
    drv = os.path.join(ROOT, "oracle", "_ref", "libref_ec.so")
    if not os.path.exists(drv) or os.environ.get("FMD_BENCH_FORCE_PORT"):
        return None
    Lb = C.CDLL(drv)
    if not hasattr(Lb, "refec_fix"):
        return None
    Lb.refec_fix.argtypes = [C.c_int, C.c_int, C.c_int, C.c_double, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                             C.c_void_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]
    return Lb


def cpu_ecfix(w, suf_len, step, trip, reads_nt6, quals):
    """ec_fix (correct.c:222-256) of n x L reads on the host cores: the reference's own static function through oracle/_ref/libref_ec.so (its tables
    filled from `trip` = (bucket, key, val) sorted by bucket) when it travelled, else the oracle's port + the marking rule.
    -> (text, quals, info) after correct.c:247-252, rate on all cores, rate on one, look-ups per read, kind"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    cores = usable_cpus()
    n, L = reads_nt6.shape
    n1 = min(n, max(1000, n // 20))
    B, K, V = trip
    Lb = ref_ec_lib()
    if Lb:
        def run(m, thr):
            txt = np.ascontiguousarray(np.frombuffer(b"$ACGTN", dtype=np.uint8)[reads_nt6[:m]])
            q = np.ascontiguousarray(quals[:m]).copy()
            info = np.zeros(m, dtype=np.int32)
            secs, nq = C.c_double(), C.c_uint64()
            rc = Lb.refec_fix(w, suf_len, step, 0.3, len(B), B.ctypes.data, K.ctypes.data, V.ctypes.data, m, L, txt.ctypes.data, q.ctypes.data, info.ctypes.data, thr,
                              C.byref(secs), C.byref(nq))
            assert rc == 0, "refec_fix: %d" % rc
            return txt, q, info, secs.value, nq.value
        _, _, _, t1, _ = run(n1, 1)
        txt, q, info, tall, nq = run(n, cores)
        kind = "reference"
    else:
        import orcbind
        t0 = time.time(); orcbind.ec_fix(w, B, K, V, list(reads_nt6[:n1]), list(quals[:n1]), step); t1 = time.time() - t0   # (the oracle's batch form runs on one thread)
        t0 = time.time(); s, q, off, info = orcbind.ec_fix(w, B, K, V, list(reads_nt6), list(quals), step); tall = time.time() - t0
        txt, q, info = mark_corrected(reads_nt6, s.reshape(n, L), q.reshape(n, L), info)
        cores, nq, kind = 1, 0, "port"
    return (txt, q, info), n / tall, n1 / t1, nq / float(n), kind, cores


def bench_ecfix(torch, api, rd, tab, n_sym, dev, local_rank, n_reads, L, steps, warmup, raw_fmd_path):
    """configs[2], the second half of `fermi correct`: ec_fix (correct.c:121-256) of every read of the raw-read set against the solid k-mer table the
    harvest leg has just built (resident, fmd_ectab_build_dev), qualities 'I' (SURVEY 8(d)), step 5 (the CLI's default).  The kernel rewrites bases and
    qualities in place, so every step (warm-up included) gets its own copy of the reads, made before the clock starts."""
    lib = api.lib()
    stream = torch.cuda.current_stream()
    sh = C.c_void_p(stream.cuda_stream)
    step_sz = int(os.environ.get("FMD_BENCH_EC_STEP", "5"))
    trace_cap = int(os.environ.get("FMD_BENCH_EC_TRACE", "1024"))
    t = C.c_void_p()
    api.check(lib.fmd_ectab_build_dev(local_rank, sh, tab["w"], tab["suf_len"], tab["n"], tab["bucket"].data_ptr(), tab["key"].data_ptr(), tab["val"].data_ptr(), C.byref(t)))
    torch.cuda.synchronize()
    nb = n_reads * L
    ncopy = steps + warmup
    seqs = [rd.flat.clone() for _ in range(ncopy)]
    quals = [torch.full((nb + 64,), ord("I"), dtype=torch.uint8, device=dev) for _ in range(ncopy)]
    info = torch.zeros(n_reads, dtype=torch.int32, device=dev)
    wb = lib.fmd_ecfix_work_bytes(t, n_reads, trace_cap)
    work = torch.empty(wb, dtype=torch.uint8, device=dev)
    turn = [0]

    def step(Lb=None, tt=None):
        k = turn[0] % ncopy
        turn[0] += 1
        api.check((Lb or lib).fmd_ecfix_dev(tt or t, sh, n_reads, seqs[k].data_ptr(), quals[k].data_ptr(), rd.off.data_ptr(), step_sz, trace_cap, info.data_ptr(), work.data_ptr(), wb))
    try:
        wall, kern_ms = timed(torch, None, dev, stream, step, steps, warmup)
        last = (turn[0] - 1) % ncopy
        g_info = info.cpu().numpy()
        n_full = int((g_info == -2147483648).sum())
        changed = int((seqs[last][:nb] != rd.flat[:nb]).sum().item())
        out = {"metric": "reads/sec through ec_fix (the correction pass of fermi correct, correct.c:121-256), k=%d, step %d, reads with 1 %% substitutions, quality 'I'" % (tab["w"], step_sz),
               "value": n_reads * steps / wall, "unit": "reads/s", "ms_per_step": wall / steps * 1e3, "solid_kmers_in_the_table": tab["n"], "bases_changed": changed,
               "reads_flagged_unfixable_by_the_kernel_word": int(((g_info >> 16) & 1).sum()), "trace_cap": trace_cap,
               "reads_whose_trace_overflowed": n_full,    # (the host form runs these again with a longer trace: fmd_ecfix_batch)
               "table_bytes": int(8 * (1 << max(10, int(np.ceil(np.log2(max(2 * tab["n"], 1))))))), "work_bytes": int(wb)}
        # ---- device bytes of one step: what the instrumented build counts (table slots probed, queue and trace entries moved) + the read / quality / info streams
        counts = None
        Lc = api.count_lib()
        if Lc is not None:
            tc = C.c_void_p()
            if Lc.fmd_ectab_build_dev(local_rank, sh, tab["w"], tab["suf_len"], tab["n"], tab["bucket"].data_ptr(), tab["key"].data_ptr(), tab["val"].data_ptr(), C.byref(tc)) == 0:
                buf, cnt = (C.c_uint64 * 3)(), C.c_int(0)
                seqs[0].copy_(rd.flat); quals[0].fill_(ord("I")); turn[0] = 0
                Lc.fmd_ectab_line_count(tc, buf, 1, C.byref(cnt))
                step(Lc, tc)
                if Lc.fmd_ectab_line_count(tc, buf, 1, C.byref(cnt)) == 0 and cnt.value:
                    counts = [int(buf[0]), int(buf[1]), int(buf[2])]
                Lc.fmd_ectab_free(tc)
        io = n_reads * (2 * L + 8 + 4) + 2 * changed
        dev_bytes = None if counts is None else counts[0] * 8 + counts[1] * 16 + counts[2] * 8 + io
        # ---- the reference on a sample of the reads: bases, qualities and info after the marking of correct.c:247-252
        ns = min(n_reads, int(os.environ.get("FMD_BENCH_CPU_SAMPLE_ECFIX", "400000")))
        sel = np.sort(np.random.default_rng(8).choice(n_reads, ns, replace=False))
        sel_d = torch.from_numpy(sel).to(dev)
        orig = rd.flat[:nb].view(n_reads, L)[sel_d].cpu().numpy()
        g_s = seqs[last][:nb].view(n_reads, L)[sel_d].cpu().numpy()
        g_q = quals[last][:nb].view(n_reads, L)[sel_d].cpu().numpy()
        keep = g_info[sel] != -2147483648
        g_txt, g_q2, g_inf = mark_corrected(orig, g_s, g_q, g_info[sel])
        order = torch.argsort(tab["bucket"][: tab["n"]].to(torch.int64), stable=True)
        trip = (tab["bucket"][: tab["n"]][order].cpu().numpy().view(np.uint32), tab["key"][: tab["n"]][order].cpu().numpy().view(np.uint32), tab["val"][: tab["n"]][order].cpu().numpy())
        del order
        (r_txt, r_q, r_inf), rate, rate1, lpr, kind, cores = cpu_ecfix(tab["w"], tab["suf_len"], step_sz, trip, orig, np.full((ns, L), ord("I"), dtype=np.uint8))
        ok = bool(np.array_equal(g_txt[keep], r_txt[keep]) and np.array_equal(g_q2[keep], r_q[keep]) and np.array_equal(g_inf[keep], r_inf[keep]))
        out["cpu_baseline"] = baseline_obj(rate, "reads/s", cores, kind, "a random sample of %d reads of the set against the whole table (%d solid k-mers), %d host threads: the reference's own "
                                           "ec_fix with its read k -> thread k mod n interleave" % (ns, tab["n"], cores), rate1)
        out["parity_vs_cpu_on_sample"] = ("bit-exact (bases, qualities and info words of %d reads after the marking of correct.c:247-252; %d bases corrected among them, %d reads marked bad)"
                                          % (int(keep.sum()), int((g_txt[keep] >= ord("a")).sum()), int(((g_inf[keep] >> 16) & 1).sum()))) if ok else "MISMATCH"
        out["speedup_vs_cpu_all_cores"] = out["value"] / rate
        lookups = lpr * n_reads if lpr else (counts[0] if counts else 0)
        out["roofline"] = roofline("k_ecfix", kern_ms, dev_bytes,
                                   {"table_slots_probed": counts and counts[0], "queue_entries_moved": counts and counts[1], "trace_entries_moved": counts and counts[2], "stream_bytes": io,
                                    "streams": "8 B per slot probed, 16 B per queue entry read or written, 8 B per trace entry; reads + qualities read, changed bytes written, offsets, info"},
                                   lookups * 2 * 8.0, "ecfix@%d" % n_reads,
                                   {"table_lookups_per_read_in_the_reference": lpr or None,
                                    "algorithmic_definition": "the reference's khash look-up is two dependent loads (flags word, key/value) per kh_get: 2 x 8 B x look-ups counted by the reference's own n_query on the sample; "
                                                              "SURVEY 8(d) prices rank queries and this pass makes none",
                                    "note": "one lane per read: a best-first search over <= 256 paths kept in the lane's slice of HBM; the table is one random 8-byte load per look-up"})
        return out
    finally:
        lib.fmd_ectab_free(t)
