"""bench.py legs: overlap discovery (the headline: fm_retrieve + fm6_is_contained + fm6_get_nei, unitig.c:274-300), check_left over its table
(unitig.c:186-204), and the same job on reads with errors completed inside the step."""
import ctypes as C
import numpy as np
import os
import sys
import time

from benchlegs.common import BLOCK_BYTES, BYTES_PER_RANK_QUERY, Counter, ROOT, baseline_obj, log, oracle_counters, ref_driver, roofline, timed, usable_cpus

# ------------------------------------------------------------------------------------------ overlap discovery
REF_OVLP_DT = np.dtype([("rank", "<u8"), ("k0", "<u8"), ("k1", "<u8"), ("len", "<i4"), ("status", "<i4"), ("n_ovlp", "<i4"),
                        ("rbeg", "<i4"), ("ext_len", "<i4"), ("n_nei", "<i4"), ("nei", "<u8", (4, 3))])  # oracle/ref_driver.c


def cpu_overlap(fmd_path, ids, min_match, g_rec, g_nei, keep=None):
    """fm_retrieve + fm6_is_contained + fm6_get_nei per sequence id on the host cores (the reference itself when
    oracle/_ref travelled, else our C port), and the parity check (over the rows of `keep` when given: rows that exceeded a
    capacity carry FMD_OVLP_F_OVERFLOW instead of a result and are re-run larger by the caller)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    cores = usable_cpus()
    n = len(ids)
    ids = np.ascontiguousarray(ids, dtype=np.uint64)
    n1 = min(n, 20_000)
    L = ref_driver()
    if L:
        e = L.refdrv_load(fmd_path.encode())
        assert e
        rec = np.zeros(n, dtype=REF_OVLP_DT)
        t1 = L.refdrv_overlap(e, n1, ids.ctypes.data, min_match, rec.ctypes.data, 1)
        tall = L.refdrv_overlap(e, n, ids.ctypes.data, min_match, rec.ctypes.data, cores)
        L.refdrv_free(e)
        kind = "reference"
        if keep is not None:
            rec, g_rec, g_nei = rec[keep], g_rec[keep], g_nei[keep]
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
        t0 = time.time(); o.overlap_batch(ids[:n1], min_match, 100, g_nei.shape[1], 1, check_left=False); t1 = time.time() - t0
        t0 = time.time(); rec, nei, _ = o.overlap_batch(ids, min_match, 100, g_nei.shape[1], cores, check_left=False); tall = time.time() - t0
        o.close()
        kind = "port"
        if keep is not None:
            rec, nei, g_rec, g_nei = rec[keep], nei[keep], g_rec[keep], g_nei[keep]
        g2 = g_rec.copy(); g2["reserved"] = rec["reserved"]; g2["lfork"] = rec["lfork"]
        ok = rec.tobytes() == g2.tobytes() and nei.tobytes() == g_nei.tobytes()
    return baseline_obj(n / 2.0 / tall, "reads/s", cores, kind,
                        "a random sample of %d sequence ids (read-strands), %d pinned host threads" % (n, cores), n1 / 2.0 / t1), bool(ok)


class OverlapJob:
    """Overlap discovery of this rank's shard of the sequence ids, buffers resident in HBM."""

    def __init__(self, torch, api, index, dev, n_ids, rank, world, L, min_match):
        self.torch, self.api, self.index, self.dev = torch, api, index, dev
        self.min_match, self.L, self.max_nei, self.stride = min_match, L, 4, 2 * L
        self.n_ids, self.rank, self.world = n_ids, rank, world
        self.ids = torch.arange(rank, n_ids, world, dtype=torch.int64, device=dev)   # start/step interleave (unitig.c:333)
        self.n = int(self.ids.numel())
        # strands per launch: the HBM work area is 3.95 kB per strand at 100 bp, -l50 (two candidate lists of 58 entries, the stash,
        # the work lists); 2*10^7 strands = 79 GB of the 288 GB
        self.batch = max(1, min(self.n, int(os.environ.get("FMD_BENCH_OVLP_BATCH", "20000000"))))
        if "FMD_BENCH_OVLP_BATCH" not in os.environ:   # beside a large index (config 5: 141 GB) the rows of the shard and the job's work area must still fit
            rows = self.n * (64 + self.max_nei * 32 + self.stride)
            free_b = torch.cuda.mem_get_info(dev)[0] + torch.cuda.memory_reserved(dev) - torch.cuda.memory_allocated(dev)
            while self.batch > 1_000_000 and rows + api.lib().fmd_ovlp_sorted_work_bytes(self.n, self.batch, L, min_match) + (6 << 30) > free_b:
                self.batch //= 2
        self.rec = torch.zeros(self.n * 64, dtype=torch.uint8, device=dev)
        self.nei = torch.zeros(self.n * self.max_nei * 32, dtype=torch.uint8, device=dev)
        self.seq = torch.zeros(self.n * self.stride, dtype=torch.uint8, device=dev)
        # the whole shard is ONE job (fmd_ovlp_sorted_dev): every strand 32 bases in, the strands sorted by the minimizer of those
        # bases, the rest batch by batch in that order; the work area holds the parked strands (64 B each), the sort arrays and the
        # work area of one batch
        self.wb = api.lib().fmd_ovlp_sorted_work_bytes(self.n, self.batch, L, min_match)
        self.work = torch.empty(self.wb, dtype=torch.uint8, device=dev)
        self.stream = torch.cuda.current_stream()
        self.sh = C.c_void_p(self.stream.cuda_stream)
        self.packed = None
        self.gatherer = None

    def compute(self, Lb=None, h=None):
        Lb = Lb or self.api.lib()
        h = h or self.index.h
        self.api.check(Lb.fmd_ovlp_sorted_dev(h, self.sh, self.n, self.ids.data_ptr(), self.min_match, self.L, self.max_nei,
                                              self.rec.data_ptr(), self.nei.data_ptr(), self.seq.data_ptr(), self.stride,
                                              self.work.data_ptr(), self.wb, self.batch))

    def compute_in_id_order(self, Lb=None, h=None):
        """The same strands batch by batch in id order through the one-pass walk (fmd_ovlp_dev): rounds 1-2's step, kept as the A/B."""
        Lb = Lb or self.api.lib()
        h = h or self.index.h
        for o in range(0, self.n, self.batch):
            c = min(self.batch, self.n - o)
            self.api.check(Lb.fmd_ovlp_dev(h, self.sh, c, self.ids.data_ptr() + o * 8, self.min_match, self.L, self.max_nei,
                                           self.rec.data_ptr() + o * 64, self.nei.data_ptr() + o * self.max_nei * 32,
                                           self.seq.data_ptr() + o * self.stride, self.stride, self.work.data_ptr(), self.wb))

    def check_left(self, Lb=None, h=None):
        Lb = Lb or self.api.lib()
        h = h or self.index.h
        for o in range(0, self.n, self.batch):
            c = min(self.batch, self.n - o)
            self.api.check(Lb.fmd_ovlp_check_left_dev(h, self.sh, c, self.min_match, self.L, self.rec.data_ptr() + o * 64,
                                                      self.seq.data_ptr() + o * self.stride, self.stride, self.work.data_ptr(), self.wb))

    # ---- check_left as the product runs it on one GPU: verdicts from lfork (fmd_ovlp_link_dev), the exact kernel for the rest
    def alloc_link(self):
        torch = self.torch
        self.row_of = torch.empty(self.n, dtype=torch.int32, device=self.dev)
        self.link = torch.empty(2 * self.n, dtype=torch.int32, device=self.dev)
        self.und = torch.empty(self.n, dtype=torch.int64, device=self.dev)
        self.n_und = torch.zeros(1, dtype=torch.int64, device=self.dev)
        self.rec16 = self.rec.view(self.torch.int16).view(self.n, 32)

    def check_left_linked(self, Lb=None, h=None):
        Lb = Lb or self.api.lib()
        h = h or self.index.h
        self.rec16[:, 30] = 2                                   # rec.reserved: nothing decided yet
        self.api.check(Lb.fmd_ovlp_link_dev(h, self.sh, self.n, self.rec.data_ptr(), self.nei.data_ptr(), 4 * self.max_nei,
                                            self.row_of.data_ptr(), self.link.data_ptr(), self.und.data_ptr(), self.n_und.data_ptr()))
        if int(self.n_und.item()):                              # (a host sync, as in the product: fmd_ovlp_packed_table reads the count back)
            self.check_left(Lb, h)                              # the exact kernel; it looks at the rows still at 2 only

    # ---- the one exchange (N > 1): packed rows of every rank -> rank 0, device to device
    def alloc_packed(self):
        torch, lib = self.torch, self.api.lib()
        cap = lib.fmd_ovlp_pack_max_bytes(self.n, self.max_nei, self.stride)
        self.packed = {"prec": torch.empty(self.n * 64, dtype=torch.uint8, device=self.dev),
                       "off": torch.zeros(self.n + 1, dtype=torch.int64, device=self.dev),
                       "var": torch.empty(cap, dtype=torch.uint8, device=self.dev), "cap": cap}
        assert lib.fmd_ovlp_pack_work_bytes(self.n) <= self.wb

    def pack(self):
        p = self.packed
        self.api.check(self.api.lib().fmd_ovlp_pack_dev(self.index.h, self.sh, self.n, self.rec.data_ptr(), self.nei.data_ptr(), self.max_nei,
                                                        self.seq.data_ptr(), self.stride, p["prec"].data_ptr(), p["off"].data_ptr(), p["var"].data_ptr(),
                                                        p["cap"], self.work.data_ptr(), self.wb))

    def gather(self, dist):
        """-> on rank 0: list over ranks of (prec, off, var) tensors (rank 0's own first); None elsewhere.  The receive buffers belong to
        the PackedGather object: allocated in the first (warm-up) step, on the device or -- where the root's HBM cannot hold them -- in
        pinned host memory, reused afterwards."""
        from fermi_amd import dist as fdist
        if self.gatherer is None:
            self.gatherer = fdist.PackedGather(self.torch, dist, self.n_ids, self.rank, self.world, timeout_s=int(os.environ.get("FMD_BENCH_GATHER_TIMEOUT", "120")))
        p = self.packed
        return self.gatherer(p["prec"], p["off"], p["var"])


def bench_overlap(torch, api, index, dev, n_reads, L, steps, warmup, dist, world, rank, fmd_path, local_rank, legs):
    min_match = int(os.environ.get("FMD_BENCH_MINMATCH", "50"))
    n_ids = 2 * n_reads
    # ---- N > 1: the step behind the C ABI (fmd_ovlp_dist_step): pass 1 on the id shard, [the parked strands all-to-all by key,] pass 2 in
    # pieces whose rows are packed and sent to rank 0 under the compute of the next piece; transport = RCCL created through the C ABI
    # (fmd_comm_rccl_*), or -- FMD_BENCH_BACKEND=gloo, the one-GPU test form -- torch.distributed through the fmd_comm_t callbacks.
    # FMD_BENCH_COMM=torch: round 3's step (compute, pack, ONE gather through torch.distributed), kept as the fallback.
    # Key shard from four ranks up: at N = 2 every parked strand that leaves (half of them, 1.6 GB per rank at 5*10^7 reads) crosses the ONE link to
    # the peer, which costs more than the 11 % pass 2 gains (profiles/r4_scale); at N = 8 it is 0.7 GB over seven links for 22 %.
    djob = comm = None
    record_gather = None
    if world > 1 and os.environ.get("FMD_BENCH_COMM", "c") != "torch":
        from fermi_amd import dist as fdist
        ok = torch.ones(1, dtype=torch.int64, device=dev if dist.get_backend() == "nccl" else "cpu")
        try:
            comm = fdist.RcclComm(api, dist, rank, world, local_rank) if dist.get_backend() == "nccl" else fdist.TorchComm(api, dist, rank, world)
        except Exception as ex:
            log("[rank %d] no transport for the C-ABI step here (%r): falling back to the torch.distributed gather" % (rank, ex))
            ok[0] = 0
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        have_comm = bool(int(ok.item()))
        if not have_comm and comm is not None:
            comm.free()
            comm = None
        stream = torch.cuda.current_stream()
        sh = C.c_void_p(stream.cuda_stream)
        # `value` is ALWAYS the id shard's -- north_star's partitioning: ids sharded, RCCL for the final record gather only.  From four ranks up the key
        # shard (one all-to-all of the parked strands on top) is timed too, K steps of its own, and listed beside it in `shardings_timed`: it has
        # only ever been measured as an emulation on one GPU, and which of the two wins on real links is for the links to say -- not for the headline.
        shardings = [int(os.environ["FMD_BENCH_KEY_SHARD"])] if "FMD_BENCH_KEY_SHARD" in os.environ else ([0, 1] if world >= 4 else [0])
        runs = []
        for ks in (shardings if have_comm else []):
            # beside a large index (config 5: 153 GB) the job's buffers must still fit: smaller pieces until every rank has room
            batches = [int(os.environ["FMD_BENCH_OVLP_BATCH"])] if "FMD_BENCH_OVLP_BATCH" in os.environ else [0, 10_000_000, 5_000_000, 2_500_000, 1_250_000]
            for bt in batches:
                ok[0] = 1
                try:
                    djob = fdist.DistJob(api, index, comm, n_ids, min_match, L, 4, pieces=int(os.environ.get("FMD_BENCH_PIECES", "0")), key_shard=ks, root=0,
                                         host_table=int(os.environ.get("FMD_BENCH_HOST_TABLE", "-1")), batch=bt)
                except Exception as ex:
                    log("[rank %d] fmd_ovlp_dist_new (key_shard %d) with pieces of at most %s strands: %r" % (rank, ks, bt or "2*10^7", ex))
                    djob = None
                    ok[0] = 0
                dist.all_reduce(ok, op=dist.ReduceOp.MIN)
                if int(ok.item()):
                    break
                if djob:
                    djob.free()
                djob = None
                torch.cuda.empty_cache()
            if djob is None:
                continue
            stats = []
            wd = fdist.Watchdog(int(os.environ.get("FMD_BENCH_GATHER_TIMEOUT", "600")), "the N > 1 overlap step (fmd_ovlp_dist_step)")

            def step():
                with wd:
                    stats.append(djob.step(sh).as_dict())
            # one untimed step first: a transport that comes up but cannot carry the step (an error from librccl on this node's links) must cost the
            # C-ABI path, not the benchmark line -- every rank then takes the torch.distributed gather below
            ok[0] = 1
            try:
                step()
                torch.cuda.synchronize()
            except Exception as ex:
                log("[rank %d] fmd_ovlp_dist_step (key_shard %d) failed (%r)" % (rank, ks, ex))
                ok[0] = 0
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if not int(ok.item()):
                try:
                    djob.free()
                except Exception:
                    pass
                djob = None
                torch.cuda.empty_cache()
                continue
            stats.clear()
            wall, _ = timed(torch, dist, dev, stream, step, steps, warmup)
            st = {k: (float(np.mean([x[k] for x in stats[-steps:]])) if isinstance(stats[-1][k], float) else stats[-1][k]) for k in stats[-1]}
            kern_ms = st["head_ms"] + st["key_exchange_ms"] + st["tail_ms"]
            # every rank's own kernel time (HIP events on its compute stream), min / max over the ranks: who the step waits for
            km = torch.tensor([kern_ms], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
            kms = [torch.zeros_like(km) for _ in range(world)]
            dist.all_gather(kms, km)
            per_rank = [float(x.item()) for x in kms]
            gather_note = None
            if rank == 0:
                try:
                    gather_note = fdist.check_table(torch, api, index, djob, n_ids, min_match, L, 4, dev)
                except Exception as ex:   # the check must not take the benchmark line down
                    gather_note = "check failed to run: %r" % (ex,)
            runs.append({"key_shard": ks, "wall": wall, "st": st, "kern_ms": kern_ms, "per_rank": per_rank, "check": gather_note})
            djob.free()
            djob = None
            torch.cuda.empty_cache()
        rccl_ranks = api.lib().fmd_comm_rccl_count(comm.ptr()) if (comm is not None and dist.get_backend() == "nccl") else None
        if comm:
            comm.free()
            comm = None
        if runs:
            conforming = [r_ for r_ in runs if not r_["key_shard"]]
            best = conforming[0] if conforming else min(runs, key=lambda r_: r_["wall"])     # (only the key shard ran: FMD_BENCH_KEY_SHARD=1, or the id shard failed)
            wall, st, kern_ms = best["wall"], best["st"], best["kern_ms"]
            if rank != 0:
                return None, None
            # rank 0 prices its own id shard as the N = 1 line does: the same kernels over the ids 0, N, 2N, ... once more, untimed
            job = OverlapJob(torch, api, index, dev, n_ids, rank, world, L, min_match)
            job.compute()
            torch.cuda.synchronize()
            tot_rx = st["bytes_received"]
            record_gather = {"path": "fmd_ovlp_dist_step (C ABI): %d pieces, %s, table %s" % (st["pieces"], "pass 2 sharded by minimizer key (one all-to-all of the parked strands)" if st["key_shard"] else "pass 2 on the id shard",
                                                                                            "in pinned host memory" if st["on_host"] else "in rank 0's HBM"),
                             "transport": "RCCL %d through fmd_comm_rccl_* (ncclAllGather + grouped ncclSend / ncclRecv), ncclCommCount = %s" % (api.lib().fmd_comm_rccl_version(), rccl_ranks) if dist.get_backend() == "nccl" else "torch.distributed/%s through the fmd_comm_t callbacks" % dist.get_backend(),
                             "ranks_in_the_communicator": rccl_ranks if rccl_ranks is not None else world,
                             "gather_exposed_ms": st["gather_exposed_ms"], "last_piece_pack_plus_send_ms": st["last_piece_pack_send_ms"],
                             "rank0_ms": {"pass1_and_sort": st["head_ms"], "key_exchange_and_resort": st["key_exchange_ms"], "pass2_all_pieces": st["tail_ms"], "step_host_clock": st["step_ms"]},
                             "kernels_ms_per_rank": {"min": min(best["per_rank"]), "max": max(best["per_rank"]), "all": best["per_rank"]},
                             "bytes_received_by_rank0": tot_rx, "bytes_per_strand": tot_rx / max(1, n_ids - st["rows_computed"]), "check": best["check"],
                             "key_rows_sent_by_rank0": st["key_rows_sent"], "discovery_kernels_ms_per_step_on_rank0": kern_ms,
                             "headline_is": "the id shard (north_star's partitioning)" if not best["key_shard"] else "the key shard -- the id shard did not run",
                             "shardings_timed": [{"key_shard": r_["key_shard"], "ms_per_step": r_["wall"] / steps * 1e3, "reads_per_s": n_reads * steps / r_["wall"],
                                                  "kernels_ms_min_max_over_ranks": [min(r_["per_rank"]), max(r_["per_rank"])], "check": r_["check"]} for r_ in runs]}
            gathered, g_ms, gather_ms = None, None, []
    if record_gather is None:
        job = OverlapJob(torch, api, index, dev, n_ids, rank, world, L, min_match)
        gathered = [None]
        gather_ms = []
        if world > 1:
            job.alloc_packed()

        def step():
            if world > 1:
                ec0 = torch.cuda.Event(enable_timing=True); ec0.record(job.stream)
            job.compute()
            if world > 1:   # the records leave the GPU they were computed on: pack, then the RCCL gather on rank 0
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(job.stream)
                job.pack()
                gathered[0] = job.gather(dist)
                e1.record(job.stream)
                gather_ms.append((ec0, e0, e1))
        wall, kern_ms = timed(torch, dist, dev, job.stream, step, steps, warmup)
        g_ms = float(np.mean([a.elapsed_time(b) for _, a, b in gather_ms[-steps:]])) if gather_ms else None
        if gather_ms:   # the discovery kernels of this rank alone (what its roofline is priced on)
            kern_ms = float(np.mean([c.elapsed_time(a) for c, a, _ in gather_ms[-steps:]]))
    out = None

    # ---- N > 1, outside the timed region: rank 0 recomputes a sample of ids itself and compares with what arrived
    if record_gather is None and world > 1 and rank == 0:
        gather_note = None
        from fermi_amd import dist as fdist
        try:
            gather_note = fdist.check_gathered(torch, api, job, gathered[0], n_ids, world)
        except Exception as ex:   # the check must not take the benchmark line down
            gather_note = "check failed to run: %r" % (ex,)
    if rank != 0:
        return None, job
    g_rec = job.rec.cpu().numpy().view(api.OVLP_DT)
    out = {"metric": "reads/sec through unitig overlap discovery (fm_retrieve + fm6_is_contained + fm6_get_nei, both strands of every read)",
           "value": n_reads * steps / wall, "unit": "reads/s", "strands_per_s": n_ids * steps / wall,
           "ms_per_step": wall / steps * 1e3, "min_match": min_match, "batch_strands": job.batch, "strands_this_rank": job.n,
           "overflow_records": int(((g_rec["flags"] & api.OVLP_F_OVERFLOW) != 0).sum()),
           "contained": int((g_rec["status"] == -3).sum()), "with_neighbour": int((g_rec["n_nei"] > 0).sum())}
    if world > 1:
        if record_gather is not None:
            out["record_gather_rccl"] = record_gather
        else:
            tot = sum(int(t[0].numel() + t[2].numel() + t[1].numel() * 8) for t in gathered[0][1:])
            out["record_gather_rccl"] = {"ms_per_step_pack_plus_gather": g_ms, "bytes_received_by_rank0": tot, "path": job.gatherer.path,
                                         "bytes_per_strand": tot / max(1, n_ids - job.n), "check": gather_note,
                                         "discovery_kernels_ms_per_step_on_rank0": kern_ms}
        if not fmd_path:
            return out, job
    n_loc = job.n               # rows of this rank (all of them at N = 1); everything below is about rank 0's own shard
    ids_host = job.ids.cpu().numpy().astype(np.uint64)
    # ---- the same job without the two-base blocks (FMD_PAIR_USE=0: pass 1 one base per request all the way), same box, same run: time and bytes
    if os.environ.get("FMD_PAIR") == "1" and os.environ.get("FMD_BENCH_PAIR_AB", "1") != "0":
        keep_rec = job.rec.clone()
        os.environ["FMD_PAIR_USE"] = "0"
        try:
            job.compute()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(job.stream)
            for _ in range(3):
                job.compute()
            e1.record(job.stream)
            torch.cuda.synchronize()
            out["without_two_base_blocks"] = {"ms_per_step": e0.elapsed_time(e1) / 3, "what": "the same job with FMD_PAIR_USE=0 (k_ovl_walk<WALK_HEAD> takes every strand to 32 bases alone), 3 passes right after the timed steps",
                                              "same_records": bool(torch.equal(job.rec, keep_rec))}
        finally:
            del os.environ["FMD_PAIR_USE"]
        job.compute()            # (the arrays as the timed steps left them)
        torch.cuda.synchronize()
        del keep_rec
    # ---- the same strands in id order (the one-pass walk of rounds 1-2), same box, same run: time and bytes
    if os.environ.get("FMD_BENCH_ID_ORDER_AB", "1") != "0":
        keep = (job.rec, job.nei, job.seq)
        job.rec, job.nei, job.seq = torch.zeros_like(job.rec), torch.zeros_like(job.nei), torch.zeros_like(job.seq)
        job.compute_in_id_order()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(job.stream)
        for _ in range(2):
            job.compute_in_id_order()
        e1.record(job.stream)
        torch.cuda.synchronize()
        g0 = keep[0].view(torch.int32).view(job.n, 16)
        same = torch.equal(job.rec, keep[0])
        for o in range(0, job.n, 1 << 22):   # neighbours up to n_nei, sequence rows up to len + ext_len (in pieces: the masks are as large as the arrays)
            e = min(job.n, o + (1 << 22))
            nn = g0[o:e, 13].clamp(0, job.max_nei)
            km = (torch.arange(job.max_nei, device=dev)[None, :] < nn[:, None])[:, :, None]
            na, nb = keep[1].view(torch.int64).view(job.n, job.max_nei, 4)[o:e], job.nei.view(torch.int64).view(job.n, job.max_nei, 4)[o:e]
            same = same and not bool(((na != nb) & km).any())
            used = (g0[o:e, 8] + g0[o:e, 12].clamp(min=0)).clamp(0, job.stride)
            sm = torch.arange(job.stride, device=dev)[None, :] < used[:, None]
            same = same and not bool(((keep[2].view(job.n, job.stride)[o:e] != job.seq.view(job.n, job.stride)[o:e]) & sm).any())
        out["id_order_one_pass_walk"] = {"ms_per_step": e0.elapsed_time(e1) / 2, "what": "fmd_ovlp_dev batch by batch over ids in input order (the step of rounds 1-2), 2 passes on this box right after the timed steps",
                                         "same_results": "identical (records, neighbours, sequences + appended bases of all %d strands)" % n_loc if same else "MISMATCH"}
        job.rec, job.nei, job.seq = keep
    ctr = Counter(api, fmd_path, local_rank, main=job.index)
    lines = ctr.run(job.compute)
    cl_lines = None
    ctr.close()
    torch.cuda.synchronize()
    ok_rows = (g_rec["status"] == 0) & ((g_rec["flags"] & api.OVLP_F_OVERFLOW) == 0)
    n_cand = int(g_rec["n_ovlp"][ok_rows].sum())
    n_neis = int(np.minimum(g_rec["n_nei"][ok_rows], job.max_nei).sum())
    n_ext = int(g_rec["ext_len"][ok_rows].sum())
    stride_r = (L + 15) // 16 * 16
    tail2 = os.environ.get("FMD_WALK_TAIL2") != "0" and stride_r <= 112     # k_ovl_walk<WALK_TAIL2>: rows written by the walk (no stash, no k_ovl_seq_out) ...
    tail2_cls = tail2 and os.environ.get("FMD_WALK_CLS") != "0"               # ... and the work lists too (no k_ovl_classify)
    streams = {"ids": 2 * 8 * n_loc, "tail_table": 0 if os.environ.get("FMD_TAIL_TABLE") == "0" else 2 * 8 * n_loc, "stash_write_and_read": 0 if tail2 else 2 * stride_r * n_loc, "sequence_rows_out": L * n_loc + 32 * n_ext,
               "head_admission_records_write_and_read": 2 * 32 * n_loc, "parked_strands_write_read_twice": 3 * 64 * n_loc,
               "two_sorts_keys_and_rows": 2 * (2 * 8 + 4 * 2 * 8) * n_loc, "slot_to_row_map_reads": 4 * 4 * n_loc,
               "records_write_classify_read_result_write": (2 if tail2_cls else 3) * 64 * n_loc, "work_lists": 16 * n_loc,
               "candidates_write_and_read": 2 * 32 * n_cand, "classify_widest_candidate": 0 if tail2_cls else 64 * n_loc, "neighbours": 32 * n_neis}
    pair_lines = ctr.pair_lines
    if pair_lines:   # the two-base pass: every strand's parked line read and written once more
        streams["parked_strands_two_base_pass_read_write"] = 2 * 64 * n_loc
    io = sum(streams.values())
    dev_bytes = None if lines is None else (lines[0] + lines[1]) * BLOCK_BYTES + pair_lines * 128 + io
    cn = oracle_counters(fmd_path, lambda o: o.overlap_batch(np.arange(4000, dtype=np.uint64), min_match, 100, 4, 1, check_left=False))
    qps = (cn["rank1a"] + cn["rank2a"] + cn["rank2a_spill"]) / 4000.0
    out["roofline"] = roofline("k_ovl_head_adm + k_ovl_walk<HEAD> + k_ovl_park_keys + one radix sort + per batch: k_ovl_walk<%s> + k_ovl_nei_lane<G, M> (k_ovl_nei_fast<32, M>) + k_ovl_nei_grp<G> + k_ovl_nei (one step = one job of %d batches of %d strands)"
                               % ("TAIL2> (rows and work lists written by the walk" if tail2_cls else ("TAIL2> + k_ovl_classify" if tail2 else "TAIL> + k_ovl_seq_out + k_ovl_classify"),
                                  (job.n + job.batch - 1) // job.batch, job.batch), kern_ms, dev_bytes,
                               {"rank_blocks": lines and lines[0], "prefix_table_lines": lines and lines[1], "two_base_blocks_128B": pair_lines, "stream_bytes": io, "streams": streams},
                               qps * BYTES_PER_RANK_QUERY * n_loc, "overlap@%d" % n_reads if world == 1 else "overlap@%d/%d" % (n_reads, world),
                               {"rank_queries_per_strand": qps, "oracle_counters_on_sample": cn,
                                "scope": "rank 0's shard of %d strands, its discovery kernels alone" % n_loc if world > 1 else "all %d strands" % n_loc})
    ns = min(n_loc, int(os.environ.get("FMD_BENCH_CPU_SAMPLE_OVLP", "400000")))
    sel = np.sort(np.random.default_rng(2).choice(n_loc, ns, replace=False))
    sel_d = torch.from_numpy(sel).to(dev)
    g_nei_s = job.nei.view(n_loc, job.max_nei * 32)[sel_d].cpu().numpy().view(api.INTV_DT).reshape(ns, job.max_nei)
    base, ok = cpu_overlap(fmd_path, ids_host[sel], min_match, g_rec[sel], g_nei_s)
    out["cpu_baseline"] = base
    out["parity_vs_cpu_on_sample"] = "bit-exact" if ok else "MISMATCH"
    out["speedup_vs_cpu_all_cores"] = out["value"] / base["value"]
    out["_check_left_lines"] = cl_lines
    # the reference's per-read functions hand their results to host code: the same discovery through the host form the CLI uses
    # (fmd_ovlp_packed_batch: chunks of 2^22 rows computed, packed and copied to host memory, copy of one chunk under the compute
    # of the next), wall clock, one batch.  Reported beside `value`, never as `value`.
    if os.environ.get("FMD_BENCH_HOST_API", "1") != "0" and world == 1:
        try:
            nb = min(n_ids, 20_000_000)
            h_rec = np.zeros(nb, dtype=api.OVLP_DT); h_off = np.zeros(nb, dtype=np.uint64)
            shift = 22
            nch = (nb + (1 << shift) - 1) >> shift
            chunks = (C.c_void_p * nch)()
            lib = api.lib()
            best = None
            for _ in range(2):
                t0 = time.time()
                api.check(lib.fmd_ovlp_packed_batch(index.h, None, 0, 1, nb, min_match, L, job.max_nei, 0, h_rec.ctypes.data, h_off.ctypes.data, shift, chunks))
                dt = time.time() - t0
                best = dt if best is None else min(best, dt)
                lib.fmd_ovlp_packed_free(chunks, nch)
            same = bool(np.array_equal(h_rec["rbeg"], g_rec["rbeg"][:nb]) and np.array_equal(h_rec["n_nei"], g_rec["n_nei"][:nb]) and np.array_equal(h_rec["k"], g_rec["k"][:nb]))
            out["host_table_pcie_inclusive"] = {"value": nb / 2 / best, "unit": "reads/s", "strands_per_s": nb / best, "ms": best * 1e3, "strands": nb,
                                                "records_equal_to_resident_results": same,
                                                "what": "fmd_ovlp_packed_batch: ids 0..%d, packed rows (record + neighbours + 2-bit bases) in host memory, best of 2" % (nb - 1)}
        except Exception as ex:
            out["host_table_pcie_inclusive"] = {"error": repr(ex)}
    return out, job


def bench_check_left(torch, api, job, n_reads, steps, warmup, fmd_path, ovl, local_rank):
    """check_left_simple (unitig.c:186-204) for every strand with a unique neighbour, as the product computes it on one GPU:
    the verdict of almost every edge follows from the lfork field the discovery kernels already wrote for the neighbour's
    reverse strand (fmd_ovlp_link_dev: two streaming kernels, which also build the walk's row map and links); the exact
    kernel (fmd_ovlp_check_left_dev) runs on the edges that field leaves open.  Timed against the discovery it follows;
    the exact kernel on EVERY edge (what round 1 shipped) is timed once beside it."""
    ovl.pop("_check_left_lines", None)
    dev, stream = job.dev, job.stream
    job.alloc_link()
    wall, kern_ms = timed(torch, None, dev, stream, job.check_left_linked, steps, warmup)
    n_und = int(job.n_und.item())
    g_rec = job.rec.cpu().numpy().view(api.OVLP_DT)
    n_edges = int(((g_rec["status"] == 0) & (g_rec["n_nei"] == 1) & (g_rec["rbeg"] >= 0)).sum())
    out = {"metric": "read-strands/sec through check_left_simple (unitig.c:186-204) over a finished overlap table: lfork verdicts + row map + links "
                     "(fmd_ovlp_link_dev), exact kernel on the undecided edges",
           "value": job.n * steps / wall, "unit": "strands/s", "ms_per_step": wall / steps * 1e3, "edges_checked": n_edges,
           "edges_left_to_the_exact_kernel": n_und, "back_bifurcations": int((g_rec["reserved"] == 1).sum()),
           "fraction_of_discovery_time": (wall / steps * 1e3) / ovl["ms_per_step"]}
    ns = min(job.n, 4000)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import orcbind
    sel = np.sort(np.random.default_rng(7).choice(job.n, ns, replace=False)).astype(np.uint64)
    o = orcbind.OrcIndex(fmd_path)
    o.counters(); o.overlap_batch(sel, job.min_match, 100, 4, 1, check_left=False); c0 = o.counters()
    rec_o, _, _ = o.overlap_batch(sel, job.min_match, 100, 4, 1, check_left=True); c1 = o.counters()
    o.close()
    q = {k: c1[k] - c0[k] for k in c1}
    qps = (q["rank1a"] + q["rank2a"] + q["rank2a_spill"]) / float(ns)
    same = bool(np.array_equal(rec_o["reserved"], g_rec["reserved"][sel.astype(np.int64)]))
    out["parity_vs_oracle_on_sample"] = ("bit-exact (check_left_simple of %d random ids, %d of them edges with a verdict, %d back-bifurcations)"
                                         % (ns, int((rec_o["reserved"] != 2).sum()), int((rec_o["reserved"] == 1).sum()))) if same else "MISMATCH"
    # device bytes of the linked form: rec read twice + reserved written, neighbour x0/x1 read, row map written + read twice, links written
    io = job.n * (2 * 64 + 64 + 16 + 3 * 4 + 8) + n_und * 8
    ctr = Counter(api, fmd_path, local_rank, main=job.index)
    lines = ctr.run(job.check_left_linked)
    ctr.close()
    dev_bytes = None if lines is None else (lines[0] + lines[1]) * BLOCK_BYTES + io
    out["roofline"] = roofline("k_link_rows + k_link_edges (+ k_ovl_cls on %d undecided edges)" % n_und, kern_ms, dev_bytes,
                               {"rank_blocks": lines and lines[0], "stream_bytes": io,
                                "streams": "records read twice + verdict written, neighbour coordinates, row map scatter + two gathers, links"},
                               qps * BYTES_PER_RANK_QUERY * job.n, "check_left@%d" % n_reads,
                               {"rank_queries_per_strand_in_the_reference": qps,
                                "note": "streaming kernels: the rank work check_left_simple would redo was already done by fm6_get_nei's rounds on the neighbour's reverse strand"})
    # the round-1 form for comparison: the exact kernel on every edge
    job.rec16[:, 30] = 2
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream); job.check_left(); e1.record(stream)
    torch.cuda.synchronize()
    out["exact_kernel_on_every_edge_ms"] = e0.elapsed_time(e1)
    g2 = job.rec.cpu().numpy().view(api.OVLP_DT)
    out["lfork_verdicts_equal_exact_kernel"] = bool(np.array_equal(g2["reserved"], g_rec["reserved"]))
    return out


def bench_overlap_raw(torch, api, index, dev, n_reads, L, err, fmd_path):
    """Not a BASELINE config: overlap discovery of ALL strands of the RAW-read index (reads with errors fork; the fast get_nei path
    hands the forked strands to the general group kernels), with the fast path and without, so that the headline -- measured on
    error-free reads, where every strand takes the fast path -- can be read for what it is.  Checked against the reference on
    random ids (records + neighbours) and against the oracle's check_left_simple on random ids, where back-bifurcations exist."""
    n_ids = 2 * n_reads
    job = OverlapJob(torch, api, index, dev, n_ids, 0, 1, L, 50)
    out = {"what": "fm_retrieve + fm6_is_contained + fm6_get_nei (-l50) for all %d sequence ids of the index of %d reads with %g substitutions per base: one sorted job, then the rows that "
                   "exceeded a capacity (more than %d neighbours) again with room for 16 (64, ...) until none is left -- all inside the timed step" % (n_ids, n_reads, err, job.max_nei)}
    # the side table of the rows that do not fit: fm6_get_nei has no capacities, so the step is only complete when every row has an answer
    lib = api.lib()
    side_cap = max(1 << 16, n_ids // 50)
    side_nei_max = 64
    side = {"ids": torch.empty(side_cap, dtype=torch.int64, device=dev), "rows": torch.empty(side_cap, dtype=torch.int32, device=dev),
            "rec": torch.empty(side_cap * 64, dtype=torch.uint8, device=dev), "nei": torch.empty(side_cap * side_nei_max * 32, dtype=torch.uint8, device=dev),
            "seq": torch.empty(side_cap * job.stride, dtype=torch.uint8, device=dev)}
    side_wb = lib.fmd_ovlp_side_work_bytes(side_cap, L, 50)
    assert side_wb <= job.wb, "the side table's work area is the job's"
    side_state = {}

    def complete():
        """job.compute() + the flagged rows again, larger, until none is left; -> rows in the side table"""
        job.compute()
        ns, still = C.c_uint64(), C.c_uint64()
        nei_cap = 16
        api.check(lib.fmd_ovlp_rerun_overflow_dev(index.h, job.sh, job.n, job.ids.data_ptr(), job.rec.data_ptr(), 50, L, nei_cap, side_cap, side["ids"].data_ptr(), side["rows"].data_ptr(),
                                                  side["rec"].data_ptr(), side["nei"].data_ptr(), side["seq"].data_ptr(), job.stride, job.work.data_ptr(), job.wb, C.byref(ns), C.byref(still)))
        side_state.update(n=ns.value, nei_cap=nei_cap, still=still.value, rounds=1 if ns.value else 0)
        while side_state["still"] and nei_cap < side_nei_max:   # (a handful of rows: the whole side table once more, larger; its rows stay where they are)
            nei_cap *= 4
            api.check(lib.fmd_ovlp_dev(index.h, job.sh, ns.value, side["ids"].data_ptr(), 50, 2 * L, nei_cap, side["rec"].data_ptr(), side["nei"].data_ptr(), side["seq"].data_ptr(), job.stride,
                                       job.work.data_ptr(), job.wb))
            torch.cuda.synchronize()
            fl = side["rec"][: ns.value * 64].view(torch.int32).view(-1, 16)[:, 14]
            side_state.update(nei_cap=nei_cap, still=int(((fl & api.OVLP_F_OVERFLOW) != 0).sum().item()), rounds=side_state["rounds"] + 1)
        return side_state["n"]
    saved = os.environ.get("FMD_OVLP_FAST")
    sums = {}
    try:
        for key, val in (("ms_general_group_kernels_only", "0"), ("ms_with_the_fast_get_nei_path", None)):
            if val is None:
                os.environ.pop("FMD_OVLP_FAST", None)
            else:
                os.environ["FMD_OVLP_FAST"] = val
            complete()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(job.stream)
            for _ in range(2):
                complete()
            e1.record(job.stream)
            torch.cuda.synchronize()
            out[key] = e0.elapsed_time(e1) / 2
            g = job.rec.view(torch.int32).view(job.n, 16)
            sums[key] = (int(g[:, 11:14].to(torch.int64).sum().item()), int(job.nei.view(torch.int64).view(job.n, job.max_nei, 4)[:, 0].sum().item()))
    finally:
        if saved is None:
            os.environ.pop("FMD_OVLP_FAST", None)
        else:
            os.environ["FMD_OVLP_FAST"] = saved
    g = job.rec.view(torch.int32).view(job.n, 16)
    ms = out["ms_with_the_fast_get_nei_path"]
    out["strands"] = job.n
    out["strands_per_s"] = job.n / ms * 1e3
    out["reads_per_s"] = job.n / 2 / ms * 1e3
    out["ms_per_20M_strands"] = ms * 2e7 / job.n
    out["with_neighbour"] = int((g[:, 13] > 0).sum().item())
    out["forked"] = int(((g[:, 14] & 1) != 0).sum().item())
    out["same_results_both_ways"] = len(set(sums.values())) == 1     # (sums of rbeg + ext_len + n_nei and of the first neighbours' intervals)
    # ---- the results (fast path on) against the reference on random ids
    g_rec = job.rec.cpu().numpy().view(api.OVLP_DT)
    ns = min(n_ids, int(os.environ.get("FMD_BENCH_CPU_SAMPLE_OVLP_RAW", "100000")))
    sel = np.sort(np.random.default_rng(5).choice(n_ids, ns, replace=False))
    sel_d = torch.from_numpy(sel).to(dev)
    g_nei_s = job.nei.view(n_ids, job.max_nei * 32)[sel_d].cpu().numpy().view(api.INTV_DT).reshape(ns, job.max_nei)
    # rows of the main table that were flagged have their answer in the side table (same step): take the sample's from there
    n_side = side_state["n"]
    s_rows = side["rows"][:n_side].cpu().numpy().astype(np.int64)
    s_rec = side["rec"][: n_side * 64].cpu().numpy().view(api.OVLP_DT)
    s_nei = side["nei"][: n_side * side_state["nei_cap"] * 32].cpu().numpy().view(api.INTV_DT).reshape(n_side, side_state["nei_cap"])
    flagged = np.nonzero((g_rec["flags"] & api.OVLP_F_OVERFLOW) != 0)[0]
    out["rows_completed_in_the_side_table"] = {"rows": int(n_side), "neighbour_capacity": side_state["nei_cap"], "rounds": side_state["rounds"],
                                               "are_exactly_the_flagged_rows": bool(np.array_equal(np.sort(s_rows), flagged)),
                                               "most_neighbours_of_a_strand": int(s_rec["n_nei"].max()) if n_side else 0}
    out["overflow_records"] = int(((s_rec["flags"] & api.OVLP_F_OVERFLOW) != 0).sum()) if n_side else 0     # rows WITHOUT an answer when the clock stops
    pos = np.full(n_ids, -1, dtype=np.int64); pos[s_rows] = np.arange(n_side)
    in_side = pos[sel] >= 0
    g_rec_s, g_nei_s = g_rec[sel].copy(), g_nei_s.copy()
    g_rec_s[in_side] = s_rec[pos[sel][in_side]]
    g_nei_s[in_side] = s_nei[pos[sel][in_side]][:, : job.max_nei]        # (the reference driver returns the first four neighbours of a strand and its n_nei)
    base, ok = cpu_overlap(fmd_path, sel, 50, g_rec_s, g_nei_s)
    out["cpu_baseline"] = base
    out["parity_vs_cpu_on_sample"] = "bit-exact" if ok else "MISMATCH"
    out["sample_rows_answered_from_the_side_table"] = int(in_side.sum())
    out["speedup_vs_cpu_all_cores"] = out["reads_per_s"] / base["value"]
    # ---- check_left as the product runs it (lfork verdicts, exact kernel on the open edges) against the oracle's check_left_simple
    job.alloc_link()
    job.check_left_linked()
    torch.cuda.synchronize()
    g_rec = job.rec.cpu().numpy().view(api.OVLP_DT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import orcbind
    nc = min(n_ids, 4000)
    selc = np.sort(np.random.default_rng(6).choice(n_ids, nc, replace=False)).astype(np.uint64)
    o = orcbind.OrcIndex(fmd_path)
    rec_o, _, _ = o.overlap_batch(selc, 50, 100, 4, usable_cpus(), check_left=True)
    o.close()
    same = bool(np.array_equal(rec_o["reserved"], g_rec["reserved"][selc.astype(np.int64)]))
    out["check_left"] = {"edges_left_to_the_exact_kernel": int(job.n_und.item()), "back_bifurcations": int((g_rec["reserved"] == 1).sum()),
                         "parity_vs_oracle_on_sample": ("bit-exact (check_left_simple of %d random ids: %d edges with a verdict, %d back-bifurcations among them)"
                                                        % (nc, int((rec_o["reserved"] != 2).sum()), int((rec_o["reserved"] == 1).sum()))) if same else "MISMATCH"}
    # ---- device bytes of one job (rank blocks counted by the instrumented build) over its time
    ctr = Counter(api, fmd_path, index.device, main=index)
    lines = ctr.run(job.compute)
    ctr.close()
    torch.cuda.synchronize()
    ok_rows = (g_rec["status"] == 0) & ((g_rec["flags"] & api.OVLP_F_OVERFLOW) == 0)
    n_cand = int(g_rec["n_ovlp"][ok_rows].sum())
    tail2 = os.environ.get("FMD_WALK_TAIL2") != "0"                       # (as in the headline's model: no stash, no classification pass over the records)
    tail2_cls = tail2 and os.environ.get("FMD_WALK_CLS") != "0"
    io = n_ids * (16 + 16 + (0 if tail2 else 2 * 112) + L + 3 * 64 + 16 + (0 if tail2_cls else 64 + 64) + 3 * 64 + 80 + 16) + 2 * 32 * n_cand
    dev_bytes = None if lines is None else (lines[0] + lines[1]) * BLOCK_BYTES + io
    cn = oracle_counters(fmd_path, lambda oo: oo.overlap_batch(selc[:2000], 50, 100, 4, 1, check_left=False))
    qps = (cn["rank1a"] + cn["rank2a"] + cn["rank2a_spill"]) / 2000.0
    out["roofline"] = roofline("the sorted job on reads with errors (k_ovl_nei_grp<G> takes the forked strands)", ms, dev_bytes,
                               {"rank_blocks": lines and lines[0], "prefix_table_lines": lines and lines[1], "stream_bytes": io,
                                "streams": "as the headline's model (per strand: ids, tail, admission, parked state, sort arrays, stash, records, lists) + 64 B per candidate"},
                               qps * BYTES_PER_RANK_QUERY * n_ids, "overlap_raw@%d" % n_reads, {"rank_queries_per_strand": qps})
    return out
