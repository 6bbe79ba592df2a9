#!/usr/bin/env python3
"""bench.py -- reads/s through fermi's FMD hot path on MI355X; headline = unitig overlap discovery.

Headline (BASELINE.json north_star, configs[3] at its single-GPU size):
    50 M x 100 bp synthetic reads, overlap discovery (fm_retrieve + fm6_is_contained + fm6_get_nei,
    unitig.c:274-300) for ALL 10^8 sequence ids, min_match 50, index + reads resident in HBM.
One "step" = one pass over every sequence id.  N GPUs (one process per GPU, full index replicated in each
GPU's HBM): STRONG scaling -- the same 10^8 ids, rank r takes ids r, r+N, ... (the reference's start/step
interleave, unitig.c:333, 398-399) and the step ends with the one exchange of the pipeline: the packed
overlap records of every rank gathered device-to-device on rank 0 over RCCL (xGMI).

Further legs at N = 1 (same JSON line): check_left (unitig.c:186-204), backward search on configs[1]
(10 M reads, its own index), SMEM and the k-mer harvest of `correct` on the raw-read index (configs[2]).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

roofline.achieved / frac are MEASURED HBM bytes: rocprofv3 --pmc FETCH_SIZE (x the calibration of the run's own 64-byte
gather probe) + WRITE_SIZE, separate passes spawned by this run over two steps of the leg, over the HIP-event time of
the timed steps, over 8 TB/s (frac_basis says so; where no counter pass could run the requested-bytes figure stands
in and frac_basis says that).  achieved_requested / frac_requested = the bytes the kernels ASK for: 64 B x rank blocks
requested (counted by the instrumented build of the same sources, libfmdhip_count.so, in one extra untimed step) + the
streams the kernels read and write.  The SURVEY 8(d) accounting (128 B per rank query of the REFERENCE's layout) is
reported beside them as algorithmic_equivalent_GBps; it exceeds the peak because this layout needs far fewer bytes
per query.

Knobs: FMD_BENCH_READS (50_000_000), FMD_BENCH_BSEARCH_READS (10_000_000), FMD_BENCH_LEGS
(overlap,check_left,bsearch,smem,kmer,ecfix), FMD_BENCH_CPU_SAMPLE* (bounded CPU samples).
"""
import argparse
import ctypes as C
import json
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from benchlegs.common import PROBE, ROOT, apply_traffic, csrc_sha, log, pmc_in_run
from benchlegs.search import bench_bsearch
from benchlegs.overlap import bench_check_left, bench_overlap
from benchlegs.raw import bench_raw_reads
# names other tools and tests reach through this module (tools/ab_overlap.py, tools/scale_check.py, tests/test_ref_ecfix.py ...)
from benchlegs.common import usable_cpus, ref_driver, roofline, Counter, timed, PMC_LIVE, LEG_SOURCES  # noqa: F401
from benchlegs.search import cpu_bsearch  # noqa: F401
from benchlegs.overlap import OverlapJob, cpu_overlap, bench_overlap_raw  # noqa: F401
from benchlegs.smem import cpu_smem, bench_smem  # noqa: F401
from benchlegs.kmer import cpu_kmer, bench_kmer  # noqa: F401
from benchlegs.ecfix import cpu_ecfix, ref_ec_lib, mark_corrected, bench_ecfix, NT6_OF_ASCII  # noqa: F401

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        # launched plainly (`python bench.py --gpus N`): this process becomes the launcher -- N ranks of this same file through torch.distributed.run
        # (one per GPU, rendezvous on 127.0.0.1 at a free port), their stdout (rank 0's one JSON line) and stderr passed through, its exit code returned:
        # non-zero as soon as any rank fails (torch.distributed.run tears the others down)
        import socket
        import subprocess
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__), "--gpus", str(args.gpus), "--steps", str(args.steps), "--warmup", str(args.warmup)]
        log("bench.py: --gpus %d without a launcher: %s" % (args.gpus, " ".join(cmd[1:])))
        sys.exit(subprocess.call(cmd, env=env))
    import torch
    from fermi_amd import api, workload

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        log("bench.py: --gpus %d but the launcher started %d ranks (WORLD_SIZE)" % (args.gpus, world))
        sys.exit(2)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = os.environ.get("FMD_BENCH_BACKEND", "nccl")       # "gloo" + FMD_BENCH_SHARE_GPU=1: the N > 1 path on a one-GPU box (tests)
        if os.environ.get("FMD_BENCH_SHARE_GPU") == "1":
            local_rank = 0
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    assert api.device_count() > 0, "bench.py needs a GPU: libfmdhip has no CPU fallback"
    if dist:
        from fermi_amd import dist as fdist
        fdist.describe_fabric(torch, dist, rank, world)

    n_reads = int(os.environ.get("FMD_BENCH_READS", "50000000"))
    L = 100
    legs = os.environ.get("FMD_BENCH_LEGS", "overlap,check_left,bsearch,smem,kmer,ecfix" if world == 1 else "overlap").split(",")

    # ---- untimed set-up: synthetic reads in HBM -> GPU index build -> (rank 0, N = 1) .fmd -> drop-in loader.
    # Every rank builds the same index from the same reads: the full index is replicated, nothing is shared.
    t0 = time.time()
    rd = workload.ReadsOnDevice.synth(n_reads, L, 30, 0.0, dev)
    torch.cuda.synchronize()
    t1 = time.time()
    d_bwt, n_sym = workload.build_bwt_on_device(rd, local_rank)
    torch.cuda.synchronize()
    t2 = time.time()
    fmd_path = None
    if rank == 0:   # (N > 1 too: rank 0 prices its own shard -- roofline, CPU baseline, parity -- as the N = 1 line does)
        fmd_path = os.path.join(tempfile.gettempdir(), "fmd_bench_%d_%d.fmd" % (n_reads, os.getpid()))
        workload.write_fmd_from_device_bwt(d_bwt, n_sym, fmd_path, local_rank)
    t3 = time.time()
    if fmd_path:        # the drop-in path: load fermi's own file format
        index = api.DevIndex.open(fmd_path, local_rank)
    else:
        index = api.DevIndex.from_bwt_dev(d_bwt, n_sym, local_rank)
    api.lib().fmd_dev_free(d_bwt)
    del rd              # the overlap path needs only the index
    torch.cuda.empty_cache()
    t4 = time.time()
    # The two-base blocks (fmd_pair.hip, round 6): an auxiliary structure of the resident index, like the prefix and tail tables -- 32 bits per symbol beside the
    # index's 8, built once on the device; pass 1 of the sorted job then takes two bases per 128-byte request between depth 16 and 32 (k_ovl_pair).  They pay
    # where an index serves many passes (this benchmark; a server) and never within one pass (`fermi-amd unitig` does not build them).  FMD_BENCH_PAIRS=0: without.
    # FMD_PAIR=1 in the environment makes every OTHER handle of this run (the instrumented build's, the PMC child's) build them too, inside its first step.
    pairs_info = None
    if os.environ.get("FMD_BENCH_PAIRS", "1") != "0" and os.environ.get("FMD_PAIR", "1") != "0":
        os.environ["FMD_PAIR"] = "1"
        tp = time.time()
        before = index.hbm_bytes
        built = index.build_pairs()
        index.refresh_info()
        pairs_info = {"built": built, "build_seconds": time.time() - tp, "hbm_bytes": index.hbm_bytes - before,
                      "what": "two-base blocks: BWT[p] and BWT[LF(p)] as bit planes + 16 pair counts, 128 bytes per 32 positions; same results (tests/test_gpu_pairs.py), "
                              "the A/B without them is in overlap_discovery.without_two_base_blocks"}
        if not built:
            os.environ.pop("FMD_PAIR", None)
    if rank == 0:
        log("setup: synth in HBM %.1fs, GPU BWT build %.2fs (%d symbols), .fmd write %.1fs, index load+transcode %.2fs (%.2f GB in HBM)"
            % (t1 - t0, t2 - t1, n_sym, t3 - t2, t4 - t3, index.hbm_bytes / 1e9))

    if rank == 0 and os.environ.get("FMD_BENCH_PROBE", "1") != "0":
        try:
            nl = 1 << 27
            pms = api.probe_gather(8 << 30, 64, nl, iters=3, device=local_rank)
            PROBE.update({"line_bytes": 64, "working_set_GiB": 8, "lines_per_s": nl / (pms * 1e-3), "GB_per_s": nl * 64 / (pms * 1e-3) / 1e9})
        except Exception:
            pass
    keep_fmd = rank == 0 and world == 1 and fmd_path and os.environ.get("FMD_BENCH_PMC", "1") != "0"
    ovl, job = bench_overlap(torch, api, index, dev, n_reads, L, args.steps, args.warmup, dist, world, rank, fmd_path, local_rank, legs)
    cl = None
    if rank == 0 and world == 1 and "check_left" in legs:
        cl = bench_check_left(torch, api, job, n_reads, max(1, min(args.steps, 3)), min(args.warmup, 1), fmd_path, ovl, local_rank)
    elif ovl:
        ovl.pop("_check_left_lines", None)
    hbm_index = index.hbm_bytes
    del job
    index.close()
    torch.cuda.empty_cache()   # the 79 GB work area goes back to HIP: the library allocates outside torch's cache
    pmc_note = None
    if keep_fmd and "roofline" in (ovl or {}):   # roofline.traffic measured in this run, now that the leg's memory is free again
        pmc_note = pmc_in_run(fmd_path, n_reads)
        log("in-run PMC pass: %s" % pmc_note)
        apply_traffic(ovl["roofline"])
        if cl and "roofline" in cl:
            apply_traffic(cl["roofline"])
    # ---- what the PRODUCT does with N GPUs: `fermi-amd unitig -g 0,1,..` is ONE process with one index replica and one host thread per
    # GPU, rows of ids i = g (mod N) streamed over each GPU's own PCIe link into the slim table, host threads linking -- no RCCL.  The step timed above is
    # fmd_ovlp_dist_step (one process per GPU behind the C ABI); this is the other path's rate, host side included, on the same .fmd, once.  The ranks have
    # closed their indexes; rank 0 opens the N replicas itself while the others wait at the barrier.
    product = None
    if world > 1 or os.environ.get("FMD_BENCH_PRODUCT_TABLE") == "1":
        if rank == 0 and fmd_path and os.path.exists(fmd_path):
            try:
                from fermi_amd import hostlib
                devs = tuple(0 if os.environ.get("FMD_BENCH_SHARE_GPU") == "1" else g for g in range(world))   # (a builder's box: the replicas share GPU 0, as the ranks did)
                tb = hostlib.slim_build(fmd_path, ovl["min_match"] if ovl else 50, devs)
                product = {"path": "`fermi-amd unitig -g %s`: fmdh_slim_build -- one process, %d index replica(s) and host thread(s), rows of ids i = g (mod %d) over each GPU's own PCIe link "
                                   "into the slim table (host/slim_table.c), host threads fold and link; no RCCL" % (",".join(str(g) for g in devs), world, world),
                           "rows": tb["n_seq"], "rows_phase_seconds": tb["rows_s"], "strands_per_s_rows_phase": tb["n_seq"] / max(tb["rows_s"], 1e-9),
                           "reads_per_s_rows_phase": tb["n_seq"] / 2 / max(tb["rows_s"], 1e-9),
                           "whole_table_seconds_without_index_load": tb["build_s"] - tb["index_load_s"], "index_load_seconds_slowest_replica": tb["index_load_s"],
                           "reads_per_s_whole_table": tb["n_seq"] / 2 / max(tb["build_s"] - tb["index_load_s"], 1e-9),
                           "table_bytes_per_row": tb["table_bytes"] / max(1, tb["n_seq"]),
                           "note": "host side included (16 host threads fold the rows into 32-byte lines while they arrive); the walk itself is not part of either number"}
            except Exception as ex:
                product = {"failed": repr(ex)}
        if dist is not None and world > 1:
            dist.barrier()
    if fmd_path and os.path.exists(fmd_path):
        os.remove(fmd_path)

    bs = sm = km = raw_ovl = ec = None
    if rank == 0 and world == 1:
        k2, w2 = max(1, min(args.steps, 5)), min(args.warmup, 1)
        if "bsearch" in legs:
            bs = bench_bsearch(torch, api, workload, dev, local_rank, k2, w2)
            torch.cuda.empty_cache()
        if "smem" in legs or "kmer" in legs or "ecfix" in legs:
            sm, km, raw_ovl, ec = bench_raw_reads(torch, api, workload, dev, local_rank, n_reads, L, max(1, min(args.steps, 3)), w2, legs)

    if rank == 0:
        out = {
            "metric": "reads/sec through FMD backward-search (unitig overlap discovery)", "value": ovl["value"], "unit": "reads/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ovl["ms_per_step"],
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": "configs[3] (north_star): %dx%d bp synthetic reads (splitmix64 seed 20260928, 30x, e=0), unitig overlap discovery "
                                   "(fm_retrieve + fm6_is_contained + fm6_get_nei, -l%d) for all %d sequence ids, index (%.2f GB) replicated in each GPU's HBM, "
                                   "ids sharded i = r (mod %d)%s" % (n_reads, L, ovl["min_match"], 2 * n_reads, hbm_index / 1e9, world,
                                                                    ", packed records gathered on rank 0 over RCCL inside the step" if world > 1 else ""),
                       "reads": n_reads, "read_len": L, "index_symbols": n_sym, "parallelism": "replicated index, sequence ids sharded x%d" % world},
            "index_build": {"symbols": n_sym, "gpu_bwt_seconds": t2 - t1, "symbols_per_s": n_sym / max(t2 - t1, 1e-9),
                            "fmd_write_seconds": t3 - t2, "fmd_load_transcode_seconds": t4 - t3, "hbm_bytes": hbm_index},
            "kernel_sources_sha": csrc_sha(),
        }
        if pairs_info:
            out["index_build"]["two_base_blocks"] = pairs_info
        for k in ("roofline", "cpu_baseline", "parity_vs_cpu_on_sample", "speedup_vs_cpu_all_cores"):
            if k in ovl:
                out[k] = ovl.pop(k)
        out["overlap_discovery"] = ovl
        # the bound that applies to a path made of random 64-byte lines is the rate of those, not the streaming peak: the bare gather
        # probe of the backward-search leg (64-byte lines over 8 GiB) beside this leg's bytes
        probe = dict(PROBE) if PROBE else None
        if probe and "roofline" in out:
            r = out["roofline"]
            r["random_gather_ceiling"] = {"probe_GBps": probe["GB_per_s"], "requested_bytes_frac_of_it": (r["achieved_requested"] / probe["GB_per_s"]) if r.get("achieved_requested") else None,
                                          "traffic_frac_of_it": (r["traffic_GBps"] / probe["GB_per_s"]) if r.get("traffic_GBps") else None}
        if product:
            out["product_path_unitig_g"] = product
        if cl:
            out["check_left"] = cl
        if bs:
            out["backward_search"] = bs
        if sm:
            out["smem"] = sm
        if km:
            out["kmer_harvest"] = km
        if ec:
            out["ec_fix"] = ec
        if raw_ovl:
            out["overlap_discovery_on_raw_reads"] = raw_ovl
        if pmc_note:
            out["pmc_in_run"] = pmc_note

        def strip(o):
            if isinstance(o, dict):
                o.pop("_traffic_key", None)
                for v in o.values():
                    strip(v)
        strip(out)
        try:   # what C libraries hold in their stdio buffers (this image's librccl prints a version banner to stdout) goes out BEFORE the line, not after it
            C.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(out), flush=True)
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
