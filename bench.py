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
import hashlib
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s HBM3E (spec)
BYTES_PER_RANK_QUERY = 128     # one reference rank block + its counts
BLOCK_BYTES = 64               # device rank block (fmd_wave.h)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


# the sources a leg's kernels are compiled from (besides the headers and the index layout, which every kernel depends on)
LEG_SOURCES = {"overlap": ("fmd_ovlp.hip", "fmd_ovlp_grp.hip", "fmd_ovlp_lane.hip", "fmd_ovlp_sort.hip"), "overlap_raw": ("fmd_ovlp.hip", "fmd_ovlp_grp.hip", "fmd_ovlp_lane.hip", "fmd_ovlp_sort.hip"),
               "check_left": ("fmd_pack.hip", "fmd_ovlp.hip"), "k_bsearch": ("fmd_ops.hip",), "smem": ("fmd_smem.hip",), "kmer": ("fmd_kmer.hip",), "ecfix": ("fmd_ecfix.hip",)}


def csrc_sha(leg=None, read=None):
    """Identity of the kernel sources PMC figures are valid for: the headers, the index layout and the files the leg's kernels
    live in (all of fermi_amd/csrc when no leg is named).  `read(name) -> bytes` lets a tool hash another revision's files."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "fermi_amd", "csrc")
    read = read or (lambda fn: open(os.path.join(d, fn), "rb").read())
    for fn in sorted(os.listdir(d)):
        if fn.endswith(".h") or fn == "fmd_index.hip" or (fn.endswith(".hip") and (leg is None or fn in LEG_SOURCES[leg])):
            h.update(fn.encode()); h.update(read(fn))
    return h.hexdigest()[:16]


PMC_LIVE = {}     # leg key -> (bytes per step, source): measured by THIS run (pmc_in_run), preferred over the look-up below
PROBE = {}        # the bare random-gather probe of this run (64-byte lines over 8 GiB): the ceiling that applies to a path made of random lines


def pmc_in_run(fmd_path, n_reads, steps=2, leg="overlap"):
    """roofline.traffic measured in the run that prints it: when rocprofv3 is on the box, two short child processes run `steps` steps of
    the headline leg (tools/pmc_legs.py on the .fmd this run wrote) under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate
    passes, counters only, as MI355X_MICROARCH.md prescribes), a third runs the gather probe for the FETCH_SIZE calibration (known byte
    count, 64-byte lines).  -> PMC_LIVE["overlap@n"], PMC_LIVE["check_left@n"] (leg "overlap") or PMC_LIVE["ecfix@n"] (leg "ecfix": k_ecfix over the
    raw-read set, table harvested by the child from the raw .fmd); on any failure the tracked look-up stays in place."""
    import csv, glob, shutil, subprocess
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if not exe or os.environ.get("FMD_BENCH_PMC", "1") == "0":
        return "not run (%s)" % ("FMD_BENCH_PMC=0" if exe else "no rocprofv3 on this box")
    if any(k.startswith(("ROCPROF", "ROCP_", "ROCPROFILER")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        return "not run (this process is itself being profiled)"
    t0 = time.time()
    out = tempfile.mkdtemp(prefix="fmd_pmc_")
    env = dict(os.environ, TMPDIR="/tmp", PMC_LEGS=leg, FMD_BENCH_READS=str(n_reads), PROBE_LINE="64")
    env["PMC_FMD_RAW" if leg == "ecfix" else "PMC_FMD"] = fmd_path     # (ecfix: the .fmd of the raw-read set; the child harvests its table from it)
    legs, probe = os.path.join(ROOT, "tools", "pmc_legs.py"), os.path.join(ROOT, "tools", "probe_once.py")
    try:
        for sub, ctr, script, args in (("f", "FETCH_SIZE", legs, [str(steps)]), ("w", "WRITE_SIZE", legs, [str(steps)]), ("p", "FETCH_SIZE", probe, [])):
            r = subprocess.run([exe, "--pmc", ctr, "--output-format", "csv", "-d", os.path.join(out, sub), "-o", "x", "--", sys.executable, script] + args,
                               cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=180)
            if r.returncode != 0:
                return "failed (%s pass: rc %d: %s)" % (ctr, r.returncode, r.stderr.decode(errors="replace")[-200:].replace("\n", " "))

        def sums(sub, ctr):
            acc = {}
            for f in glob.glob(os.path.join(out, sub, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if row["Counter_Name"] == ctr:
                        k = row["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
                        acc[k] = acc.get(k, 0.0) + float(row["Counter_Value"])
            return acc
        fetch, write, pr = sums("f", "FETCH_SIZE"), sums("w", "WRITE_SIZE"), sums("p", "FETCH_SIZE")
        if not pr.get("k_probe") or not fetch:
            return "failed (no counter rows)"
        cal = 2 * (1 << 27) * 64 / (pr["k_probe"] * 1024.0)      # probe_once: warm-up + one launch, 2^27 lines of 64 bytes each
        src = "measured in this run: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) over %d steps of the leg on the index this run built; KB units, FETCH_SIZE x %.4f (gather probe, 64-byte lines, same run)" % (steps, cal)
        OVL = ("k_ovl_head_adm", "k_ovl_walk", "k_ovl_pair", "k_ovl_strag_adm", "k_ovl_park_keys", "k_ovl_seq_out", "k_ovl_seq_redo", "k_ovl_classify", "k_ovl_nei_fast", "k_ovl_nei_lane", "k_ovl_nei_grp", "k_ovl_nei", "k_ovl_fix")
        legs_of = {"overlap": (("overlap@%d" % n_reads, OVL), ("check_left@%d" % n_reads, ("k_link_rows", "k_link_edges", "k_link_row_of", "k_link_rows32", "k_link_edges32", "k_ovl_cls"))), "ecfix": (("ecfix@%d" % n_reads, ("k_ecfix",)),)}
        for key, names in legs_of[leg]:
            fk = sum(v for k, v in fetch.items() if k in names) / steps
            wk = sum(v for k, v in write.items() if k in names) / steps
            if fk:
                PMC_LIVE[key] = ((fk * cal + wk) * 1024.0, src)
        return "ok (%.0f s)" % (time.time() - t0)
    except Exception as ex:
        return "failed (%r)" % (ex,)
    finally:
        shutil.rmtree(out, ignore_errors=True)


def pmc_traffic(key):
    """HBM bytes per step: measured by this run when it could (pmc_in_run), else from the separate rocprofv3 --pmc passes of the builder
    (tools/pmc_collect.sh -> profiles/pmc_traffic.json) -- None unless that entry was measured on the kernel sources of this tree."""
    if key in PMC_LIVE:
        return PMC_LIVE[key]
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json"))).get(key)
        if pmc and pmc.get("csrc_sha") == csrc_sha(key.split("@")[0]):
            return (pmc["fetch_kb"] * pmc["fetch_calibration"] + pmc["write_kb"]) * 1024.0, pmc["source"]
    except Exception:
        pass
    return None, None


def usable_cpus():
    """CPUs this process may run on: affinity mask, further bounded by a cgroup v2 quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = max(1, min(n, int(float(q) / float(p) + 0.5)))
    except Exception:
        pass
    return n


def roofline(kernel, kern_ms, device_bytes, model, alg_bytes, traffic_key, extra=None):
    """The roofline object of one leg.  device_bytes may be None (instrumented build missing)."""
    r = {"bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s", "kernel": kernel, "kernel_ms": kern_ms,
         "achieved": None, "frac": None, "frac_basis": None, "traffic": None, "traffic_source": None, "_traffic_key": traffic_key,
         "achieved_definition": "HBM bytes of one step from the PMC counters (FETCH_SIZE x the calibration of the run's own gather probe + WRITE_SIZE, separate "
                                "rocprofv3 --pmc passes) / HIP-event time of the timed steps; frac = achieved / peak.  Where no counter pass could run, the "
                                "requested-bytes figure stands in and frac_basis says so",
         "achieved_requested": None, "frac_requested": None,
         "requested_definition": "device bytes the kernels ASK for (64 B x rank blocks requested, counted by the instrumented build of the same kernels, "
                                 "+ the streams they read/write; L2 / Infinity Cache hits included) / the same time",
         "device_bytes_model": model,
         "algorithmic_equivalent_GBps": alg_bytes / (kern_ms * 1e-3) / 1e9,
         "algorithmic_definition": "SURVEY 8(d): 128 B per rank query of the reference's layout, queries counted by the instrumented oracle"}
    if device_bytes is not None:
        r["achieved_requested"] = device_bytes / (kern_ms * 1e-3) / 1e9
        r["frac_requested"] = r["achieved_requested"] / HBM_PEAK_GBS
    if extra:
        r.update(extra)
    return apply_traffic(r)


def apply_traffic(r):
    """(Re)fill the PMC fields of a roofline object from the best source there is now (the in-run pass comes after the legs it prices).
    `achieved` / `frac` are the MEASURED bytes; the requested-bytes figure has its own fields."""
    tr, src = pmc_traffic(r["_traffic_key"])
    r["traffic"], r["traffic_source"] = tr, src
    if tr:
        r["traffic_GBps"] = tr / (r["kernel_ms"] * 1e-3) / 1e9
        r["traffic_frac_of_peak"] = r["traffic_GBps"] / HBM_PEAK_GBS
        r["achieved"], r["frac"], r["frac_basis"] = r["traffic_GBps"], r["traffic_frac_of_peak"], "measured: PMC bytes / time / peak"
    else:
        r["achieved"], r["frac"] = r["achieved_requested"], r["frac_requested"]
        r["frac_basis"] = "requested bytes (no counter pass for this leg in this run)" if r["achieved"] is not None else None
    if PROBE.get("GB_per_s"):   # the ceiling of a path whose unit of work is a random 64-byte line: the bare gather probe of this run
        r["frac_of_random_gather_probe"] = {"probe_GBps": PROBE["GB_per_s"], "requested_bytes": r["achieved_requested"] / PROBE["GB_per_s"] if r["achieved_requested"] else None,
                                            "traffic": r["traffic_GBps"] / PROBE["GB_per_s"] if tr else None}
    return r


class Counter:
    """One untimed step of a leg through libfmdhip_count.so (same sources, gathers instrumented)."""

    def __init__(self, api, fmd_path, device):
        self.L = api.count_lib()
        self.h = None
        self.pair_lines = 0
        if self.L is None or not fmd_path:
            return
        h = C.c_void_p()
        if self.L.fmd_dev_open_file(device, fmd_path.encode(), C.byref(h)) == 0:
            self.h = h

    def run(self, step):
        """step(L, h) launches one step on library L / handle h; returns (rank blocks, other lines) or None."""
        if self.h is None:
            return None
        buf = (C.c_uint64 * 3)()
        cnt = C.c_int(0)
        self.L.fmd_dev_line_count3(self.h, buf, 1, C.byref(cnt))
        # The timed handle has its two-base blocks (FMD_PAIR=1); this one builds its own inside the step and must not be turned down by the "only where the
        # job still finds its room" rule of fmd_pairs_ensure, or the step counted is not the step timed: torch's cached blocks back to the device, and FMD_PAIR=2.
        was = os.environ.get("FMD_PAIR")
        if was == "1":
            try:
                import torch
                torch.cuda.empty_cache()
            except Exception:
                pass
            os.environ["FMD_PAIR"] = "2"
        try:
            step(self.L, self.h)
        finally:
            if was == "1":
                os.environ["FMD_PAIR"] = "1"
        if self.L.fmd_dev_line_count3(self.h, buf, 1, C.byref(cnt)) != 0 or not cnt.value:
            return None
        self.pair_lines = int(buf[2])      # 128-byte two-base blocks requested (k_ovl_pair)
        return int(buf[0]), int(buf[1])

    def close(self):
        if self.h is not None:
            self.L.fmd_dev_close(self.h)
            self.h = None


def timed(torch, dist, dev, stream, step, steps, warmup):
    """W untimed + K timed steps, barrier + synchronize on both sides, max wall over ranks; HIP events per step."""
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
        t = torch.tensor([wall], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())
    return wall, float(np.mean([a.elapsed_time(b) for a, b in evs]))


def ref_driver():
    drv = os.path.join(ROOT, "oracle", "_ref", "libref_driver.so")
    if not os.path.exists(drv) or os.environ.get("FMD_BENCH_FORCE_PORT"):
        return None
    L = C.CDLL(drv)
    L.refdrv_load.restype = C.c_void_p; L.refdrv_load.argtypes = [C.c_char_p]
    L.refdrv_free.argtypes = [C.c_void_p]
    L.refdrv_bsearch.restype = C.c_double
    L.refdrv_bsearch.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    L.refdrv_overlap.restype = C.c_double
    L.refdrv_overlap.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_int, C.c_void_p, C.c_int]
    L.refdrv_smem.restype = C.c_double
    L.refdrv_smem.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_int, C.c_uint32, C.c_void_p, C.c_void_p, C.c_int]
    return L


def baseline_obj(value, unit, cores, kind, sample, rate1):
    return {"value": value, "unit": unit, "cores": cores, "kind": kind, "sample": sample,
            "one_thread": rate1, "scaling_efficiency": value / (rate1 * cores) if rate1 else None}


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
        ctr = Counter(api, fmd_path, local_rank)
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


def oracle_counters(fmd_path, fn):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import orcbind
    o = orcbind.OrcIndex(fmd_path)
    o.counters()
    fn(o)
    c = o.counters()
    o.close()
    return c


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
    ctr = Counter(api, fmd_path, local_rank)
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
    ctr = Counter(api, fmd_path, local_rank)
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
    ctr = Counter(api, fmd_path, local_rank)
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
    ctr = Counter(api, fmd_path, local_rank)
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


def bench_raw_reads(torch, api, workload, dev, local_rank, n_reads, L, steps, warmup, legs):
    """the index of reads that carry 1 % substitutions (what `fermi exact` and `fermi correct` see
    before error correction).  Built once, used by the SMEM leg and the k-mer harvest leg."""
    err = float(os.environ.get("FMD_BENCH_SMEM_ERR", "0.01"))
    t0 = time.time()
    rd = workload.ReadsOnDevice.synth(n_reads, L, 30, err, dev)
    d_bwt, n_sym = workload.build_bwt_on_device(rd, local_rank)
    torch.cuda.synchronize()
    fmd_path = os.path.join(tempfile.gettempdir(), "fmd_bench_raw_%d_%d.fmd" % (n_reads, os.getpid()))
    workload.write_fmd_from_device_bwt(d_bwt, n_sym, fmd_path, local_rank)
    index = api.DevIndex.from_bwt_dev(d_bwt, n_sym, local_rank)
    api.lib().fmd_dev_free(d_bwt)
    log("raw-read index: %d reads at e=%g, %d symbols, %.1fs" % (n_reads, err, n_sym, time.time() - t0))
    sm = km = raw = ec = None
    try:
        if "smem" in legs:
            sm = bench_smem(torch, api, index, rd, err, n_sym, fmd_path, dev, local_rank, n_reads, L, steps, warmup)
            torch.cuda.empty_cache()
        if "kmer" in legs or "ecfix" in legs:
            km, tab = bench_kmer(torch, api, index, n_sym, fmd_path, dev, local_rank, n_reads, steps, warmup)
            torch.cuda.empty_cache()
            if "ecfix" in legs:
                ec = bench_ecfix(torch, api, rd, tab, n_sym, dev, local_rank, n_reads, L, steps, min(warmup, 1), fmd_path)
                if os.environ.get("FMD_BENCH_PMC", "1") != "0" and "roofline" in ec:
                    del tab
                    torch.cuda.empty_cache()
                    note = pmc_in_run(fmd_path, n_reads, leg="ecfix")
                    log("in-run PMC pass (ecfix): %s" % note)
                    apply_traffic(ec["roofline"])
                    ec["pmc_in_run"] = note
            tab = None
            torch.cuda.empty_cache()
        if "overlap" in legs and os.environ.get("FMD_BENCH_RAW_OVERLAP", "1") != "0":
            raw = bench_overlap_raw(torch, api, index, dev, n_reads, L, err, fmd_path)
    finally:
        if os.path.exists(fmd_path):
            os.remove(fmd_path)
        index.close()
    return sm, km, raw, ec


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
    ctr = Counter(api, fmd_path, index.device)
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
