"""What every leg of bench.py shares: logging, the source fingerprint of a leg, the counter passes (rocprofv3 --pmc) a run spawns over itself,
the roofline object, the instrumented counting handle, the timed region, the reference driver and the CPU-baseline object."""
import ctypes as C
import hashlib
import json
import numpy as np
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
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

    def __init__(self, api, fmd_path, device, main=None):
        """main: the timed handle -- the work areas it keeps between calls go back to the device first (fmd_dev_trim), the counting handle needs the room"""
        self.L = api.count_lib()
        if main is not None:
            api.lib().fmd_dev_trim.restype = C.c_uint64
            api.lib().fmd_dev_trim(main.h)
        self.h = None
        self.pair_lines = 0
        if self.L is None or not fmd_path:
            return
        h = C.c_void_p()
        rc = self.L.fmd_dev_open_file(device, fmd_path.encode(), C.byref(h))
        if rc == 0:
            self.h = h
        else:
            log("  counting handle: fmd_dev_open_file failed (%d): no requested-bytes figure for this leg" % rc)
        if os.environ.get("FMD_BENCH_DEBUG_MEM"):
            import torch
            f, t = torch.cuda.mem_get_info()
            log("  [mem] counter opened (rc %d): %.1f GB free of %.1f (torch: %.1f allocated, %.1f reserved)" % (rc, f / 1e9, t / 1e9, torch.cuda.memory_allocated() / 1e9, torch.cuda.memory_reserved() / 1e9))

    def run(self, step):
        """step(L, h) launches one step on library L / handle h; returns (rank blocks, other lines) or None."""
        if self.h is None:
            return None
        buf = (C.c_uint64 * 3)()
        cnt = C.c_int(0)
        self.L.fmd_dev_line_count3(self.h, buf, 1, C.byref(cnt))
        # The timed handle has its two-base blocks (FMD_PAIR=1); this one builds its own inside the step and must not be turned down by the "only where the
        # job still finds its room" rule of fmd_pairs_ensure, or the step counted is not the step timed: torch's cached blocks back to the device, and FMD_PAIR=2.
        def mem(tag):
            if os.environ.get("FMD_BENCH_DEBUG_MEM"):
                import torch
                f, t = torch.cuda.mem_get_info()
                log("  [mem] %s: %.1f GB free of %.1f (torch: %.1f allocated, %.1f reserved)" % (tag, f / 1e9, t / 1e9, torch.cuda.memory_allocated() / 1e9, torch.cuda.memory_reserved() / 1e9))
        mem("counter: before the step")
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
        mem("counter: after the step")
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


def oracle_counters(fmd_path, fn):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import orcbind
    o = orcbind.OrcIndex(fmd_path)
    o.counters()
    fn(o)
    c = o.counters()
    o.close()
    return c
