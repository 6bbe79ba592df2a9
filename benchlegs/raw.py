"""bench.py: the legs that run on the index of reads with errors (configs[2]) -- one index, built once: SMEM, k-mer harvest, ec_fix, overlap discovery on raw reads."""
import os
import tempfile
import time

from benchlegs.common import apply_traffic, log, pmc_in_run
from benchlegs.ecfix import bench_ecfix
from benchlegs.kmer import bench_kmer
from benchlegs.overlap import bench_overlap_raw
from benchlegs.smem import bench_smem

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
