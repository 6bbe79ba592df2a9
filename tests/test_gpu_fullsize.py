"""GPU, full size: BASELINE.json's configs at the sizes they name, inside `pytest -m gpu`.

  configs[1], [2], [3] at the sizes they name (10 M reads for backward search, 50 M for the rest): index built on the GPU, backward
      search, overlap discovery (+ check_left, error-free and raw reads), SMEM, the k-mer harvest and the correction pass compared with
      the reference (oracle/_ref when it travelled, the oracle otherwise) on RANDOM samples of the ids / reads / buckets, through
      bench.py's own legs at its default sizes (the same code the driver times), one step each;
  the 32-bit size class at scale: 10 M x 100 bp -- `fermi-amd build`, `unitig -l50` (error-free and raw) and `correct` against the md5s
      of the reference binary's output (tests/golden/md5_10m.json: two hours of the reference in the build container);
  configs[0]: 1 M x 100 bp -- `fermi-amd build`, `unitig -l50` (one GPU and two replicas) and `correct` against the md5s of
      the reference binary's output (tests/golden/md5_1m.json, made by tests/golden/make_md5_1m.py where
      /root/reference exists).
"""
import hashlib
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
AMD = os.path.join(ROOT, "fermi_amd", "bin", "fermi-amd")


def _md5_stream(cmd, env=None):
    h = hashlib.md5()
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, env=env)
    n = 0
    for blk in iter(lambda: p.stdout.read(1 << 24), b""):
        h.update(blk); n += len(blk)
    assert p.wait() == 0, cmd
    return [h.hexdigest(), n]


def test_50m_reads_every_leg_bit_exact_on_random_samples(gpu):
    """bench.py exactly as the driver runs it (default sizes: configs[2] / configs[3] at 5*10^7 reads -- 1.01*10^10 symbols, the 64-bit kernels --, configs[1] at 10^7), one
    step per leg: what the driver's bench line would only say in a string turns this suite red."""
    env = dict(os.environ, FMD_BENCH_CPU_SAMPLE="200000", FMD_BENCH_CPU_SAMPLE_OVLP="100000", FMD_BENCH_CPU_SAMPLE_SMEM="100000", FMD_BENCH_CPU_SAMPLE_KMER="2048",
               FMD_BENCH_CPU_SAMPLE_ECFIX="100000", FMD_BENCH_CPU_SAMPLE_OVLP_RAW="50000", FMD_BENCH_PROBE="0", FMD_BENCH_PMC="0", FMD_BENCH_HOST_API="0")
    for k in ("FMD_BENCH_READS", "FMD_BENCH_BSEARCH_READS", "FMD_BENCH_LEGS"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=900)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    d = json.loads(p.stdout.decode().strip().splitlines()[-1])
    assert d["config"]["reads"] == 50_000_000 and d["overlap_discovery"]["strands_this_rank"] == 100_000_000 and d["config"]["index_symbols"] == 10_100_000_000
    assert d["parity_vs_cpu_on_sample"] == "bit-exact"                                   # overlap records + neighbours, random ids
    assert d["overlap_discovery"]["id_order_one_pass_walk"]["same_results"].startswith("identical")   # the sorted job against the one-pass walk in id order: every strand
    assert d["overlap_discovery"]["overflow_records"] == 0
    assert d["check_left"]["parity_vs_oracle_on_sample"].startswith("bit-exact") and d["check_left"]["lfork_verdicts_equal_exact_kernel"]
    assert d["backward_search"]["parity_vs_cpu_on_sample"] == "bit-exact" and d["backward_search"]["hits"] == 10_000_000
    assert d["smem"]["parity_vs_cpu_on_sample"] == "bit-exact" and d["smem"]["overflow_reads"] == 0
    assert d["kmer_harvest"]["parity_vs_cpu_on_sample"] == "bit-exact"
    ec = d["ec_fix"]                                                                     # the correction pass on the table the harvest leg built: bases, qualities, info words
    assert ec["parity_vs_cpu_on_sample"].startswith("bit-exact") and ec["bases_changed"] > 25_000_000 and ec["reads_whose_trace_overflowed"] < 1000
    raw = d["overlap_discovery_on_raw_reads"]                                            # reads with 1 % errors: forks, the general group kernels
    assert raw["parity_vs_cpu_on_sample"] == "bit-exact" and raw["same_results_both_ways"] and raw["forked"] > 0
    assert raw["overflow_records"] == 0 and raw["rows_completed_in_the_side_table"]["are_exactly_the_flagged_rows"]   # every row has its answer when the clock stops
    assert raw["check_left"]["parity_vs_oracle_on_sample"].startswith("bit-exact") and raw["check_left"]["back_bifurcations"] > 0
    for leg in (d, d["check_left"], d["backward_search"], d["smem"], d["kmer_harvest"], ec, raw):
        r = leg["roofline"]
        for f in (r["frac"], r["frac_requested"]):                                      # frac = PMC bytes / time / peak where the run could measure them
            assert f is None or 0 < f <= 1.0, r                                         # a fraction is a fraction
        assert r["frac"] is None or r["frac_basis"].startswith("measured") or r["frac"] == r["frac_requested"], r


def test_1m_reads_cli_md5_equals_the_reference(gpu, tmp_path):
    sys.path.insert(0, os.path.join(HERE, "golden"))
    import make_md5_1m as gen
    want = json.load(open(os.path.join(HERE, "golden", "md5_1m.json")))
    d = str(tmp_path)
    gen.write_fastq(d + "/clean.fq", 0.0, False)
    subprocess.check_call([AMD, "build", "-fo", d + "/clean.fmd", d + "/clean.fq"], stderr=subprocess.DEVNULL)
    assert _md5_stream(["cat", d + "/clean.fmd"]) == want["clean_fmd"]                  # the .fmd `fermi build` writes, byte for byte
    assert _md5_stream([AMD, "unitig", "-l50", d + "/clean.fmd"]) == want["unitig_l50_t1"]
    assert _md5_stream([AMD, "unitig", "-l50", "-g", "0,0", d + "/clean.fmd"]) == want["unitig_l50_t1"]   # two replicas, sharded ids
    assert _md5_stream([AMD, "unitig", "-l50", d + "/clean.fmd"], env=dict(os.environ, FMD_CHECK_LEFT_EXACT="1")) == want["unitig_l50_t1"]
    gen.write_fastq(d + "/raw.fq", 0.01, True)
    subprocess.check_call([AMD, "build", "-fo", d + "/raw.fmd", d + "/raw.fq"], stderr=subprocess.DEVNULL)
    assert _md5_stream(["cat", d + "/raw.fmd"]) == want["raw_fmd"]
    assert _md5_stream([AMD, "correct", "-t8", d + "/raw.fmd", d + "/raw.fq"]) == want["correct_t1"]
    assert _md5_stream([AMD, "correct", "-t8", "-g", "0,0,0", d + "/raw.fmd", d + "/raw.fq"]) == want["correct_t1"]   # harvest by last base + batches split over three replicas
    assert _md5_stream([AMD, "unitig", "-l50", d + "/raw.fmd"]) == want["unitig_raw_l50_t1"]   # reads with errors: forks, tips, back-bifurcations


def test_10m_reads_cli_md5_equals_the_reference(gpu, tmp_path):
    """Whole-output parity at 10^7 reads (2.02*10^9 symbols: the 32-bit kernels at the top of their range): the .fmd of `build`, the MAG of `unitig -l50` on error-free and on
    raw reads (1 % substitutions: forks, tips, back-bifurcations, the side table) and the FASTQ of `correct`, against the md5s of the reference binary's `-t1` output
    (tests/golden/md5_10m.json, made by tests/golden/make_md5_1m.py 10000000 where /root/reference exists)."""
    import torch
    sys.path.insert(0, os.path.join(HERE, "golden"))
    sys.path.insert(0, ROOT)
    import make_md5_1m as gen
    from fermi_amd import synth
    if not os.path.exists(os.path.join(HERE, "golden", "md5_10m.json")):
        pytest.skip("tests/golden/md5_10m.json has not been made yet (two hours of the reference)")
    want = json.load(open(os.path.join(HERE, "golden", "md5_10m.json")))
    n = want["n_reads"]
    assert n == 10_000_000
    d = str(tmp_path)
    g = synth.genome_torch(synth.DEFAULT_SEED, n, 100, 30, "cuda")

    def on_gpu(err):
        return lambda s, c: synth.reads_torch(synth.DEFAULT_SEED, n, 100, 30, err, "cuda", start=s, count=c, gen=g).cpu().numpy()
    gen.write_fastq(d + "/clean.fq", 0.0, False, n, on_gpu(0.0))
    subprocess.check_call([AMD, "build", "-fo", d + "/clean.fmd", d + "/clean.fq"], stderr=subprocess.DEVNULL)
    os.remove(d + "/clean.fq")
    assert _md5_stream(["cat", d + "/clean.fmd"]) == want["clean_fmd"]
    assert _md5_stream([AMD, "unitig", "-l50", d + "/clean.fmd"]) == want["unitig_l50_t1"]
    os.remove(d + "/clean.fmd")
    gen.write_fastq(d + "/raw.fq", 0.01, True, n, on_gpu(0.01))
    del g
    torch.cuda.empty_cache()
    subprocess.check_call([AMD, "build", "-fo", d + "/raw.fmd", d + "/raw.fq"], stderr=subprocess.DEVNULL)
    assert _md5_stream(["cat", d + "/raw.fmd"]) == want["raw_fmd"]
    assert _md5_stream([AMD, "correct", "-t16", d + "/raw.fmd", d + "/raw.fq"]) == want["correct_t1"]
    os.remove(d + "/raw.fq")
    assert _md5_stream([AMD, "unitig", "-l50", d + "/raw.fmd"]) == want["unitig_raw_l50_t1"]


def test_bench_n2_path_sharded_ids_and_gather_on_one_gpu(gpu):
    """`python bench.py --gpus 2` launched PLAINLY, as the driver's N = 1 command is: bench.py becomes its own launcher (N ranks of itself through
    torch.distributed.run, one per GPU), here with both ranks on GPU 0 and gloo instead of RCCL (FMD_BENCH_BACKEND / FMD_BENCH_SHARE_GPU): rank r
    computes the ids i = r (mod 2), the packed rows are gathered on rank 0 inside the timed step, and rank 0 recomputes a sample of rank 1's rows."""
    env = dict(os.environ, FMD_BENCH_READS="1000000", FMD_BENCH_BACKEND="gloo", FMD_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=600)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    d = json.loads([l for l in p.stdout.decode().splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["reads"] == 1_000_000
    assert d["overlap_discovery"]["strands_this_rank"] == 1_000_000            # half of the 2 * 10^6 ids
    g = d["overlap_discovery"]["record_gather_rccl"]
    assert g["check"].startswith("ok"), g
    assert 64 < g["bytes_per_strand"] < 200
    pp = d["product_path_unitig_g"]                                           # what `fermi-amd unitig -g 0,0` does with the same .fmd: two replicas -> slim table, host side included
    assert pp["rows"] == 2_000_000 and pp["strands_per_s_rows_phase"] > 1e6 and 40 < pp["table_bytes_per_row"] < 50, pp


def test_repeat_rich_ragged_2m_reads_md5_and_random_ids_vs_the_reference(gpu, oracle_lib, tmp_path):
    """The read set a uniform random genome never gives: 2 M reads of 70-150 bp, 5 % of the genome in repeat families,
    1 % substitutions, exact duplicates and proper substrings of other reads (tests/golden/make_md5_repeat.py, fermi_amd/synth.py).
    `fermi-amd build`, `unitig -l50` and `correct` against the md5s of the reference binary's output; 10^5 random ids (records + neighbours)
    against the reference's own functions and check_left_simple of 4 000 ids against the oracle where oracle/_ref travelled."""
    import ctypes as C
    import numpy as np
    sys.path.insert(0, os.path.join(HERE, "golden"))
    import make_md5_repeat as gen
    want = json.load(open(os.path.join(HERE, "golden", "md5_repeat.json")))
    d = str(tmp_path)
    reads = gen.make_reads()
    assert len(reads) == want["n_reads"] and sum(len(r) for r in reads) == want["total_bases"]
    gen.write_fastq(d + "/rep.fq", reads)
    del reads
    assert _md5_stream(["cat", d + "/rep.fq"]) == want["fastq"]
    subprocess.check_call([AMD, "build", "-fo", d + "/rep.fmd", d + "/rep.fq"], stderr=subprocess.DEVNULL)
    assert _md5_stream(["cat", d + "/rep.fmd"]) == want["fmd"]
    p = subprocess.run([AMD, "unitig", "-l50", d + "/rep.fmd"], stdout=open(d + "/rep.mag", "wb"), stderr=subprocess.PIPE, env=dict(os.environ, FMD_TIMING="1", FMD_OVLP_STATS="1"))
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    print("\n".join(l for l in p.stderr.decode().splitlines() if "M::" in l)[-6000:])      # the kernel mix of this read set (pytest -s; DESIGN.md quotes it)
    assert _md5_stream(["cat", d + "/rep.mag"]) == want["unitig_l50_t1"]
    assert _md5_stream([AMD, "unitig", "-l50", "-g", "0,0", d + "/rep.fmd"], env=dict(os.environ, FMD_WALK_THREADS="1")) == want["unitig_l50_t1"]   # two replicas, the sequential walk
    assert _md5_stream([AMD, "correct", "-t16", d + "/rep.fmd", d + "/rep.fq"]) == want["correct_t1"]
    # ---- random ids against the reference's own fm_retrieve + fm6_is_contained + fm6_get_nei
    sys.path.insert(0, ROOT)
    import bench
    import orcbind
    api = gpu
    ix = api.DevIndex.open(d + "/rep.fmd")
    n_ids = int(ix.mcnt[1])
    sel = np.sort(np.random.default_rng(11).choice(n_ids, 100_000, replace=False)).astype(np.uint64)
    rec, nei, _ = ix.overlap_sorted(sel, 50, 150, 16, 0)
    keep = (rec["flags"] & api.OVLP_F_OVERFLOW) == 0
    assert keep.mean() > 0.99
    assert (rec["status"] == -3).sum() > 500 and (rec["n_nei"] > 4).sum() > 100 and (rec["flags"] & api.OVLP_F_FORKED).sum() > 1000   # contained reads, many neighbours, forks: all there
    if bench.ref_driver() is not None:
        _, ok = bench.cpu_overlap(d + "/rep.fmd", sel, 50, rec, nei, keep=keep)
        assert ok, "records / neighbours of random ids differ from the reference"
    sub = sel[::25]
    r_cl, _, _ = ix.overlap_sorted(sub, 50, 150, 16, 0, check_left=True)
    o = orcbind.OrcIndex(d + "/rep.fmd")
    w_cl, _, _ = o.overlap_batch(sub, 50, 150, 16, 8, check_left=True)
    o.close(); ix.close()
    k2 = (r_cl["flags"] & api.OVLP_F_OVERFLOW) == 0
    assert np.array_equal(r_cl["reserved"][k2], w_cl["reserved"][k2]) and (w_cl["reserved"] == 1).sum() > 10
