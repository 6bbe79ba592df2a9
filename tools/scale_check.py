#!/usr/bin/env python3
"""Indexes beyond 2.6*10^10 symbols on one MI355X (task 7; BASELINE config 5 = 7*10^8 x 100 bp): synthetic
N x 100 bp reads (30x, e = 0) generated in HBM piece by piece, the index built by the prefix-bucketed GPU builder -- through
the byte BWT (`bwt`: what fits text + BWT + index) or in place (`inplace`: packed text, BWT slices written straight into the
device layout; the only form that carries 1.4*10^11 symbols) --, then
  * the device-wide rank self-check (the `chkbwt -r` equivalent) over every position,
  * backward search of a sample of the reads (every one must hit its own index),
  * backward search and overlap discovery on random samples against the REFERENCE (oracle/_ref when it travelled, the oracle
    otherwise) through the .fmd the product writes (skipped with `noref`),
  * config 5's share of one GPU out of `share` (ids i = 0 mod share): one timed pass of overlap discovery on this index.
  * `props`: size-independent properties of the discovery on random ids, from what the GENERATOR knows (no reference needed):
    the sequence rows are the generated reads (fm_retrieve inverts the index), a strand whose successor on the genome is unambiguous
    has exactly that read as its one neighbour with the overlap the two start positions give, and the edge is mutual (the
    successor's reverse strand is followed by the strand's reverse strand with the same overlap),
  * `kmer` with `noref`: the harvest's solid k-mers cross-checked by backward search on a sample.
  * `raw`: the reads carry 1 % substitutions (the forked path of fm6_get_nei: general group kernels, side table, 64-bit fast kernels beyond 2^32 symbols): the
    discovery of the random ids runs through BOTH host forms (fmd_ovlp_batch in id order, the sorted job) with room for 16 neighbours, and check_left of a
    sub-sample is compared with the oracle; `props` (which assume error-free reads) is ignored.
  * `dry`: the allocations of one rank of the `share`-rank step (fmd_ovlp_dist_new with a stand-in communicator, FMD_DIST_DRY=1), as the root and as a peer.
  * `unitig`: `fermi-amd unitig -l50` on the .fmd as its own process: time, peak resident set, md5 of the MAG (and the same by build/old/fermi-amd-old where present).
  * `genmag` (with `unitig`, error-free reads): the MAG of `fermi-amd unitig` checked EXACTLY against the generator (tools/mag_vs_generator.py: every unitig's
    sequence, coverage string and number of reads, no other records) -- the check that needs no reference run; `noref fmd`: write the .fmd without the reference's checks.
Usage: python tools/scale_check.py [n_reads=250000000] [bwt|inplace] [sample=20000] [share=8] [noref] [fmd] [kmer] [props] [raw] [dry] [unitig] [genmag]"""
import ctypes as C, os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from fermi_amd import api, hostlib, synth, workload
import bench

n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 250_000_000
mode = sys.argv[2] if len(sys.argv) > 2 else "bwt"
sample = int(sys.argv[3]) if len(sys.argv) > 3 else 20000
share = int(sys.argv[4]) if len(sys.argv) > 4 else 8
noref = "noref" in sys.argv[5:]
WRITE_FMD = (not noref) or "fmd" in sys.argv[5:]      # `noref fmd`: the .fmd is written (for `unitig`), the comparisons with the reference are not run
raw = "raw" in sys.argv[5:]
ERR = 0.01 if raw else 0.0
PROPS = "props" in sys.argv[5:] and not raw
L = 100
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
lib = api.lib()
seed = synth.DEFAULT_SEED
rng = np.random.default_rng(11)


def hbm_used():
    free_b, total_b = torch.cuda.mem_get_info()
    return (total_b - free_b) / 1e9


t0 = time.time()
gen = synth.genome_torch(seed, n_reads, L, 30, dev)
PIECE = 25_000_000
fmd_path = os.path.join(tempfile.gettempdir(), "fmd_scale_%d.fmd" % n_reads)
if mode == "inplace":
    b = C.c_void_p()
    api.check(lib.fmd_builder_new(0, n_reads, L, C.byref(b)))
    for s in range(0, n_reads, PIECE):
        c = min(PIECE, n_reads - s)
        piece = synth.reads_torch(seed, n_reads, L, 30, ERR, dev, start=s, count=c, gen=gen)
        api.check(lib.fmd_builder_add_dev(b, None, c, piece.data_ptr()))
        torch.cuda.synchronize()
        del piece
    print("text in HBM (4 bits per symbol): %d reads, %.1f s, HBM in use %.1f GB" % (n_reads, time.time() - t0, hbm_used()), flush=True)
    torch.cuda.empty_cache()
    t0 = time.time()
    h = C.c_void_p()
    api.check(lib.fmd_builder_finish(b, C.byref(h)))
    index = api.DevIndex(h)
    n_sym = index.n
    t_build = time.time() - t0
    print("index built in place: %d symbols in %.1f s (%.2e symbols/s), %.1f GB in HBM, HBM in use %.1f GB" % (n_sym, t_build, n_sym / t_build, index.hbm_bytes / 1e9, hbm_used()), flush=True)
    if WRITE_FMD:   # the reference (and `fermi-amd unitig`) read fermi's file: decode the BWT back from the device layout, encode RLD on the host
        t0 = time.time()
        bwt = np.empty(n_sym, dtype=np.uint8)
        for o in range(0, n_sym, 1 << 30):
            m = min(1 << 30, n_sym - o)
            api.check(lib.fmd_dev_export_bwt(index.h, o, m, bwt.ctypes.data + o))
        hostlib.write_rld_from_bwt(bwt, fmd_path)
        del bwt
        print(".fmd written (BWT decoded from the device layout + host RLD encoder): %.1f GB in %.1f s" % (os.path.getsize(fmd_path) / 1e9, time.time() - t0), flush=True)
else:
    rd = workload.ReadsOnDevice.__new__(workload.ReadsOnDevice)
    rd.n, rd.L = n_reads, L
    rd.flat = torch.zeros(n_reads * L + 64, dtype=torch.uint8, device=dev)
    for s in range(0, n_reads, PIECE):
        c = min(PIECE, n_reads - s)
        rd.flat[s * L:(s + c) * L].view(c, L).copy_(synth.reads_torch(seed, n_reads, L, 30, ERR, dev, start=s, count=c, gen=gen))
    rd.off = torch.arange(n_reads + 1, dtype=torch.int64, device=dev) * L
    rd.total = n_reads * L
    torch.cuda.synchronize()
    print("reads in HBM: %d x %d bp, %.1f s" % (n_reads, L, time.time() - t0), flush=True)
    t0 = time.time()
    d_bwt, n_sym = workload.build_bwt_on_device(rd, 0)
    torch.cuda.synchronize()
    t_build = time.time() - t0
    print("GPU BWT build: %d symbols in %.1f s (%.2e symbols/s); HBM in use after the build %.1f GB" % (n_sym, t_build, n_sym / t_build, hbm_used()), flush=True)
    del rd
    torch.cuda.empty_cache()
    if WRITE_FMD:
        t0 = time.time()
        workload.write_fmd_from_device_bwt(d_bwt, n_sym, fmd_path, 0)
        print(".fmd written (GPU run-length pass + host RLD encoder): %.1f GB in %.1f s" % (os.path.getsize(fmd_path) / 1e9, time.time() - t0), flush=True)
    t0 = time.time()
    index = api.DevIndex.from_bwt_dev(d_bwt, n_sym, 0)
    lib.fmd_dev_free(d_bwt)
    print("device index: %.1f GB in HBM, %.1f s" % (index.hbm_bytes / 1e9, time.time() - t0), flush=True)

t0 = time.time()
bad, first = C.c_uint64(), C.c_uint64()
api.check(lib.fmd_dev_check_rank(index.h, C.byref(bad), C.byref(first)))
print("rank self-check over all %d positions: %d bad (%.1f s)" % (n_sym, bad.value, time.time() - t0), flush=True)
assert bad.value == 0
# a spread sample of the reads: blocks of 1000 consecutive reads
starts = np.sort(rng.choice(n_reads // 1000, max(1, sample // 1000), replace=False)) * 1000
q = torch.cat([synth.reads_torch(seed, n_reads, L, 30, ERR, dev, start=int(s), count=1000, gen=gen) for s in starts]).cpu().numpy()
if PROPS:
    # ---- what the generator knows: where every read starts and which way round it is
    G = gen.shape[0]
    t0 = time.time()
    pos = torch.empty(n_reads, dtype=torch.int64, device=dev); sbit = torch.empty(n_reads, dtype=torch.int8, device=dev)
    for s0 in range(0, n_reads, 50_000_000):
        r_ = torch.arange(s0, min(n_reads, s0 + 50_000_000), dtype=torch.int64, device=dev)
        pos[s0:s0 + r_.numel()] = synth._umod(synth.rnd_torch(seed, 2, r_), G - L + 1)
        sbit[s0:s0 + r_.numel()] = synth._lsr(synth.rnd_torch(seed, 3, r_), 63).to(torch.int8)
    P, order = torch.sort(pos)
    del pos
    m = min(sample, n_reads - 4)
    ii = torch.from_numpy(np.sort(rng.choice(n_reads - 4, m, replace=False)) + 2).to(dev)     # ranks in start order, away from the ends
    d_next, d_prev = P[ii + 1] - P[ii], P[ii] - P[ii - 1]
    # unambiguous successor on the + strand: the next start is 1..50 bases on, nobody shares this start or that one
    okp = (d_next >= 1) & (d_next <= L - 50) & (P[ii + 2] != P[ii + 1]) & (d_prev >= 1)
    okm = (d_prev >= 1) & (d_prev <= L - 50) & (P[ii - 2] != P[ii - 1]) & (d_next >= 1)
    r_a = order[ii]
    plus = lambda rr: 2 * rr + sbit[rr].to(torch.int64)          # the strand of read rr that reads like the genome
    minus = lambda rr: 2 * rr + 1 - sbit[rr].to(torch.int64)
    a_p, b_p, o_p = plus(r_a)[okp], plus(order[ii + 1])[okp], (L - d_next)[okp]
    a_m, b_m, o_m = minus(r_a)[okm], minus(order[ii - 1])[okm], (L - d_prev)[okm]
    A = torch.cat([a_p, a_m]).cpu().numpy().astype(np.uint64); B = torch.cat([b_p, b_m]).cpu().numpy().astype(np.uint64)
    OV = torch.cat([o_p, o_m]).cpu().numpy()
    # the reads themselves (for the retrieve check): strand A[i] as the generator makes it
    ra = torch.from_numpy((A >> np.uint64(1)).astype(np.int64)).to(dev)
    win = gen[synth._umod(synth.rnd_torch(seed, 2, ra), G - L + 1)[:, None] + torch.arange(L, device=dev)[None, :]]
    fw = sbit[ra].to(torch.int64) == torch.from_numpy((A & np.uint64(1)).astype(np.int64)).to(dev)     # strand 2r + s reads like the genome
    want_seq = torch.where(fw[:, None], win, (5 - win).flip(1)).cpu().numpy()
    del P, order, sbit, win
    torch.cuda.empty_cache()
    print("generator: %d strands with an unambiguous successor among %d sampled reads (%.1f s)" % (len(A), m, time.time() - t0), flush=True)
del gen
torch.cuda.empty_cache()
cnt, beg, end = index.backward_search(q)
assert (cnt >= 1).all() and np.array_equal(end - beg + 1, cnt), "a read of the set does not hit its own index"
print("backward search: all %d sampled reads hit their own index (multiplicities 1..%d)" % (len(q), int(cnt.max())), flush=True)
ids = np.sort(rng.choice(2 * n_reads, min(sample, 2 * n_reads), replace=False)).astype(np.uint64)
MAX_NEI = 16 if raw else 4
rec, nei, seq = index.overlap(ids, 50, max_len=100, max_nei=MAX_NEI, check_left=False)
print("overlap discovery of %d random sequence ids: %d with a neighbour, %d contained, %d overflow" % (len(ids), int((rec["n_nei"] > 0).sum()), int((rec["status"] == -3).sum()),
      int(((rec["flags"] & api.OVLP_F_OVERFLOW) != 0).sum())), flush=True)
if PROPS:
    t0 = time.time()
    all_ids = np.unique(np.concatenate([A, A ^ np.uint64(1), B, B ^ np.uint64(1)]))
    prec, pnei, pseq = index.overlap(all_ids, 50, max_len=100, max_nei=4, check_left=False)
    at = lambda x: np.searchsorted(all_ids, x)
    ia, ia1, ib, ib1 = at(A), at(A ^ np.uint64(1)), at(B), at(B ^ np.uint64(1))
    assert (prec["len"][ia] == L).all() and np.array_equal(pseq[ia, :L], want_seq), "fm_retrieve does not give the generated reads back"
    assert (prec["status"][ia] == 0).all() and (prec["n_nei"][ia] == 1).all(), "a strand with an unambiguous successor must have exactly one neighbour"
    assert np.array_equal(pnei["info"][ia, 0].astype(np.int64), OV.astype(np.int64)), "overlap length differs from what the start positions give"
    assert np.array_equal(pnei["x"][ia, 0, 0], prec["k"][ib, 0]), "the neighbour is not the successor on the genome"
    assert (prec["n_nei"][ib1] >= 1).all() and np.array_equal(pnei["x"][ib1, 0, 0], prec["k"][ia1, 0]) and np.array_equal(pnei["info"][ib1, 0].astype(np.int64), OV.astype(np.int64)), "the edge is not mutual"
    assert np.array_equal(prec["k"][ia, 0], prec["k"][ia1, 1]) and np.array_equal(prec["k"][ia, 2], prec["k"][ia1, 2]), "`$read$` intervals of the two strands do not mirror each other"
    print("properties on %d strands (+ their reverse strands, successors and the successors' reverse strands: %d ids): sequences = the generated reads, one neighbour = "
          "the successor on the genome with the overlap the start positions give, edges mutual, strand intervals mirrored (%.1f s)" % (len(A), len(all_ids), time.time() - t0), flush=True)
    del prec, pnei, pseq
if noref and "kmer" in sys.argv[5:]:   # the harvest of `fermi correct` over the whole index, cross-checked by backward search
    import math
    w = min(27, int(math.log(n_sym) / math.log(4) + 8.499)); suf_len = w - 15 if w > 15 else 1
    t0 = time.time()
    kb, kk, kv, kc = index.kmer_collect(w, 3, suf_len)
    t_h = time.time() - t0
    print("k-mer harvest (k = %d, min_occ 3): %d solid %d-mers in %.1f s incl. the copy to the host (%.2e per s), %d informative; HBM in use %.1f GB"
          % (w, len(kb), w - 1, t_h, len(kb) / t_h, kc[1], hbm_used()), flush=True)
    sel = np.sort(rng.choice(len(kb), min(len(kb), 200_000), replace=False))
    Ks = (kk[sel].astype(np.uint64) >> np.uint64(2)) << np.uint64(2 * suf_len) | kb[sel].astype(np.uint64)
    km = np.empty((len(sel), w + 1), dtype=np.uint8)
    km[:, 0] = (kk[sel] & 3).astype(np.uint8) + 1                                    # the base the table predicts, to the left
    for dd in range(w):
        km[:, w - dd] = ((Ks >> np.uint64(2 * dd)) & np.uint64(3)).astype(np.uint8) + 1
    del kb, kk, kv
    c1, _, _ = index.backward_search(km)
    c0, _, _ = index.backward_search(np.ascontiguousarray(km[:, 1:]))
    assert (c1 >= 3).all() and (c0 >= c1).all(), "a k-mer the harvest calls solid does not occur min_occ times"
    print("k-mer harvest cross-checked by backward search on %d random solid k-mers: each occurs >= min_occ times with its predicted base" % len(sel), flush=True)
if not noref:
    base, ok = bench.cpu_bsearch(fmd_path, q, cnt, beg, end)
    print("backward search vs %s on the sample: %s (%.0f reads/s on %d host threads)" % (base["kind"], "bit-exact" if ok else "MISMATCH", base["value"], base["cores"]), flush=True)
    assert ok
    keep = (rec["flags"] & api.OVLP_F_OVERFLOW) == 0
    base, ok = bench.cpu_overlap(fmd_path, ids, 50, rec, nei, keep=keep)
    print("overlap discovery vs %s on %d random sequence ids (%d of them without an answer at %d neighbour slots): %s (reference %.0f reads/s on %d threads)"
          % (base["kind"], len(ids), int((~keep).sum()), MAX_NEI, "bit-exact" if ok else "MISMATCH", base["value"], base["cores"]), flush=True)
    assert ok and keep.mean() > 0.99
    if raw:   # the forked path beyond 2^32 symbols: the sorted job (walk passes, fast and general group kernels, 64-bit forms) on a larger sample, and check_left
        ids2 = np.sort(rng.choice(2 * n_reads, min(10 * sample, 2 * n_reads), replace=False)).astype(np.uint64)
        rec2, nei2, _ = index.overlap_sorted(ids2, 50, 100, MAX_NEI, 0)
        keep2 = (rec2["flags"] & api.OVLP_F_OVERFLOW) == 0
        forked = int(((rec2["flags"] & api.OVLP_F_FORKED) != 0).sum())
        _, ok2 = bench.cpu_overlap(fmd_path, ids2, 50, rec2, nei2, keep=keep2)
        print("the sorted job vs the reference on %d random sequence ids: %s (%d forked strands, %d with more than one neighbour, %d without an answer)"
              % (len(ids2), "bit-exact" if ok2 else "MISMATCH", forked, int((rec2["n_nei"] > 1).sum()), int((~keep2).sum())), flush=True)
        assert ok2 and keep2.mean() > 0.99 and forked > len(ids2) // 100
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import orcbind
        sub = ids2[:: max(1, len(ids2) // 4000)]
        r_cl, _, _ = index.overlap_sorted(sub, 50, 100, MAX_NEI, 0, check_left=True)
        o = orcbind.OrcIndex(fmd_path)
        w_cl, _, _ = o.overlap_batch(sub, 50, 100, MAX_NEI, bench.usable_cpus(), check_left=True)
        o.close()
        k3 = (r_cl["flags"] & api.OVLP_F_OVERFLOW) == 0
        same = np.array_equal(r_cl["reserved"][k3], w_cl["reserved"][k3])
        print("check_left_simple of %d ids vs the oracle: %s (%d edges with a verdict, %d back-bifurcations)" % (len(sub), "bit-exact" if same else "MISMATCH", int((w_cl["reserved"] != 2).sum()), int((w_cl["reserved"] == 1).sum())), flush=True)
        assert same and (w_cl["reserved"] == 1).sum() > 0
    if "kmer" in sys.argv[5:]:   # the harvest of `fermi correct` (fm6_traverse + ec_collect) over the whole index, host form (parts as the free HBM asks)
        import math
        w = min(27, int(math.log(n_sym) / math.log(4) + 8.499)); suf_len = w - 15 if w > 15 else 1
        t0 = time.time()
        kb, kk, kv, kc = index.kmer_collect(w, 3, suf_len)
        t_h = time.time() - t0
        print("k-mer harvest (k = %d, min_occ 3): %d solid %d-mers in %.1f s incl. the copy to the host (%.2e per s), %d informative; HBM in use %.1f GB"
              % (w, len(kb), w - 1, t_h, len(kb) / t_h, kc[1], hbm_used()), flush=True)
        nb = 256
        m = kb < nb
        trip = np.sort(kb[m].astype(np.uint64) << np.uint64(40) | kk[m].astype(np.uint64) << np.uint64(8) | kv[m].astype(np.uint64))
        del kb, kk, kv
        base, ok = bench.cpu_kmer(fmd_path, w, 3, suf_len, nb, trip)
        print("k-mer harvest vs %s on suffix buckets 0..%d (%d triples): %s" % (base["kind"], nb - 1, len(trip), "bit-exact" if ok else "MISMATCH"), flush=True)
        assert ok
    if "unitig" not in sys.argv[5:]:
        os.remove(fmd_path)
elif WRITE_FMD and "unitig" not in sys.argv[5:]:
    os.remove(fmd_path)
# ---- one GPU's share of the sharded overlap discovery on this index (BASELINE configs[3] / [4]: ids i = 0 (mod share))
job = bench.OverlapJob(torch, api, index, dev, 2 * n_reads, 0, share, L, 50)
job.compute()                       # warm-up
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(job.stream); job.compute(); e1.record(job.stream)
torch.cuda.synchronize()
ms = e0.elapsed_time(e1)
g = job.rec.view(torch.int32).view(job.n, 16)
print("share 1/%d of the overlap discovery on this index: %d strands in %.1f ms = %.3e strands/s = %.3e reads/s per GPU (%d overflow records, %d with a neighbour); HBM in use %.1f GB"
      % (share, job.n, ms, job.n / ms * 1e3, job.n / 2 / ms * 1e3, int(((g[:, 14] & 2) != 0).sum().item()), int((g[:, 13] > 0).sum().item()), hbm_used()), flush=True)
if "dry" in sys.argv[5:]:   # one rank's allocations of the `share`-rank step on this index, at full size, beside the index: rank 0 (the root: table + arena) and a peer
    from fermi_amd import dist as fdist
    del job
    torch.cuda.empty_cache()
    os.environ["FMD_DIST_DRY"] = "1"
    for r_ in (0, 1):
        for ks in ((1, 0) if share >= 4 else (0,)):   # both shardings of pass 2 (bench.py times both from four ranks up)
            dj = None
            for bt in (0, 10_000_000, 5_000_000, 2_500_000, 1_250_000):   # bench.py's ladder: smaller pieces until the rank has room beside the index
                t0 = time.time()
                try:
                    dj = fdist.DistJob(api, index, fdist.DryComm(api, r_, share), 2 * n_reads, 50, L, 4, pieces=0, key_shard=ks, root=0, host_table=-1, batch=bt)
                except api.FmdError as ex:
                    print("dry run of rank %d of %d, %s, pieces of at most %s strands: %s" % (r_, share, "key shard" if ks else "id shard", bt or "2*10^7", ex), flush=True)
                    torch.cuda.empty_cache()
                    continue
                print("dry run of rank %d of %d, %s: every buffer of a step at full size with pieces of at most %s strands (HBM in use %.1f GB), allocated in %.1f s"
                      % (r_, share, "key shard" if ks else "id shard", bt or "2*10^7", hbm_used(), time.time() - t0), flush=True)
                break
            assert dj is not None, "no piece size leaves rank %d of %d room beside this index" % (r_, share)
            dj.free()
            torch.cuda.empty_cache()
index.close()
if "unitig" in sys.argv[5:] and WRITE_FMD:
    # `fermi-amd unitig -l50` on the .fmd written above, as its own process: wall time, phase times, the peak resident set (FMD_TIMING) and the md5 of the MAG.
    # build/old/fermi-amd-old, where a builder put one (the CLI of an earlier tree linked against this libfmdhip.so), runs beside it: same MAG, its memory.
    import hashlib, subprocess
    try:
        del job
    except NameError:
        pass
    torch.cuda.empty_cache()
    print("HBM in use by this process before the CLI starts: %.1f GB" % hbm_used(), flush=True)
    seen = None
    for name, exe in (("fermi-amd", os.path.join(ROOT, "fermi_amd", "bin", "fermi-amd")), ("the earlier CLI (build/old)", os.path.join(ROOT, "build", "old", "fermi-amd-old"))):
        if not os.path.exists(exe):
            continue
        t0 = time.time()
        h, nb = hashlib.md5(), 0
        mag_path = os.path.join(tempfile.gettempdir(), "fmd_scale_%d.mag" % n_reads) if ("genmag" in sys.argv[5:] and name == "fermi-amd") else None
        mag_fp = open(mag_path, "wb") if mag_path else None
        pr = subprocess.Popen(os.environ.get("FMD_CLI_WRAP", "").split() + [exe, "unitig", "-l50", fmd_path], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, FMD_TIMING="1"))   # (FMD_CLI_WRAP: e.g. a profiler in front of the CLI)
        import threading
        errbuf = []
        th = threading.Thread(target=lambda: errbuf.append(pr.stderr.read())); th.start()
        for blk in iter(lambda: pr.stdout.read(1 << 24), b""):
            h.update(blk); nb += len(blk)
            if mag_fp:
                mag_fp.write(blk)
        rc = pr.wait(); th.join()
        if mag_fp:
            mag_fp.close()
        lines = [l for l in errbuf[0].decode().splitlines() if "M::" in l and "fmd_ovlp]" not in l]
        rss = [l for l in lines if "peak resident set" in l]
        print("unitig -l50 by %s: rc %d, %.1f s, MAG %d bytes md5 %s; %s" % (name, rc, time.time() - t0, nb, h.hexdigest(), rss[-1].split(": ", 1)[1] if rss else "no resident-set line"), flush=True)
        print("\n".join("    " + l for l in lines if "slim_build_core" in l or "slim_finish" in l or "table_build_core" in l or "fmdh_unitig" in l or "packed_batch_core" in l), flush=True)
        if rc != 0:
            print(errbuf[0].decode()[-3000:], flush=True)
        assert rc == 0
        assert seen is None or seen == h.hexdigest(), "the two CLIs print different MAGs"
        seen = h.hexdigest()
        if mag_path:   # `genmag`: the MAG against what the generator knows -- exact, and possible where the reference cannot run (tools/mag_vs_generator.py)
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import mag_vs_generator
            ok_mag = mag_vs_generator.check(mag_path, n_reads, 50, L, 30, seed, log=lambda m_: print("    " + m_, flush=True))
            print("unitig -l50 by %s against the generator (every unitig's sequence, coverage string and number of reads; no other records): %s" % (name, "EXACT" if ok_mag else "MISMATCH"), flush=True)
            os.remove(mag_path)
            assert ok_mag
    os.remove(fmd_path)
print("scale check passed: %d reads, %d symbols, %s builder" % (n_reads, n_sym, mode))
