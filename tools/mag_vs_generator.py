#!/usr/bin/env python3
"""An exact check of `unitig -l<min>` on ERROR-FREE synthetic reads against what the GENERATOR knows -- no reference run needed, so it works at sizes
the reference cannot reach on the box (config 5: 7*10^8 reads).

Reads are windows of one random genome (fermi_amd/synth.py); two reads overlap by >= min bases iff their starts are at most L - min apart, a random
genome has no other 50-base repeats, and identical windows (either strand) are ONE vertex of the graph.  So the unitigs are exactly the maximal runs of
distinct start positions in which consecutive starts are at most L - min apart (unitig.c:227-317: extension stops where there is no overlap, never at a
bifurcation -- there is none), and for the run from start a to start b:
    sequence  = genome[a, b + L) or its reverse complement (which one depends on the seed's strand),
    nsr       = the number of distinct starts in the run (unitig.c:248, 296: one per vertex),
    coverage  = '!' + min(93, number of distinct starts whose read covers the base)   (unitig.c:249-253, 303: '"' for a new base, +1 per later read, capped at '~'),
    both neighbour lists empty ('.').
The header's vertex ids (SA coordinates) are not the generator's to know and are not compared.  The check is exact on everything else: the MAG must hold
exactly one record per run, no other records, every record's sequence AND coverage string equal (by md5 over the whole string) in one of the two orientations.
The rule itself is pinned against the REFERENCE: tests/test_oracle_golden.py::test_generator_rule_reproduces_the_reference_mag runs it on a MAG made by
oracle/_ref/fermi (tests/golden/gen_rule_20k.mag) -- and on the 10^6-read MAG of the reference when this script is given one (profiles/r6_cfg5/rule_vs_reference_1M.txt).

Usage: python tools/mag_vs_generator.py <mag file> <n_reads> [min_match=50] [read_len=100] [coverage=30] [seed]   (torch on a GPU when there is one, numpy otherwise)"""
import hashlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from fermi_amd import synth

LUT = np.frombuffer(b"$ACGTN", dtype=np.uint8)
COMP = np.zeros(256, dtype=np.uint8)
for a_, b_ in zip(b"ACGTN", b"TGCAN"):
    COMP[a_] = b_


def expected(n_reads, L, cov, seed, min_match, use_torch=None):
    """-> (genome as ASCII bytes [G], per run its coverage characters as uint8 [b - a + L], runs as int64 [n_runs, 3] = (first start, last start, distinct starts))"""
    if use_torch is None:
        try:
            import torch
            use_torch = torch.cuda.is_available()
        except Exception:
            use_torch = False
    if use_torch:
        import torch
        dev = torch.device("cuda", 0)
        gen = synth.genome_torch(seed, n_reads, L, cov, dev)
        G = int(gen.shape[0])
        starts = []
        for s0 in range(0, n_reads, 50_000_000):
            r = torch.arange(s0, min(n_reads, s0 + 50_000_000), dtype=torch.int64, device=dev)
            starts.append(torch.unique(synth._umod(synth.rnd_torch(seed, 2, r), G - L + 1)))
        P = torch.unique(torch.cat(starts))
        del starts
        d = torch.zeros(G + 1, dtype=torch.int32, device=dev)
        d[P] += 1
        S = torch.cumsum(d, 0, dtype=torch.int32)          # S[j] = distinct starts <= j
        del d
        brk = torch.nonzero(P[1:] - P[:-1] > (L - min_match)).flatten()
        first = torch.cat([P[:1], P[brk + 1]]); last = torch.cat([P[brk], P[-1:]])
        idx0 = torch.cat([torch.zeros(1, dtype=torch.int64, device=dev), brk + 1]); idx1 = torch.cat([brk + 1, torch.tensor([P.numel()], device=dev)])
        runs = torch.stack([first, last, idx1 - idx0], 1).cpu().numpy()
        covs = []
        for a, b, _ in runs:                                # reads of the NEXT run may reach into this one's bases (overlap < min): they are not in the unitig
            a, b = int(a), int(b)
            for j0 in range(a, b + L, 1 << 28):
                j = torch.arange(j0, min(b + L, j0 + (1 << 28)), dtype=torch.int64, device=dev)
                hi = S[torch.clamp(j, max=b)]
                lo_i = torch.clamp(j - L + 1, min=a) - 1
                lo = torch.where(lo_i >= 0, S[torch.clamp(lo_i, min=0)], torch.zeros_like(hi))
                covs.append((33 + torch.clamp(hi - lo, max=93)).to(torch.uint8).cpu().numpy())
        g_ascii = LUT[gen.cpu().numpy()]
        return g_ascii, covs, runs
    gen = synth.genome(seed, n_reads, L, cov)
    G = gen.shape[0]
    P = np.unique((synth.rnd(seed, 2, np.arange(n_reads, dtype=np.uint64)) % np.uint64(G - L + 1)).astype(np.int64))
    d = np.zeros(G + 1, dtype=np.int32)
    d[P] += 1
    S = np.cumsum(d)                                        # S[j] = distinct starts <= j
    brk = np.nonzero(np.diff(P) > (L - min_match))[0]
    first = np.concatenate([P[:1], P[brk + 1]]); last = np.concatenate([P[brk], P[-1:]])
    idx0 = np.concatenate([[0], brk + 1]); idx1 = np.concatenate([brk + 1, [len(P)]])
    runs = np.stack([first, last, idx1 - idx0], 1)
    covs = []
    for a, b, _ in runs:                                    # reads of the NEXT run may reach into this one's bases (overlap < min): they are not in the unitig
        j = np.arange(a, b + L, dtype=np.int64)
        lo_i = np.maximum(j - L + 1, a) - 1
        lo = np.where(lo_i >= 0, S[np.maximum(lo_i, 0)], 0)
        covs.append((33 + np.minimum(S[np.minimum(j, b)] - lo, 93)).astype(np.uint8))
    return LUT[gen], covs, runs


def md5(a):
    return hashlib.md5(memoryview(np.ascontiguousarray(a))).hexdigest()


def check(mag_path, n_reads, min_match=50, L=100, cov=30, seed=synth.DEFAULT_SEED, log=print):
    t0 = time.time()
    g, covc, runs = expected(n_reads, L, cov, seed, min_match)
    log("generator: genome of %d bases, %d runs of starts at most %d apart (longest %d bases, %d runs of a single start), %.1f s"
        % (len(g), len(runs), L - min_match, int((runs[:, 1] - runs[:, 0]).max()) + L, int((runs[:, 2] == 1).sum()), time.time() - t0))
    want = {}
    for (a, b, n), c in zip(runs, covc):
        s = g[a:b + L]
        key_f = (len(s), md5(s), md5(c), int(n))
        key_r = (len(s), md5(COMP[s[::-1]]), md5(c[::-1]), int(n))
        assert key_f not in want and key_r not in want, "two runs with the same sequence: the genome is not random enough for this check"
        want[key_f] = want[key_r] = (int(a), int(b))
    t1 = time.time()
    seen, n_rec, n_bases, problems = set(), 0, 0, []
    with open(mag_path, "rb") as fp:
        while True:
            head = fp.readline()
            if not head:
                break
            seq = fp.readline().rstrip(b"\n"); plus = fp.readline(); cv = fp.readline().rstrip(b"\n")
            n_rec += 1; n_bases += len(seq)
            f = head.rstrip(b"\n").split(b"\t")
            if not head.startswith(b"@") or len(f) != 4 or plus != b"+\n" or len(cv) != len(seq):
                problems.append("record %d is not a MAG record: %r" % (n_rec, head[:80])); break
            key = (len(seq), hashlib.md5(seq).hexdigest(), hashlib.md5(cv).hexdigest(), int(f[1]))
            if key not in want:
                problems.append("record %d (%s, %d bases, nsr %s) is no run of the generator (sequence, coverage or number of reads differ)" % (n_rec, f[0].decode(), len(seq), f[1].decode()))
                continue
            if f[2] != b"." or f[3] != b".":
                problems.append("record %d has neighbours (%r, %r): a run ends where nothing overlaps" % (n_rec, f[2][:40], f[3][:40]))
            if want[key] in seen:
                problems.append("the run starting at %d is printed twice" % want[key][0])
            seen.add(want[key])
    missing = [tuple(int(x) for x in r[:2]) for r in runs if (int(r[0]), int(r[1])) not in seen]
    for a, b in missing[:5]:
        problems.append("the run of starts %d .. %d (%d bases) is not in the MAG" % (a, b, b - a + L))
    log("MAG: %d records, %d bases, parsed and hashed in %.1f s; %d runs expected, %d matched exactly (sequence, coverage string, number of reads, no neighbours)"
        % (n_rec, n_bases, time.time() - t1, len(runs), len(seen)))
    for p in problems[:20]:
        log("  MISMATCH: " + p)
    return not problems and n_rec == len(runs)


if __name__ == "__main__":
    a = sys.argv[1:]
    ok = check(a[0], int(a[1]), int(a[2]) if len(a) > 2 else 50, int(a[3]) if len(a) > 3 else 100, int(a[4]) if len(a) > 4 else 30, int(a[5]) if len(a) > 5 else synth.DEFAULT_SEED)
    print("generator check of the MAG: %s" % ("EXACT" if ok else "FAILED"))
    sys.exit(0 if ok else 1)
