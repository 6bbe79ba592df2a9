#!/usr/bin/env python3
"""Golden vectors for `fermi remap` (smem.c:114-394), made with the compiled reference
(oracle/_ref/fermi): a small paired-end read set, its index, its seqsort rank file, a contig file
(unitigs of the reads, the true genome, a chimera, a reverse-complemented piece) and what `fermi remap -t1`
prints for it in the three modes (no rank file; rank file; rank file + break at low paired coverage).

    python tests/golden/make_golden_remap.py      (needs /root/reference built: make -C oracle ref)
"""
import gzip
import json
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import refbind  # noqa: E402

FERMI = refbind.REF_BIN
TMP = "/tmp/fmd_golden_remap"
os.makedirs(TMP, exist_ok=True)


def run(cmd, out=None):
    with open(out, "wb") if out else open(os.devnull, "wb") as fo:
        p = subprocess.run(cmd, stdout=fo, stderr=subprocess.PIPE, check=True)
    return p.stderr.decode()


def main():
    rng = np.random.default_rng(20260928)
    L, n_pairs = 60, 700
    genome = rng.integers(1, 5, 6000).astype(np.uint8)
    tab = np.frombuffer(b"$ACGTN", dtype=np.uint8)

    def txt(a):
        return tab[a].tobytes().decode()
    reads = []
    for _ in range(n_pairs):
        ins = int(rng.normal(300, 30))
        ins = max(2 * L, min(ins, 500))
        p = int(rng.integers(0, len(genome) - ins))
        a, b = genome[p:p + L], (5 - genome[p + ins - L:p + ins])[::-1]
        if rng.integers(0, 2):
            a, b = b, a
        reads += [a, b]
    # a few duplicated reads (intervals of size > 1) and a far-away mate
    reads[10] = reads[4].copy(); reads[11] = reads[5].copy()
    reads[21] = (5 - genome[5000:5000 + L])[::-1]
    fq = os.path.join(TMP, "pairs.fq")
    with open(fq, "w") as f:
        for i, r in enumerate(reads):
            f.write("@p%d/%d\n%s\n+\n%s\n" % (i // 2, i % 2 + 1, txt(r), "I" * len(r)))
    fmd = os.path.join(HERE, "pairs.fmd")
    run([FERMI, "build", "-fo", fmd, fq])
    rank = os.path.join(HERE, "pairs.rank")
    run([FERMI, "seqsort", fmd], rank)
    run([FERMI, "unitig", "-l30", "-t1", fmd], os.path.join(TMP, "pairs.mag"))
    # contigs: the unitigs (MAG records: "<name>\t<nsr>\t<nei>\t<nei>"), then hand-made ones
    ctg = open(os.path.join(TMP, "pairs.mag")).read()
    chim = np.concatenate([genome[100:900], (5 - genome[3000:3777])[::-1], genome[4000:4300]])
    # (no contig with N: the reference's iterator does not terminate on a base absent from the index)
    rc_part = (5 - genome[1500:2400])[::-1]
    ctg += "@genome 7 whole thing\n%s\n+\n%s\n" % (txt(genome), "I" * len(genome))
    ctg += "@chimera\n%s\n+\n%s\n" % (txt(chim), "I" * len(chim))
    ctg += "@revcomp 3x\n%s\n+\n%s\n" % (txt(rc_part), "I" * len(rc_part))
    ctg += "@tiny 12 short\n%s\n+\n%s\n" % (txt(genome[77:100]), "I" * 23)
    cfq = os.path.join(TMP, "pairs_contigs.fq")
    open(cfq, "w").write(ctg)
    err = {}
    err["u"] = run([FERMI, "remap", fmd, cfq], os.path.join(TMP, "pairs.remap_u"))
    err["p"] = run([FERMI, "remap", "-l", "20", "-D", "600", "-r", rank, fmd, cfq], os.path.join(TMP, "pairs.remap_p"))
    err["c"] = run([FERMI, "remap", "-l", "20", "-D", "600", "-c", "2", "-r", rank, fmd, cfq], os.path.join(TMP, "pairs.remap_c"))
    err["d"] = run([FERMI, "remap", "-D", "310", "-c", "1", "-r", rank, fmd, cfq], os.path.join(TMP, "pairs.remap_d"))   # default -l50, tight insert cap
    # `fermi exact` with long queries (contigs against the read index), both self_match settings
    run([FERMI, "exact", fmd, cfq], os.path.join(TMP, "pairs.exact_contigs"))
    run([FERMI, "exact", "-s", fmd, cfq], os.path.join(TMP, "pairs.exact_s_contigs"))
    for name in ("pairs.fq", "pairs_contigs.fq", "pairs.remap_u", "pairs.remap_p", "pairs.remap_c", "pairs.remap_d", "pairs.exact_contigs", "pairs.exact_s_contigs"):
        with gzip.open(os.path.join(HERE, name + ".gz"), "wb", 9) as f:
            f.write(open(os.path.join(TMP, name), "rb").read())
    json.dump({k: [l for l in v.split("\n") if "fm6_remap" in l] for k, v in err.items()}, open(os.path.join(HERE, "pairs.remap_stderr.json"), "w"), indent=1)
    # add the new files to the manifest written by make_golden.py
    import hashlib
    mp = os.path.join(HERE, "MANIFEST.json")
    man = json.load(open(mp))
    for fn in sorted(os.listdir(HERE)):
        if fn.startswith("pairs") and fn.endswith((".fmd", ".gz", ".rank")):
            man["files"][fn] = {"md5": hashlib.md5(open(os.path.join(HERE, fn), "rb").read()).hexdigest(), "bytes": os.path.getsize(os.path.join(HERE, fn))}
    json.dump(man, open(mp, "w"), indent=1, sort_keys=True)
    print({k: v.strip().split("\n")[-1] for k, v in err.items()})
    for name in ("pairs.remap_u", "pairs.remap_p", "pairs.remap_c", "pairs.remap_d"):
        print(name, os.path.getsize(os.path.join(TMP, name)))


if __name__ == "__main__":
    main()
