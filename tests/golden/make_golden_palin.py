#!/usr/bin/env python3
"""Reads that are their OWN reverse complement.  The FMD index holds both strands of every read; for such a read the two are one string, its two
rows (ids 2i and 2i + 1) carry one `$read$` interval with k[0] == k[1], and whoever maps an interval back to a row (the link passes: the smallest id
wins) meets that row for the read AND for its other strand.  Two genomes with a reverse-complement palindrome of exactly one read length (80) in the
middle, tiled so that one read is the palindrome itself -- once in the first genome, three identical copies in the second -- and its neighbours on both
sides overlap it.  Made HERE with the reference binary compiled in place: palin.fq.gz, palin.fmd, palin.mag.gz (`fermi unitig -l40 -t1`).
Usage: python tests/golden/make_golden_palin.py"""
import gzip, os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import numpy as np
from fermi_amd import synth
REF = os.path.join(ROOT, "oracle", "_ref", "fermi")
lut = np.frombuffer(b"$ACGTN", dtype=np.uint8)
def rnd(seed, n): return (1 + (synth.rnd(synth.DEFAULT_SEED + seed, 1, np.arange(n, dtype=np.uint64)) >> np.uint64(62))).astype(np.uint8)
reads = []
for gi, copies in enumerate((1, 3)):
    half = rnd(700 + gi, 40)
    pal = np.concatenate([half, (5 - half)[::-1]])                      # 80 bases, its own reverse complement
    assert (pal == (5 - pal)[::-1]).all()
    g = np.concatenate([rnd(710 + gi, 600), pal, rnd(720 + gi, 600)])
    for k, p in enumerate(range(0, len(g) - 80 + 1, 8)):                # 600 is a multiple of 8: one read starts at the palindrome
        r = g[p:p + 80]
        for _ in range(copies if p == 600 else 1):
            reads.append(r if k % 2 == 0 else (5 - r)[::-1])
rng = np.random.default_rng(11)
order = rng.permutation(len(reads))
fq = b"".join(b"@p%d\n%s\n+\n%s\n" % (i, lut[reads[j]].tobytes(), b"I" * 80) for i, j in enumerate(order))
with gzip.GzipFile(os.path.join(HERE, "palin.fq.gz"), "wb", mtime=0) as f:
    f.write(fq)
subprocess.run([REF, "build", "-fo", os.path.join(HERE, "palin.fmd"), os.path.join(HERE, "palin.fq.gz")], check=True, stderr=subprocess.DEVNULL)
mag = subprocess.run([REF, "unitig", "-l40", "-t1", os.path.join(HERE, "palin.fmd")], check=True, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout
with gzip.GzipFile(os.path.join(HERE, "palin.mag.gz"), "wb", mtime=0) as f:
    f.write(mag)
print(len(reads), "reads;", mag.count(b"\n@") + 1, "unitigs;", [l.split(b"\t")[:2] for l in mag.split(b"\n") if l.startswith(b"@")][:8])
