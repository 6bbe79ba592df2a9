#!/usr/bin/env python3
"""A circular genome tiled by reads: its unitig is a LOOP (unitig.c:247, a>>b>>c>>a) -- the one piece of a walk's own history that matters on a
static chain (host/unitig_walk.c: hops).  Made HERE with the reference binary compiled in place: circle.fq.gz, circle.fmd, circle.mag.gz
(`fermi unitig -l40 -t1`).  Two circles (3000 and 1700 bases, reads of 80 bases every 7 / 5 positions, both strands alternating) and a
linear piece, so that loops and open unitigs sit in one table.  Usage: python tests/golden/make_golden_circle.py"""
import gzip, os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import numpy as np
from fermi_amd import synth
REF = os.path.join(ROOT, "oracle", "_ref", "fermi")
lut = np.frombuffer(b"$ACGTN", dtype=np.uint8)
reads = []
for gi, (G, step) in enumerate(((3000, 7), (1700, 5))):
    gen = (1 + (synth.rnd(synth.DEFAULT_SEED + 900 + gi, 1, np.arange(G, dtype=np.uint64)) >> np.uint64(62))).astype(np.uint8)
    circ = np.concatenate([gen, gen[:200]])
    for k, p in enumerate(range(0, G, step)):
        r = circ[p:p + 80]
        reads.append(r if k % 2 == 0 else (5 - r)[::-1])
lin = (1 + (synth.rnd(synth.DEFAULT_SEED + 990, 1, np.arange(2500, dtype=np.uint64)) >> np.uint64(62))).astype(np.uint8)
for k, p in enumerate(range(0, 2500 - 80, 6)):
    r = lin[p:p + 80]
    reads.append(r if k % 3 else (5 - r)[::-1])
rng = np.random.default_rng(8)
order = rng.permutation(len(reads))
fq = b"".join(b"@c%d\n%s\n+\n%s\n" % (i, lut[reads[j]].tobytes(), b"I" * 80) for i, j in enumerate(order))
with gzip.GzipFile(os.path.join(HERE, "circle.fq.gz"), "wb", mtime=0) as f:
    f.write(fq)
subprocess.run([REF, "build", "-fo", os.path.join(HERE, "circle.fmd"), os.path.join(HERE, "circle.fq.gz")], check=True, stderr=subprocess.DEVNULL)
mag = subprocess.run([REF, "unitig", "-l40", "-t1", os.path.join(HERE, "circle.fmd")], check=True, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout
with gzip.GzipFile(os.path.join(HERE, "circle.mag.gz"), "wb", mtime=0) as f:
    f.write(mag)
print(len(reads), "reads;", mag.count(b"\n@") + 1, "unitigs;", [l.split(b"\t")[:2] for l in mag.split(b"\n") if l.startswith(b"@")][:8])
