#!/usr/bin/env python3
"""tests/golden/gen_rule_20k.mag.gz: `fermi build` + `fermi unitig -l50 -t1` of the REFERENCE (oracle/_ref/fermi, compiled in place) on 20 000 error-free
synthetic reads at 8-fold coverage (346 unitigs: gaps wider than 50 bases are common at 8-fold) -- the MAG that pins the rule of tools/mag_vs_generator.py,
which checks `unitig` on error-free reads exactly from what the generator knows at sizes the reference cannot reach (config 5)."""
import gzip, os, subprocess, sys, tempfile
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import numpy as np
from fermi_amd import synth
N, L, C = 20000, 100, 8
REF = os.path.join(ROOT, "oracle", "_ref", "fermi")
r = synth.reads(synth.DEFAULT_SEED, N, L, C, 0.0)
lut = np.frombuffer(b"$ACGTN", dtype=np.uint8)
with tempfile.TemporaryDirectory() as d:
    with open(d + "/r.fq", "wb") as f:
        for i in range(N):
            f.write(b"@r%d\n" % i + lut[r[i]].tobytes() + b"\n+\n" + b"I" * L + b"\n")
    subprocess.check_call([REF, "build", "-fo", d + "/r.fmd", d + "/r.fq"], stderr=subprocess.DEVNULL)
    mag = subprocess.run([REF, "unitig", "-l50", "-t1", d + "/r.fmd"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout
with gzip.GzipFile(os.path.join(HERE, "gen_rule_20k.mag.gz"), "wb", mtime=0) as f:
    f.write(mag)
print(len(mag), "bytes of MAG")
