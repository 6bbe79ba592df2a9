#!/usr/bin/env python3
"""`fermi correct` at bench scale against the reference BINARY (oracle/_ref/fermi travels with the repo):
N reads with 1 % substitutions, index built by `fermi-amd build`, then `fermi-amd correct -t T` vs
`fermi correct -t T` -- corrected bases, qualities and headers must be byte-identical.
Usage: python tools/validate_correct_large.py [n_reads=10000000] [threads=64]"""
import hashlib
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fermi_amd import synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
T = sys.argv[2] if len(sys.argv) > 2 else "64"
REF = os.path.join(ROOT, "oracle", "_ref", "fermi")
AMD = os.path.join(ROOT, "fermi_amd", "bin", "fermi-amd")
D = "/tmp/fmd_validate_correct"
os.makedirs(D, exist_ok=True)
t0 = time.time()
with open(D + "/r.fq", "wb") as fp:
    tab = bytes(b"$ACGTN")
    import numpy as np
    lut = np.frombuffer(tab, dtype=np.uint8)
    for s in range(0, n, 1_000_000):
        c = min(1_000_000, n - s)
        r = lut[synth.reads(synth.DEFAULT_SEED, n, 100, 30, 0.01, start=s, count=c)]
        q = b"I" * 100
        fp.write(b"".join(b"@r%d\n%s\n+\n%s\n" % (s + i, r[i].tobytes(), q) for i in range(c)))
print("fastq of %d reads: %.0f s" % (n, time.time() - t0), flush=True)


def run(cmd, out=None):
    t = time.time()
    with open(out, "wb") if out else open(os.devnull, "wb") as fo:
        subprocess.check_call(cmd, stdout=fo, stderr=subprocess.DEVNULL)
    return time.time() - t


def md5(p):
    h = hashlib.md5()
    with open(p, "rb") as f:
        for b in iter(lambda: f.read(1 << 24), b""):
            h.update(b)
    return h.hexdigest()


print("fermi-amd build: %.1f s" % run([AMD, "build", "-fo", D + "/a.fmd", D + "/r.fq"]), flush=True)
ta = run([AMD, "correct", "-t", T, D + "/a.fmd", D + "/r.fq"], D + "/a.ec.fq")
print("fermi-amd correct -t%s: %.1f s" % (T, ta), flush=True)
tr = run([REF, "correct", "-t", T, D + "/a.fmd", D + "/r.fq"], D + "/b.ec.fq")
print("fermi     correct -t%s: %.1f s" % (T, tr), flush=True)
a, b = md5(D + "/a.ec.fq"), md5(D + "/b.ec.fq")
print("corrected FASTQ: %d bytes, md5 %s / %s -> %s" % (os.path.getsize(D + "/b.ec.fq"), a, b, "IDENTICAL" if a == b else "MISMATCH"))
sys.exit(0 if a == b else 1)
