#!/usr/bin/env python3
"""End-to-end check on the GPU box against the reference BINARY (oracle/_ref/fermi travels with the
repo): build / unitig / correct / exact on a synthetic read set with sequencing errors, outputs
compared byte for byte.  Usage: python tools/validate_large.py [n_reads] [err]"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fermi_amd import synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
err = float(sys.argv[2]) if len(sys.argv) > 2 else 0.01
REF = os.path.join(ROOT, "oracle", "_ref", "fermi")
AMD = os.path.join(ROOT, "fermi_amd", "bin", "fermi-amd")
T = "/tmp/fmd_validate"
os.makedirs(T, exist_ok=True)
reads = synth.reads(synth.DEFAULT_SEED, n, 100, 30, err)
synth.to_fastq(reads, T + "/r.fq")


def run(cmd, out):
    t0 = time.time()
    with open(out, "wb") as fo:
        subprocess.check_call(cmd, stdout=fo, stderr=subprocess.DEVNULL)
    return time.time() - t0


def same(a, b):
    return open(a, "rb").read() == open(b, "rb").read()


ok = True
ta = run([AMD, "build", "-fo", T + "/a.fmd", T + "/r.fq"], T + "/a.log")
tr = run([REF, "build", "-fo", T + "/b.fmd", T + "/r.fq"], T + "/b.log")
r = same(T + "/a.fmd", T + "/b.fmd"); ok &= r
print("build   : identical=%s  fermi-amd %.1fs  fermi %.1fs" % (r, ta, tr), flush=True)
for name, a_args, r_args in [
        ("unitig", ["unitig", "-l50", T + "/a.fmd"], ["unitig", "-l50", "-t1", T + "/a.fmd"]),
        ("correct", ["correct", T + "/a.fmd", T + "/r.fq"], ["correct", "-t1", T + "/a.fmd", T + "/r.fq"]),
        ("exact", ["exact", T + "/a.fmd", T + "/r.fq"], ["exact", T + "/a.fmd", T + "/r.fq"])]:
    ta = run([AMD] + a_args, T + "/a." + name)
    tr = run([REF] + r_args, T + "/b." + name)
    r = same(T + "/a." + name, T + "/b." + name); ok &= r
    print("%-8s: identical=%s  fermi-amd %.1fs  fermi(-t1) %.1fs  (%d bytes)" % (name, r, ta, tr, os.path.getsize(T + "/b." + name)), flush=True)
print("ALL IDENTICAL" if ok else "MISMATCH")
sys.exit(0 if ok else 1)
