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

# ---- paired-end part: seqsort, unitig -r, remap (smem.c:114-394) on a paired read set made from the
# same genome (FR pairs, insert ~ N(300, 30), every other read reverse-complemented, same error rate)
import numpy as np  # noqa: E402
g = synth.genome(synth.DEFAULT_SEED, n, 100, 30)
rng = np.random.default_rng(7)
npairs = n // 2
ins = np.clip(rng.normal(300, 30, npairs).astype(np.int64), 200, 500)
pos = rng.integers(0, len(g) - 500, npairs)
pr = np.empty((2 * npairs, 100), dtype=np.uint8)
idx = np.arange(100)
pr[0::2] = g[pos[:, None] + idx[None, :]]
pr[1::2] = (5 - g[(pos + ins)[:, None] - 1 - idx[None, :]])
if err > 0:
    m = rng.random(pr.shape) < err
    pr[m] = 1 + (pr[m] - 1 + rng.integers(1, 4, int(m.sum()))) % 4
synth.to_fastq(pr, T + "/p.fq")
run([AMD, "build", "-fo", T + "/p.fmd", T + "/p.fq"], T + "/p.log")
for name, a_args, r_args in [
        ("seqsort", ["seqsort", T + "/p.fmd"], ["seqsort", "-t8", T + "/p.fmd"]),
        ("unitig-r", ["unitig", "-l50", "-r", T + "/a.seqsort", T + "/p.fmd"], ["unitig", "-l50", "-t1", "-r", T + "/b.seqsort", T + "/p.fmd"]),
        ("remap-u", ["remap", T + "/p.fmd", T + "/b.unitig-r"], ["remap", T + "/p.fmd", T + "/b.unitig-r"]),
        ("remap-p", ["remap", "-r", T + "/b.seqsort", T + "/p.fmd", T + "/b.unitig-r"], ["remap", "-r", T + "/b.seqsort", T + "/p.fmd", T + "/b.unitig-r"]),
        ("remap-c", ["remap", "-c", "2", "-D", "420", "-r", T + "/b.seqsort", T + "/p.fmd", T + "/b.unitig-r"],
         ["remap", "-c", "2", "-D", "420", "-r", T + "/b.seqsort", T + "/p.fmd", T + "/b.unitig-r"])]:
    ta = run([AMD] + a_args, T + "/a." + name)
    tr = run([REF] + r_args, T + "/b." + name)
    r = same(T + "/a." + name, T + "/b." + name); ok &= r
    print("%-8s: identical=%s  fermi-amd %.1fs  fermi(-t1) %.1fs  (%d bytes)" % (name, r, ta, tr, os.path.getsize(T + "/b." + name)), flush=True)
print("ALL IDENTICAL" if ok else "MISMATCH")
sys.exit(0 if ok else 1)
