#!/usr/bin/env python3
"""End to end at the size the north star is quoted on: `fermi-amd build`, `unitig -l50` and `correct` on
N x 100-bp synthetic reads (30x), error-free and with 1 % substitutions, stage times under FMD_TIMING, wall clock and peak RSS of
every command; and -- where oracle/_ref/fermi travelled -- the REFERENCE binary's `-t16` / `-t1` wall clock on a 30x set of M reads
made the same way (the reference needs hours at 50 M), with the md5 of either output.
Usage: python tools/time_e2e.py [n_reads=50000000] [ref_subset=1000000] [dir=/tmp/fmd_e2e]"""
import hashlib, os, resource, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from fermi_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
m_ref = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
D = sys.argv[3] if len(sys.argv) > 3 else "/tmp/fmd_e2e"
os.makedirs(D, exist_ok=True)
AMD = os.path.join(ROOT, "fermi_amd", "bin", "fermi-amd")
REF = os.path.join(ROOT, "oracle", "_ref", "fermi")
lut = np.frombuffer(b"$ACGTN", dtype=np.uint8)


GEN = synth.genome(synth.DEFAULT_SEED, n, 100, 30)


def write_fastq(path, n_total, count, err, qual_rng, gen=None):
    """reads 0 .. count-1 of the n_total-read set; fixed-width names so that a chunk is one array operation"""
    t = time.time()
    with open(path, "wb") as fp:
        for s in range(0, count, 1_000_000):
            c = min(1_000_000, count - s)
            r = lut[synth.reads(synth.DEFAULT_SEED, n_total, 100, 30, err, start=s, count=c, gen=GEN if gen is None else gen)]
            rec = np.empty((c, 11 + 1 + 100 + 3 + 100 + 1), dtype=np.uint8)
            rec[:, 0] = ord("@"); rec[:, 1] = ord("r")
            ids = np.arange(s, s + c, dtype=np.int64)
            for d in range(9):
                rec[:, 10 - d] = 48 + (ids // 10 ** d) % 10
            rec[:, 11] = 10; rec[:, 12:112] = r; rec[:, 112] = 10; rec[:, 113] = ord("+"); rec[:, 114] = 10
            rec[:, 115:215] = ord("I") if qual_rng is None else qual_rng.integers(33 + 5, 33 + 41, size=(c, 100)).astype(np.uint8)
            rec[:, 215] = 10
            fp.write(rec.tobytes())
    print("wrote %s: %d reads (e = %g), %.1f GB, %.0f s" % (path, count, err, os.path.getsize(path) / 1e9, time.time() - t), flush=True)


def run(cmd, out, env_extra=None):
    env = dict(os.environ, FMD_TIMING="1", **(env_extra or {}))
    r0 = resource.getrusage(resource.RUSAGE_CHILDREN).ru_maxrss
    t = time.time()
    p = subprocess.run(cmd, stdout=open(out, "wb") if out else subprocess.DEVNULL, stderr=subprocess.PIPE, env=env)
    dt = time.time() - t
    rss = resource.getrusage(resource.RUSAGE_CHILDREN).ru_maxrss
    print("%-46s %8.1f s   rc %d   peak RSS of the children so far %.1f GB (before this command %.1f)" % (" ".join(os.path.basename(c) for c in cmd[:4]), dt, p.returncode, rss / 1e6, r0 / 1e6), flush=True)
    lines = [l for l in p.stderr.decode(errors="replace").splitlines() if "M::" in l and "part of" not in l]
    print("\n".join("    " + l for l in (lines[:4] + ["    ..."] + lines[-10:] if len(lines) > 16 else lines)), flush=True)
    assert p.returncode == 0, p.stderr.decode(errors="replace")[-2000:]
    return dt


def md5(path):
    h = hashlib.md5()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 24), b""):
            h.update(blk)
    return h.hexdigest()


for tag, err in (("clean", 0.0), ("raw", 0.01)):
    print("==== %d reads, e = %g" % (n, err), flush=True)
    fq, fmd = "%s/%s.fq" % (D, tag), "%s/%s.fmd" % (D, tag)
    write_fastq(fq, n, n, err, None if err == 0 else np.random.default_rng(5))
    run([AMD, "build", "-fo", fmd, fq], None)
    t_u = run([AMD, "unitig", "-l50", fmd], "%s/%s.mag" % (D, tag))
    print("    unitig: %.3e reads/s end to end, MAG %d bytes md5 %s" % (n / t_u, os.path.getsize("%s/%s.mag" % (D, tag)), md5("%s/%s.mag" % (D, tag))), flush=True)
    if err > 0:
        t_c = run([AMD, "correct", "-t16", fmd, fq], "%s/%s.ec.fq" % (D, tag))
        print("    correct: %.3e reads/s end to end, output %d bytes md5 %s" % (n / t_c, os.path.getsize("%s/%s.ec.fq" % (D, tag)), md5("%s/%s.ec.fq" % (D, tag))), flush=True)
    for f in (fq, fmd, "%s/%s.mag" % (D, tag), "%s/%s.ec.fq" % (D, tag)):
        if os.path.exists(f):
            os.remove(f)
    # ---- the reference on a subset of the same read set (its own index of the subset), and the product on the same subset
    if os.path.exists(REF) and m_ref > 0:
        m = min(m_ref, n)
        sfq, sfmd = "%s/%s_sub.fq" % (D, tag), "%s/%s_sub.fmd" % (D, tag)
        write_fastq(sfq, m, m, err, None if err == 0 else np.random.default_rng(5), gen=synth.genome(synth.DEFAULT_SEED, m, 100, 30))   # a 30x set of its own
        run([AMD, "build", "-fo", sfmd, sfq], None)
        t_a = run([AMD, "unitig", "-l50", sfmd], D + "/a.mag")
        t = time.time()
        subprocess.run([REF, "unitig", "-l50", "-t16", sfmd], stdout=open(D + "/r16.mag", "wb"), stderr=subprocess.DEVNULL)
        t_r16 = time.time() - t
        t = time.time()
        subprocess.run([REF, "unitig", "-l50", "-t1", sfmd], stdout=open(D + "/r1.mag", "wb"), stderr=subprocess.DEVNULL)
        t_r1 = time.time() - t
        print("    subset of %d reads, unitig -l50: fermi-amd %.1f s; fermi -t16 %.1f s (non-deterministic output), fermi -t1 %.1f s; fermi-amd MAG %s fermi -t1's"
              % (m, t_a, t_r16, t_r1, "==" if md5(D + "/a.mag") == md5(D + "/r1.mag") else "!="), flush=True)
        if err > 0:
            t_a = run([AMD, "correct", "-t16", sfmd, sfq], D + "/a.ec.fq")
            t = time.time()
            subprocess.run([REF, "correct", "-t16", sfmd, sfq], stdout=open(D + "/r.ec.fq", "wb"), stderr=subprocess.DEVNULL)
            t_r = time.time() - t
            print("    subset of %d reads, correct: fermi-amd %.1f s; fermi -t16 %.1f s; outputs %s" % (m, t_a, t_r, "IDENTICAL" if md5(D + "/a.ec.fq") == md5(D + "/r.ec.fq") else "DIFFERENT"), flush=True)
        for f in (sfq, sfmd, D + "/a.mag", D + "/r16.mag", D + "/r1.mag", D + "/a.ec.fq", D + "/r.ec.fq"):
            if os.path.exists(f):
                os.remove(f)
