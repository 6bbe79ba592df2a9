#!/usr/bin/env python3
"""Golden md5s of the reference's output at BASELINE config 0's size (1 M x 100 bp synthetic reads), made HERE with the
reference binary compiled in place (oracle/_ref/fermi): `fermi build` + `fermi unitig -l50 -t1` on error-free reads and
`fermi build` + `fermi correct -t1` + `fermi unitig -l50 -t1` on reads with 1 % substitutions (qualities from a seeded generator).  Only the md5s
are committed (tests/golden/md5_1m.json); tests/test_gpu_fullsize.py regenerates the same inputs on the GPU box.
Usage: python tests/golden/make_md5_1m.py"""
import hashlib, json, os, subprocess, sys, time
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import numpy as np
from fermi_amd import synth

REF = os.path.join(ROOT, "oracle", "_ref", "fermi")
N = 1_000_000


def write_fastq(path, err, with_random_quals):
    lut = np.frombuffer(b"$ACGTN", dtype=np.uint8)
    rng = np.random.default_rng(5)
    with open(path, "wb") as fp:
        for s in range(0, N, 250_000):
            r = lut[synth.reads(synth.DEFAULT_SEED, N, 100, 30, err, start=s, count=250_000)]
            q = rng.integers(33 + 5, 33 + 41, size=(250_000, 100)).astype(np.uint8) if with_random_quals else np.full((250_000, 100), ord("I"), dtype=np.uint8)
            fp.write(b"".join(b"@r%d\n%s\n+\n%s\n" % (s + i, r[i].tobytes(), q[i].tobytes()) for i in range(250_000)))


def md5_of(cmd):
    h = hashlib.md5()
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
    n = 0
    for blk in iter(lambda: p.stdout.read(1 << 24), b""):
        h.update(blk); n += len(blk)
    assert p.wait() == 0, cmd
    return h.hexdigest(), n


if __name__ == "__main__":
    d = "/tmp/fmd_md5_1m"; os.makedirs(d, exist_ok=True)
    out = {"n_reads": N, "seed": synth.DEFAULT_SEED, "made_with": "oracle/_ref/fermi (the reference compiled in place)"}
    t = time.time()
    write_fastq(d + "/clean.fq", 0.0, False)
    subprocess.check_call([REF, "build", "-fo", d + "/clean.fmd", d + "/clean.fq"], stderr=subprocess.DEVNULL)
    out["clean_fmd"] = md5_of(["cat", d + "/clean.fmd"])
    out["unitig_l50_t1"] = md5_of([REF, "unitig", "-l50", "-t1", d + "/clean.fmd"])
    print("unitig done", time.time() - t, out["unitig_l50_t1"], flush=True)
    write_fastq(d + "/raw.fq", 0.01, True)
    subprocess.check_call([REF, "build", "-fo", d + "/raw.fmd", d + "/raw.fq"], stderr=subprocess.DEVNULL)
    out["raw_fmd"] = md5_of(["cat", d + "/raw.fmd"])
    out["correct_t1"] = md5_of([REF, "correct", "-t1", d + "/raw.fmd", d + "/raw.fq"])
    print("correct done", time.time() - t, out["correct_t1"], flush=True)
    out["unitig_raw_l50_t1"] = md5_of([REF, "unitig", "-l50", "-t1", d + "/raw.fmd"])   # reads with errors: forks, tips, back-bifurcations
    print("unitig on raw reads done", time.time() - t, out["unitig_raw_l50_t1"], flush=True)
    json.dump(out, open(os.path.join(HERE, "md5_1m.json"), "w"), indent=1)
