#!/usr/bin/env python3
"""Golden md5s of the reference's output at BASELINE config 0's size (1 M x 100 bp synthetic reads), made HERE with the
reference binary compiled in place (oracle/_ref/fermi): `fermi build` + `fermi unitig -l50 -t1` on error-free reads and
`fermi build` + `fermi correct -t1` + `fermi unitig -l50 -t1` on reads with 1 % substitutions (qualities from a seeded generator).  Only the md5s
are committed (tests/golden/md5_1m.json); tests/test_gpu_fullsize.py regenerates the same inputs on the GPU box.
Usage: python tests/golden/make_md5_1m.py            -> md5_1m.json
       python tests/golden/make_md5_1m.py 10000000   -> md5_10m.json  (the single-GPU size class of configs[1]; about two hours of the reference here,
                                                        the error-free and the raw chain side by side)"""
import hashlib, json, os, subprocess, sys, time
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import numpy as np
from fermi_amd import synth

REF = os.path.join(ROOT, "oracle", "_ref", "fermi")
N = 1_000_000


def _records(first_id, r, q):
    """the FASTQ text of reads first_id .. : b"@r%d\n%s\n+\n%s\n" per read, built as byte arrays (ids of equal digit count at a time)"""
    n, L = r.shape
    out = []
    ids = np.arange(first_id, first_id + n, dtype=np.int64)
    nd = np.where(ids == 0, 1, np.floor(np.log10(np.maximum(ids, 1))).astype(np.int64) + 1)
    nd = np.where(10 ** (nd - 1) > ids, nd - 1, np.where(10 ** nd <= ids, nd + 1, nd))      # (log10 at the powers of ten)
    nd = np.maximum(nd, 1)
    lo = 0
    while lo < n:
        hi = lo + int(np.searchsorted(nd[lo:], nd[lo], side="right"))
        d, m = int(nd[lo]), hi - lo
        blk = np.empty((m, 2 + d + 1 + L + 3 + L + 1), dtype=np.uint8)
        blk[:, 0] = ord("@"); blk[:, 1] = ord("r")
        v = ids[lo:hi].copy()
        for k in range(d - 1, -1, -1):
            blk[:, 2 + k] = ord("0") + v % 10
            v //= 10
        blk[:, 2 + d] = 10
        blk[:, 3 + d:3 + d + L] = r[lo:hi]
        blk[:, 3 + d + L] = 10; blk[:, 4 + d + L] = ord("+"); blk[:, 5 + d + L] = 10
        blk[:, 6 + d + L:6 + d + 2 * L] = q[lo:hi]
        blk[:, 6 + d + 2 * L] = 10
        out.append(blk.tobytes())
        lo = hi
    return b"".join(out)


def write_fastq(path, err, with_random_quals, n=None, reads_of=None):
    """reads_of(start, count) -> nt6 reads (count x 100) of the set: the numpy generator by default; a test on a GPU box passes the torch form"""
    N = n or globals()["N"]
    lut = np.frombuffer(b"$ACGTN", dtype=np.uint8)
    rng = np.random.default_rng(5)
    if reads_of is None:
        gen = synth.genome(synth.DEFAULT_SEED, N, 100, 30)
        reads_of = lambda s, c: synth.reads(synth.DEFAULT_SEED, N, 100, 30, err, start=s, count=c, gen=gen)
    with open(path, "wb") as fp:
        for s in range(0, N, 250_000):
            r = lut[reads_of(s, 250_000)]
            q = rng.integers(33 + 5, 33 + 41, size=(250_000, 100)).astype(np.uint8) if with_random_quals else np.full((250_000, 100), ord("I"), dtype=np.uint8)
            fp.write(_records(s, r, q))


def md5_of(cmd):
    h = hashlib.md5()
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
    n = 0
    for blk in iter(lambda: p.stdout.read(1 << 24), b""):
        h.update(blk); n += len(blk)
    assert p.wait() == 0, cmd
    return h.hexdigest(), n


def clean_chain(d, n, out, t):
    write_fastq(d + "/clean.fq", 0.0, False, n)
    subprocess.check_call([REF, "build", "-fo", d + "/clean.fmd", d + "/clean.fq"], stderr=subprocess.DEVNULL)
    out["clean_fmd"] = md5_of(["cat", d + "/clean.fmd"])
    out["unitig_l50_t1"] = md5_of([REF, "unitig", "-l50", "-t1", d + "/clean.fmd"])
    print("unitig done", time.time() - t, out["unitig_l50_t1"], flush=True)


def raw_chain(d, n, out, t):
    write_fastq(d + "/raw.fq", 0.01, True, n)
    subprocess.check_call([REF, "build", "-fo", d + "/raw.fmd", d + "/raw.fq"], stderr=subprocess.DEVNULL)
    out["raw_fmd"] = md5_of(["cat", d + "/raw.fmd"])
    out["correct_t1"] = md5_of([REF, "correct", "-t1", d + "/raw.fmd", d + "/raw.fq"])
    print("correct done", time.time() - t, out["correct_t1"], flush=True)
    out["unitig_raw_l50_t1"] = md5_of([REF, "unitig", "-l50", "-t1", d + "/raw.fmd"])   # reads with errors: forks, tips, back-bifurcations
    print("unitig on raw reads done", time.time() - t, out["unitig_raw_l50_t1"], flush=True)


if __name__ == "__main__":
    import threading
    n = int(sys.argv[1]) if len(sys.argv) > 1 else N
    assert n % 250_000 == 0
    tag = "%dm" % (n // 1_000_000)
    d = "/tmp/fmd_md5_" + tag; os.makedirs(d, exist_ok=True)
    out = {"n_reads": n, "seed": synth.DEFAULT_SEED, "made_with": "oracle/_ref/fermi (the reference compiled in place)"}
    t = time.time()
    a, b = {}, {}
    th = [threading.Thread(target=clean_chain, args=(d, n, a, t)), threading.Thread(target=raw_chain, args=(d, n, b, t))]   # (the work is in child processes)
    for x in th:
        x.start()
    for x in th:
        x.join()
    for k in ("clean_fmd", "unitig_l50_t1"):
        out[k] = a[k]
    for k in ("raw_fmd", "correct_t1", "unitig_raw_l50_t1"):
        out[k] = b[k]
    json.dump(out, open(os.path.join(HERE, "md5_%s.json" % tag), "w"), indent=1)
