#!/usr/bin/env python3
"""Golden md5s of the reference on a REPEAT-RICH, RAGGED read set at scale: 2 M reads of 70-150 bp, 30x of a genome
with 5 % of its positions in repeat families (2-50 copies of 300-5000 bp), 1 % substitutions, 1 % exact duplicates and 1 % proper
substrings of other reads (contained reads) -- where fm6_get_nei's slow paths live (unitig.c:77-91, :128-135, :158-176).  Made HERE with
the reference binary compiled in place (oracle/_ref/fermi): `build`, `unitig -l50 -t1`, `correct -t1`.  Only the md5s are committed
(tests/golden/md5_repeat.json); tests/test_gpu_fullsize.py regenerates the same reads on the GPU box (fermi_amd/synth.py).
Usage: python tests/golden/make_md5_repeat.py"""
import hashlib, json, os, subprocess, sys, time
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import numpy as np
from fermi_amd import synth

REF = os.path.join(ROOT, "oracle", "_ref", "fermi")
N = 2_000_000
SEED = synth.DEFAULT_SEED + 404


def make_reads():
    gen = synth.repeat_genome(SEED, N * 110 // 30)
    return synth.ragged_reads(SEED, N, gen)


def write_fastq(path, reads):
    lut = np.frombuffer(b"$ACGTN", dtype=np.uint8)
    rng = np.random.default_rng(6)
    with open(path, "wb") as fp:
        for s in range(0, len(reads), 100_000):
            blk = []
            for i in range(s, min(len(reads), s + 100_000)):
                q = rng.integers(33 + 5, 33 + 41, size=len(reads[i])).astype(np.uint8)
                blk.append(b"@r%d\n%s\n+\n%s\n" % (i, lut[reads[i]].tobytes(), q.tobytes()))
            fp.write(b"".join(blk))


def md5_of(cmd):
    h = hashlib.md5()
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
    n = 0
    for blk in iter(lambda: p.stdout.read(1 << 24), b""):
        h.update(blk); n += len(blk)
    assert p.wait() == 0, cmd
    return h.hexdigest(), n


if __name__ == "__main__":
    d = "/tmp/fmd_md5_repeat"; os.makedirs(d, exist_ok=True)
    out = {"n_reads": N, "seed": SEED, "made_with": "oracle/_ref/fermi (the reference compiled in place)"}
    t = time.time()
    reads = make_reads()
    out["total_bases"] = int(sum(len(r) for r in reads))
    write_fastq(d + "/rep.fq", reads)
    out["fastq"] = md5_of(["cat", d + "/rep.fq"])
    print("reads written", time.time() - t, out["fastq"], flush=True)
    subprocess.check_call([REF, "build", "-fo", d + "/rep.fmd", d + "/rep.fq"], stderr=subprocess.DEVNULL)
    out["fmd"] = md5_of(["cat", d + "/rep.fmd"])
    print("build done", time.time() - t, out["fmd"], flush=True)
    out["unitig_l50_t1"] = md5_of([REF, "unitig", "-l50", "-t1", d + "/rep.fmd"])
    print("unitig done", time.time() - t, out["unitig_l50_t1"], flush=True)
    out["correct_t1"] = md5_of([REF, "correct", "-t1", d + "/rep.fmd", d + "/rep.fq"])
    print("correct done", time.time() - t, out["correct_t1"], flush=True)
    json.dump(out, open(os.path.join(HERE, "md5_repeat.json"), "w"), indent=1)
