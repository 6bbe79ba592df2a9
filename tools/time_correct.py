#!/usr/bin/env python3
"""`fermi-amd build` + `fermi-amd correct` on N synthetic 100-bp reads with substitution errors, stage times (FMD_TIMING),
and -- when oracle/_ref/fermi travelled and the read set is small enough -- the md5 of the reference binary's output.
Usage: python tools/time_correct.py [n_reads=10000000] [err=0.01] [compare_up_to=2000000]"""
import hashlib, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from fermi_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
err = float(sys.argv[2]) if len(sys.argv) > 2 else 0.01
cmp_max = int(sys.argv[3]) if len(sys.argv) > 3 else 2_000_000
AMD = os.path.join(ROOT, "fermi_amd", "bin", "fermi-amd")
REF = os.path.join(ROOT, "oracle", "_ref", "fermi")
D = "/tmp/fmd_time_correct"; os.makedirs(D, exist_ok=True)
lut = np.frombuffer(b"$ACGTN", dtype=np.uint8)
rng = np.random.default_rng(5)
with open(D + "/r.fq", "wb") as fp:
    for s in range(0, n, 1_000_000):
        c = min(1_000_000, n - s)
        r = lut[synth.reads(synth.DEFAULT_SEED, n, 100, 30, err, start=s, count=c)]
        q = rng.integers(33 + 5, 33 + 41, size=(c, 100)).astype(np.uint8)     # qualities matter to ec_fix: not all 'I'
        fp.write(b"".join(b"@r%d\n%s\n+\n%s\n" % (s + i, r[i].tobytes(), q[i].tobytes()) for i in range(c)))
env = dict(os.environ, FMD_TIMING="1")
def run(cmd, out):
    t = time.time()
    p = subprocess.run(cmd, stdout=open(out, "wb") if out else subprocess.DEVNULL, stderr=subprocess.PIPE, env=env)
    print(" ".join(os.path.basename(c) for c in cmd[:3]), "%.1f s" % (time.time() - t), "rc", p.returncode)
    print("\n".join(l for l in p.stderr.decode().splitlines() if "M::" in l))
    return time.time() - t
def md5(path):
    h = hashlib.md5()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 24), b""):
            h.update(blk)
    return h.hexdigest()
run([AMD, "build", "-fo", D + "/a.fmd", D + "/r.fq"], None)
t_amd = run([AMD, "correct", "-t16", D + "/a.fmd", D + "/r.fq"], D + "/a.ec.fq")
print("fermi-amd correct: %d reads, %.0f reads/s, output md5 %s, %d bytes" % (n, n / t_amd, md5(D + "/a.ec.fq"), os.path.getsize(D + "/a.ec.fq")))
if os.path.exists(REF) and n <= cmp_max:
    t = time.time()
    subprocess.run([REF, "correct", "-t16", D + "/a.fmd", D + "/r.fq"], stdout=open(D + "/ref.ec.fq", "wb"), stderr=subprocess.DEVNULL)
    print("fermi correct -t16 (reference binary): %.1f s, output md5 %s -> %s" % (time.time() - t, md5(D + "/ref.ec.fq"),
          "IDENTICAL" if md5(D + "/ref.ec.fq") == md5(D + "/a.ec.fq") else "DIFFERENT"))
