#!/usr/bin/env python3
"""`fermi-amd build` on N 100-bp reads (error-free, then 1 % substitutions): as shipped (a byte per symbol to the host) against FMD_BUILD_RUNS=1 (the BWT leaves
the device as runs); phase times and the md5 of the .fmd each way.  Usage: python tools/ab_build.py [n_reads=10000000]"""
import hashlib, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from fermi_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
AMD = os.path.join(ROOT, "fermi_amd", "bin", "fermi-amd")
D = "/tmp/fmd_ab_build"; os.makedirs(D, exist_ok=True)
lut = np.frombuffer(b"$ACGTN", dtype=np.uint8)
env = dict(os.environ, FMD_TIMING="1")
for err in (0.0, 0.01):
    with open(D + "/r.fq", "wb") as fp:
        for s in range(0, n, 1_000_000):
            c = min(1_000_000, n - s)
            r = lut[synth.reads(synth.DEFAULT_SEED, n, 100, 30, err, start=s, count=c)]
            fp.write(b"".join(b"@r%d\n%s\n+\n%s\n" % (s + i, r[i].tobytes(), b"I" * 100) for i in range(c)))
    print("==== %d reads, e = %g" % (n, err), flush=True)
    for rep in range(2):
        for name, extra in (("as shipped", {}), ("FMD_BUILD_RUNS=1", {"FMD_BUILD_RUNS": "1"})):
            t = time.time()
            p = subprocess.run([AMD, "build", "-fo", D + "/a.fmd", D + "/r.fq"], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=dict(env, **extra))
            dt = time.time() - t
            h = hashlib.md5(open(D + "/a.fmd", "rb").read()).hexdigest()
            print("-- %s: %.2f s, rc %d, .fmd %d bytes md5 %s" % (name, dt, p.returncode, os.path.getsize(D + "/a.fmd"), h), flush=True)
            print("\n".join("   " + l[:260] for l in p.stderr.decode().splitlines() if "M::" in l), flush=True)
