#!/usr/bin/env python3
"""`fermi-amd unitig -l50` on N 100-bp reads WITH sequencing errors (many short unitigs, forks, edges the lfork field does not
decide): phase times (FMD_TIMING) and how many edges go through the exact check_left kernel.
Usage: python tools/time_unitig_err.py [n_reads=10000000] [err=0.01]"""
import hashlib, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from fermi_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
err = float(sys.argv[2]) if len(sys.argv) > 2 else 0.01
AMD = os.path.join(ROOT, "fermi_amd", "bin", "fermi-amd")
D = "/tmp/fmd_time_unitig_err"; os.makedirs(D, exist_ok=True)
lut = np.frombuffer(b"$ACGTN", dtype=np.uint8)
with open(D + "/r.fq", "wb") as fp:
    for s in range(0, n, 1_000_000):
        c = min(1_000_000, n - s)
        r = lut[synth.reads(synth.DEFAULT_SEED, n, 100, 30, err, start=s, count=c)]
        fp.write(b"".join(b"@r%d\n%s\n+\n%s\n" % (s + i, r[i].tobytes(), b"I" * 100) for i in range(c)))
env = dict(os.environ, FMD_TIMING="1", FMD_OVLP_STATS="1")
for cmd, out in (([AMD, "build", "-fo", D + "/a.fmd", D + "/r.fq"], None), ([AMD, "unitig", "-l50", D + "/a.fmd"], D + "/a.mag")):
    t = time.time()
    p = subprocess.run(cmd, stdout=open(out, "wb") if out else subprocess.DEVNULL, stderr=subprocess.PIPE, env=env)
    print(" ".join(cmd[1:3]), "%.1f s" % (time.time() - t), "rc", p.returncode)
    lines = [l for l in p.stderr.decode().splitlines() if "M::" in l]
    print("\n".join(lines[:3] + ["..."] + lines[-8:] if len(lines) > 12 else lines))
print("MAG bytes", os.path.getsize(D + "/a.mag"), "md5", hashlib.md5(open(D + "/a.mag", "rb").read()).hexdigest())
ref = os.path.join(ROOT, "oracle", "_ref", "fermi")
if os.path.exists(ref) and n <= 2_000_000:
    t = time.time()
    subprocess.run([ref, "unitig", "-l50", "-t1", D + "/a.fmd"], stdout=open(D + "/ref.mag", "wb"), stderr=subprocess.DEVNULL)
    print("fermi unitig -t1: %.1f s, md5 %s" % (time.time() - t, hashlib.md5(open(D + "/ref.mag", "rb").read()).hexdigest()))
