#!/usr/bin/env python3
"""`fermi-amd build` + `fermi-amd unitig -l50` on N error-free 100-bp reads with phase times (FMD_TIMING).
Usage: python tools/time_unitig_10m.py [n_reads=10000000]"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from fermi_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
AMD = os.path.join(ROOT, "fermi_amd", "bin", "fermi-amd")
D = "/tmp/fmd_time_unitig"; os.makedirs(D, exist_ok=True)
lut = np.frombuffer(b"$ACGTN", dtype=np.uint8)
with open(D + "/r.fq", "wb") as fp:
    for s in range(0, n, 1_000_000):
        c = min(1_000_000, n - s)
        r = lut[synth.reads(synth.DEFAULT_SEED, n, 100, 30, 0.0, start=s, count=c)]
        fp.write(b"".join(b"@r%d\n%s\n+\n%s\n" % (s + i, r[i].tobytes(), b"I" * 100) for i in range(c)))
env = dict(os.environ, FMD_TIMING="1")
for cmd, out in (([AMD, "build", "-fo", D + "/a.fmd", D + "/r.fq"], None), ([AMD, "unitig", "-l50", D + "/a.fmd"], D + "/a.mag")):
    t = time.time()
    p = subprocess.run(cmd, stdout=open(out, "wb") if out else subprocess.DEVNULL, stderr=subprocess.PIPE, env=env)
    print(" ".join(cmd[1:3]), "%.1f s" % (time.time() - t), "rc", p.returncode)
    print("\n".join(l for l in p.stderr.decode().splitlines() if "M::" in l))
import hashlib
print("MAG bytes", os.path.getsize(D + "/a.mag"), "md5", hashlib.md5(open(D + "/a.mag", "rb").read()).hexdigest())
# A/B: the walk as a plain pointer chase (no skip-list prefetch), and on reads with 1 % errors (many short unitigs)
t = time.time()
p = subprocess.run([AMD, "unitig", "-l50", D + "/a.fmd"], stdout=open(D + "/b.mag", "wb"), stderr=subprocess.PIPE, env=dict(env, FMD_WALK_NO_JUMP="1"))
print("unitig -l50 with FMD_WALK_NO_JUMP=1: %.1f s" % (time.time() - t), "md5", hashlib.md5(open(D + "/b.mag", "rb").read()).hexdigest())
print("\n".join(l for l in p.stderr.decode().splitlines() if "walk" in l))
# the streamed form of the table: every large block of it in file pages under FMD_TABLE_DIR (fmd_table_alloc), here a directory of /tmp
os.makedirs(D + "/pages", exist_ok=True)
t = time.time()
p = subprocess.run([AMD, "unitig", "-l50", D + "/a.fmd"], stdout=open(D + "/c.mag", "wb"), stderr=subprocess.PIPE, env=dict(env, FMD_TABLE_DIR=D + "/pages"))
print("unitig -l50 with FMD_TABLE_DIR=%s/pages (%s): %.1f s" % (D, subprocess.run(["df", "--output=fstype,avail", "-h", D + "/pages"], capture_output=True, text=True).stdout.split("\n")[1].strip(), time.time() - t),
      "md5", hashlib.md5(open(D + "/c.mag", "rb").read()).hexdigest())
print("\n".join(l for l in p.stderr.decode().splitlines() if "M::" in l))
# the host-linked form (several GPUs: here two replicas on GPU 0, and one GPU with FMD_HOST_LINK=1) and the skip list without its temporary array
for label, cmd, e in (("-g 0,0 (two replicas, host threads link the rows)", [AMD, "unitig", "-l50", "-g", "0,0", D + "/a.fmd"], env),
                      ("FMD_HOST_LINK=1", [AMD, "unitig", "-l50", D + "/a.fmd"], dict(env, FMD_HOST_LINK="1")),
                      ("FMD_FAR_CHASE=1", [AMD, "unitig", "-l50", D + "/a.fmd"], dict(env, FMD_FAR_CHASE="1"))):
    t = time.time()
    p = subprocess.run(cmd, stdout=open(D + "/d.mag", "wb"), stderr=subprocess.PIPE, env=e)
    print("unitig -l50 %s: %.1f s rc %d" % (label, time.time() - t, p.returncode), "md5", hashlib.md5(open(D + "/d.mag", "rb").read()).hexdigest())
    print("\n".join(l for l in p.stderr.decode().splitlines() if "link pass" in l or "skip list" in l or "table of" in l or "peak resident" in l))
