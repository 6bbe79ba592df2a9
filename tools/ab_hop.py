#!/usr/bin/env python3
"""`fermi-amd unitig -l50` on N 100-bp reads, error-free and with 1 % substitutions, each three ways: as shipped; FMD_WALK_NO_HOP=1 (every step of
the walk through the record, the link, the packed row: round 3's step); FMD_HOST_RELINK=1 (the whole table linked again by host threads after the
overflow pass instead of the rows that changed).  Phase times (FMD_TIMING) and the MAG's md5 each way.
Usage: python tools/ab_hop.py [n_reads=10000000]"""
import hashlib, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from fermi_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
AMD = os.path.join(ROOT, "fermi_amd", "bin", "fermi-amd")
D = "/tmp/fmd_ab_hop"; os.makedirs(D, exist_ok=True)
lut = np.frombuffer(b"$ACGTN", dtype=np.uint8)
env = dict(os.environ, FMD_TIMING="1")
ERRS = [float(x) for x in os.environ.get("AB_ERRS", "0,0.01").split(",")]
MODES = [("as shipped", {}), ("FMD_WALK_NO_HOP=1", {"FMD_WALK_NO_HOP": "1"}), ("FMD_HOST_RELINK=1", {"FMD_HOST_RELINK": "1"})]
if "AB_MODES" in os.environ:   # e.g. AB_MODES="FMD_WALK_NO_DEFER=1;FMD_WALK_NO_HOP=1" (as shipped always runs first)
    MODES = [("as shipped", {})] + [(m, dict(kv.split("=", 1) for kv in m.split(","))) for m in os.environ["AB_MODES"].split(";") if m]
for err in ERRS:
    with open(D + "/r.fq", "wb") as fp:
        for s in range(0, n, 1_000_000):
            c = min(1_000_000, n - s)
            r = lut[synth.reads(synth.DEFAULT_SEED, n, 100, 30, err, start=s, count=c)]
            fp.write(b"".join(b"@r%d\n%s\n+\n%s\n" % (s + i, r[i].tobytes(), b"I" * 100) for i in range(c)))
    subprocess.run([AMD, "build", "-fo", D + "/a.fmd", D + "/r.fq"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, env=env, check=True)
    print("==== %d reads, e = %g" % (n, err), flush=True)
    for rep in range(2):
        for name, extra in MODES:
            if name.startswith("FMD_HOST") and err == 0.0:
                continue
            t = time.time()
            p = subprocess.run([AMD, "unitig", "-l50", D + "/a.fmd"], stdout=open(D + "/a.mag", "wb"), stderr=subprocess.PIPE, env=dict(env, **extra))
            dt = time.time() - t
            h = hashlib.md5()
            with open(D + "/a.mag", "rb") as f:
                for blk in iter(lambda: f.read(1 << 24), b""):
                    h.update(blk)
            print("-- %s: %.2f s, rc %d, MAG %d bytes md5 %s" % (name, dt, p.returncode, os.path.getsize(D + "/a.mag"), h.hexdigest()), flush=True)
            print("\n".join("   " + l[:260] for l in p.stderr.decode().splitlines() if "M::" in l and ("table" in l or "walk" in l or "link" in l or "hop" in l or "rows again" in l or "packed_batch" in l or "main]" in l or "released" in l)), flush=True)
