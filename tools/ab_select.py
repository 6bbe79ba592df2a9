#!/usr/bin/env python3
"""A/B of the in-place builder's bucket selection: the positions of a bucket found sixteen per 64-bit word of the 4-bit text (default) against symbol by symbol
(FMD_BUILD_SELECT_BYTES=1), each in its own process at the depth config 5 uses (FMD_BUILD_DEPTH=4): phase times (FMD_TIMING) and an md5 of the exported BWT.
Usage: python tools/ab_select.py [n_reads=20000000]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import ctypes as C, hashlib, time
    sys.path.insert(0, ROOT)
    import numpy as np, torch
    from fermi_amd import api, synth
    n = int(sys.argv[2]); L = 100
    dev = torch.device("cuda", 0); torch.cuda.set_device(0)
    lib = api.lib()
    gen = synth.genome_torch(synth.DEFAULT_SEED, n, L, 30, dev)
    b = C.c_void_p()
    api.check(lib.fmd_builder_new(0, n, L, C.byref(b)))
    for s in range(0, n, 10_000_000):
        c = min(10_000_000, n - s)
        piece = synth.reads_torch(synth.DEFAULT_SEED, n, L, 30, 0.0, dev, start=s, count=c, gen=gen)
        api.check(lib.fmd_builder_add_dev(b, None, c, piece.data_ptr()))
        torch.cuda.synchronize(); del piece
    torch.cuda.empty_cache()
    t0 = time.time(); h = C.c_void_p()
    api.check(lib.fmd_builder_finish(b, C.byref(h)))
    ix = api.DevIndex(h); t = time.time() - t0
    md = hashlib.md5(); buf = np.empty(1 << 28, dtype=np.uint8)
    for o in range(0, ix.n, 1 << 28):
        m = min(1 << 28, ix.n - o)
        api.check(lib.fmd_dev_export_bwt(ix.h, o, m, buf.ctypes.data)); md.update(buf[:m].tobytes())
    print("built in place: %d symbols in %.2f s; BWT md5 %s" % (ix.n, t, md.hexdigest()), flush=True)
    sys.exit(0)
n = sys.argv[1] if len(sys.argv) > 1 else "20000000"
res = []
for label, env in (("symbol by symbol (FMD_BUILD_SELECT_BYTES=1)", {"FMD_BUILD_SELECT_BYTES": "1"}), ("sixteen positions per word", {})):
    p = subprocess.run([sys.executable, os.path.abspath(__file__), "child", n], env=dict(os.environ, FMD_TIMING="1", FMD_BUILD_DEPTH="4", **env), stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    out = [l for l in p.stdout.decode().splitlines() if "fmd_build]" in l or "built in place" in l]
    print(label + ":"); print("\n".join("    " + l for l in out), flush=True)
    res.append([l for l in out if "md5" in l][-1].split("md5 ")[1] if p.returncode == 0 and any("md5" in l for l in out) else "failed rc %d" % p.returncode)
print("SAME BWT" if res[0] == res[1] and "failed" not in res[0] else "DIFFERENT: %s" % res)
