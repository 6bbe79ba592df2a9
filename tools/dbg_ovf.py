import sys, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests/golden")
from fermi_amd import api, synth
import make_md5_repeat as m
m.N = 100000
reads = m.make_reads()
d = api.DevIndex.from_bwt(api.build_bwt(reads))
n = int(d.mcnt[1])
ids = np.arange(n, dtype=np.uint64)
r0, _, _ = d.overlap(ids, 50, 128, 4, check_left=False)
f = np.nonzero(r0["flags"] & 2)[0].astype(np.uint64)
print("flagged at 128/4:", len(f), "of", n, "; longer than 128:", int((r0["len"][f.astype(np.int64)] > 128).sum()))
for ml, mn in ((256, 16), (512, 32), (160, 16)):
    r1, n1, _ = d.overlap(f, 50, ml, mn, check_left=False)
    g = (r1["flags"] & 2) != 0
    print(ml, mn, "still flagged:", int(g.sum()), "len>128 among them", int((r1["len"][g] > 128).sum()), "n_nei max", int(r1["n_nei"].max()), "n_ovlp max", int(r1["n_ovlp"].max()))
    if g.any():
        print(r1[g][:3])
    r2, _, _ = d.overlap_sorted(f, 50, ml, mn, 0)
    print("   sorted job: still flagged", int(((r2["flags"] & 2) != 0).sum()))
