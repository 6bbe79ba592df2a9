"""Step-by-step sanity run of one HIP library build (used when bringing up a new block geometry):
every step prints before the next starts, so a hang is attributable.  Run under `timeout`."""
import sys, os, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from fermi_amd import api
import orcbind
g = np.load('tests/golden/tiny_vectors.npz')
d = api.DevIndex.open('tests/golden/tiny.fmd')
o = orcbind.OrcIndex('tests/golden/tiny.fmd')
ok, sym = d.rank1a(g['rank1a_k'])
print("rank1a", np.array_equal(ok, g['rank1a_ok']), flush=True)
ids = np.arange(0, 600, dtype=np.uint64)
want = o.overlap_batch(ids, 50, 100, 4, 4, check_left=False)
print("mode", os.environ.get("FMD_OVLP_SLOW_ONLY"), flush=True)
rec, nei, seq = d.overlap(ids, 50, 100, 4, check_left=False)
print("overlap rec", rec.tobytes() == want[0].tobytes(), "nei", nei.tobytes() == want[1].tobytes(), "seq", seq.tobytes() == want[2].tobytes(), flush=True)
if rec.tobytes() != want[0].tobytes():
    for f in rec.dtype.names:
        bad = np.where(rec[f] != want[0][f])[0] if rec[f].ndim == 1 else np.where((rec[f] != want[0][f]).any(axis=1))[0]
        if len(bad): print(f, len(bad), bad[:5], rec[f][bad[:3]], want[0][f][bad[:3]], flush=True)
