#!/usr/bin/env python3
"""What a SMALL overlap call costs (the tests make thousands): fmd_ovlp_batch of 4000 ids on the tiny fixture, ms per call, under the switches of round 5."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from fermi_amd import api
d = api.DevIndex.open(os.path.join(ROOT, "tests", "golden", "tiny.fmd"))
ids = np.arange(4000, dtype=np.uint64)
for env in ({}, {"FMD_NEI_LANE": "0"}, {"FMD_GRP_DOWN": "0"}, {"FMD_NEI_LANE": "0", "FMD_GRP_DOWN": "0"}):
    for k, v in env.items():
        os.environ[k] = v
    d.overlap(ids, 50, 100, 8)
    t = time.time()
    for _ in range(20):
        d.overlap(ids, 50, 100, 8)
    print(env or "default", "%.1f ms per call" % ((time.time() - t) / 20 * 1e3), flush=True)
    for k in env:
        del os.environ[k]
