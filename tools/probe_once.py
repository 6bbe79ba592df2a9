import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fermi_amd import api
n = 1 << 27
ms = api.probe_gather(64 << 30, 128, n, iters=1)
print("probe: %d lines of 128 B from 64 GiB: %.3f ms, %.1f GB/s" % (n, ms, n * 128 / ms / 1e6))
