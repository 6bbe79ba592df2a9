import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fermi_amd import api
n = 1 << 27
line = int(os.environ.get("PROBE_LINE", "128"))
ms = api.probe_gather(64 << 30, line, n, iters=1)
print("probe: %d lines of %d B from 64 GiB: %.3f ms, %.1f GB/s" % (n, line, ms, n * line / ms / 1e6))
