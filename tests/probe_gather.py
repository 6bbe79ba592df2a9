import sys; sys.path.insert(0,'.')
from fermi_amd import api
for ws in (1<<30, 8<<30, 64<<30):
    for lb in (64,128,256):
        n = 1<<27
        ms = api.probe_gather(ws, lb, n, iters=2)
        print(f"ws={ws>>30}GiB line={lb}B  {n/ms/1e6:.2f} Glines/s  {n*lb/ms/1e6:.1f} GB/s", flush=True)
