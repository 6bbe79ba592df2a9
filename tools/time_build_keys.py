#!/usr/bin/env python3
"""GPU BWT construction (fmd_build_bwt_dev on device-resident reads, as bench.py's setup runs it) timed on N synthetic 100-bp reads, error-free and with
1 % substitutions, with two checksums of the BWT each time.  FMD_BUILD_KEY_BYTES=1 (the A/B switch): the key kernels load byte by byte as through round 3.
Usage: [FMD_TIMING=1] [BUILD_ERRS=0,0.01] [BUILD_MODES=1,0,1,0] python tools/time_build_keys.py [n_reads=50000000]"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from fermi_amd import api, workload
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
lib = api.lib()
hip = C.CDLL("libamdhip64.so")
for err in [float(x) for x in os.environ.get("BUILD_ERRS", "0,0.01").split(",")]:
    rd = workload.ReadsOnDevice.synth(n, 100, 30, err, dev)
    for mode in os.environ.get("BUILD_MODES", "1,0,1,0").split(","):
        os.environ["FMD_BUILD_KEY_BYTES"] = mode
        torch.cuda.synchronize(); t = time.time()
        d_bwt, n_sym = workload.build_bwt_on_device(rd, 0)
        torch.cuda.synchronize(); dt = time.time() - t
        s = x = 0
        for o in range(0, n_sym, 1 << 30):   # checksums in pieces of 2^30 symbols
            m = min(1 << 30, n_sym - o)
            buf = torch.empty(m, dtype=torch.uint8, device=dev)
            assert hip.hipMemcpy(C.c_void_p(buf.data_ptr()), C.c_void_p(d_bwt.value + o), C.c_size_t(m), 3) == 0
            b = buf.to(torch.int64)
            s += int(b.sum().item()); x += int((b * ((torch.arange(m, device=dev) + o) % 1000003)).sum().item())
            del buf, b
        lib.fmd_dev_free(d_bwt)
        print("e = %g, FMD_BUILD_KEY_BYTES=%s: %.2f s for %d symbols (%.2f G symbols/s), checksums %d %d" % (err, mode, dt, n_sym, n_sym / dt / 1e9, s, x), flush=True)
    del rd
    torch.cuda.empty_cache()
