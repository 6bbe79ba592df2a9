"""Plumbing shared by bench.py / smoke / tests: put a read set in HBM, build its FMD index on
the GPU, optionally write the fermi-compatible .fmd.  torch is used only for device memory."""
import ctypes as C

import numpy as np

from . import api, hostlib, synth


def synth_reads_host(n_reads, read_len=100, coverage=30, err=0.0, seed=synth.DEFAULT_SEED, chunk=1_000_000):
    out = np.empty((n_reads, read_len), dtype=np.uint8)
    for s in range(0, n_reads, chunk):
        c = min(chunk, n_reads - s)
        out[s:s + c] = synth.reads(seed, n_reads, read_len, coverage, err, start=s, count=c)
    return out


class ReadsOnDevice:
    """Fixed-length nt6 reads resident in HBM: flat uint8 buffer (padded) + uint64 offsets."""

    def __init__(self, reads_host, device):
        import torch
        n, L = reads_host.shape
        self.n, self.L = n, L
        flat = torch.from_numpy(reads_host.reshape(-1))
        self.flat = torch.zeros(n * L + 64, dtype=torch.uint8, device=device)
        self.flat[: n * L].copy_(flat, non_blocking=False)
        self.off = (torch.arange(n + 1, dtype=torch.int64, device=device) * L)
        self.total = n * L

    @classmethod
    def synth(cls, n_reads, read_len, coverage, err, device, seed=synth.DEFAULT_SEED):
        """The synthetic read set of SURVEY.md 8(d), generated in HBM (synth.reads_torch == synth.reads)."""
        import torch
        self = cls.__new__(cls)
        self.n, self.L = n_reads, read_len
        self.flat = torch.zeros(n_reads * read_len + 64, dtype=torch.uint8, device=device)
        self.flat[: n_reads * read_len].view(n_reads, read_len).copy_(synth.reads_torch(seed, n_reads, read_len, coverage, err, device))
        self.off = (torch.arange(n_reads + 1, dtype=torch.int64, device=device) * read_len)
        self.total = n_reads * read_len
        return self


def build_bwt_on_device(rd, device_index=0, stream=None):
    """-> (device pointer of the BWT, n_sym); release with api.lib().fmd_dev_free."""
    d_bwt, n_sym = C.c_void_p(), C.c_uint64()
    api.check(api.lib().fmd_build_bwt_dev(device_index, stream, rd.n, rd.flat.data_ptr(), rd.off.data_ptr(),
                                          rd.total, rd.L, 1, C.byref(d_bwt), C.byref(n_sym)))
    return d_bwt, n_sym.value


def write_fmd_from_device_bwt(d_bwt, n_sym, path, device_index=0):
    """device BWT -> RLE\\6 stream on the GPU -> RLD\\2 .fmd written by the host C encoder."""
    p, nb = C.c_void_p(), C.c_uint64()
    api.check(api.lib().fmd_bwt_to_rle6(device_index, d_bwt, n_sym, C.byref(p), C.byref(nb)))
    try:
        hostlib.write_rld_from_rle6_ptr(p, nb.value, path)
    finally:
        api.lib().fmd_host_free(p)
    return nb.value
