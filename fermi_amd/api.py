"""ctypes view of the C ABI in include/fmd_hip.h (libfmdhip.so) -- the product path.

Names mirror the reference's C API (rld.h:45-58, fermi.h:61-103): `rld_restore` -> DevIndex.open,
`rld_rank1a` -> rank1a, `rld_rank2a` -> rank2a, `fm6_extend` -> extend, `fm_backward_search` ->
backward_search, `fm_retrieve` -> retrieve.  Everything here runs on the GPU; there is no CPU
fallback -- if the library or a device is missing the calls raise FmdError.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FMD_HIP_LIB") or os.path.join(_HERE, "lib", "libfmdhip.so")  # FMD_HIP_LIB: A/B builds
INTV_DT = np.dtype([("x", "<u8", 3), ("info", "<u8")])  # fmd_intv_t == fmintv_t (fermi.h:13-16)
NONE64 = np.uint64(0xFFFFFFFFFFFFFFFF)
OVLP_DT = np.dtype([("rank", "<u8"), ("k", "<u8", 3), ("len", "<i4"), ("status", "<i4"), ("n_ovlp", "<i4"),
                    ("rbeg", "<i4"), ("ext_len", "<i4"), ("n_nei", "<i4"), ("flags", "<u4"), ("reserved", "<u2"), ("lfork", "<u2")])
OVLP_F_FORKED, OVLP_F_OVERFLOW, OVLP_F_FIXED = 1, 2, 4

# every symbol include/fmd_hip.h declares (tests check the library exports all of them)
ABI_SYMBOLS = [
    "fmd_strerror", "fmd_last_hip_error", "fmd_device_count",
    "fmd_dev_open_file", "fmd_dev_open_rld", "fmd_dev_open_rle6", "fmd_dev_open_bwt", "fmd_dev_open_bwt_dev",
    "fmd_dev_close", "fmd_dev_trim", "fmd_dev_info", "fmd_dev_sync", "fmd_dev_line_count",
    "fmd_rank1a_dev", "fmd_rank2a_dev", "fmd_rank1a_batch", "fmd_rank2a_batch",
    "fmd_extend_dev", "fmd_extend_batch", "fmd_bsearch_dev", "fmd_bsearch_batch",
    "fmd_retrieve_dev", "fmd_retrieve_batch", "fmd_probe_gather",
    "fmd_build_bwt", "fmd_build_bwt_dev", "fmd_dev_free", "fmd_builder_new", "fmd_builder_add_dev", "fmd_builder_finish", "fmd_builder_free", "fmd_bwt_to_rle6", "fmd_host_free",
    "fmd_dev_malloc", "fmd_memcpy_h2d", "fmd_memcpy_d2h",
    "fmd_smem_work_bytes", "fmd_smem_dev", "fmd_smem_batch", "fmd_smem_win_dev", "fmd_smem_win_batch", "fmd_reach_dev", "fmd_reach_batch", "fmd_dev_export_bwt", "fmd_dev_check_rank", "fmd_dev_build_pairs", "fmd_dev_check_pairs", "fmd_dev_line_count3",
    "fmd_kmer_work_bytes", "fmd_kmer_collect_dev", "fmd_kmer_collect_part_dev", "fmd_kmer_collect", "fmd_kmer_collect_seeds",
    "fmd_ectab_build_dev", "fmd_ectab_build", "fmd_ectab_free", "fmd_ecfix_work_bytes", "fmd_ecfix_dev", "fmd_ecfix_batch", "fmd_ectab_line_count",
    "fmd_ovlp_work_bytes", "fmd_ovlp_dev", "fmd_ovlp_sorted_work_bytes", "fmd_ovlp_sorted_dev", "fmd_ovlp_batch", "fmd_ovlp_check_left_dev", "fmd_seqinfo_dev", "fmd_seqinfo_batch",
    "fmd_ovlp_pack_max_bytes", "fmd_ovlp_pack_work_bytes", "fmd_ovlp_pack_dev", "fmd_ovlp_packed_batch", "fmd_ovlp_packed_free", "fmd_table_alloc", "fmd_table_free", "fmd_ovlp_link_dev", "fmd_ovlp_packed_table",
    "fmd_ovlp_packed_stream", "fmd_ovlp_tabjob_rows", "fmd_ovlp_tabjob_patch", "fmd_ovlp_tabjob_link", "fmd_ovlp_tabjob_free",
    "fmd_ovlp_two_pass_ok", "fmd_ovlp_head_work_bytes", "fmd_ovlp_head_dev", "fmd_ovlp_tail_dev", "fmd_ovlp_pack_rows_dev",
    "fmd_comm_rccl_unique_id", "fmd_comm_rccl_init", "fmd_comm_rccl_version", "fmd_comm_rccl_count", "fmd_comm_free",
    "fmd_ovlp_side_work_bytes", "fmd_ovlp_rerun_overflow_dev",
    "fmd_ovlp_dist_new", "fmd_ovlp_dist_step", "fmd_ovlp_dist_table", "fmd_ovlp_dist_local", "fmd_ovlp_dist_free",
]


class FmdError(RuntimeError):
    pass


# status codes of include/fmd_hip.h
FMD_OK, FMD_E_NODEV, FMD_E_ARG, FMD_E_FORMAT, FMD_E_IO, FMD_E_NOMEM, FMD_E_HIP, FMD_E_OVERFLOW = 0, -1, -2, -3, -4, -5, -6, -7


class Info(C.Structure):
    _fields_ = [("cnt", C.c_uint64 * 7), ("mcnt", C.c_uint64 * 7), ("n_blocks", C.c_uint64),
                ("hbm_bytes", C.c_uint64), ("device", C.c_int)]


_lib = None
_count_lib = None
COUNT_LIB_PATH = os.path.join(_HERE, "lib", "libfmdhip_count.so")  # same sources, gathers instrumented (-DFMD_COUNT_LINES=1)


def _configure(L):
    """Prototypes of include/fmd_hip.h on a loaded library."""
    vp, sz, u64p = C.c_void_p, C.c_size_t, C.c_void_p
    L.fmd_strerror.restype = C.c_char_p; L.fmd_strerror.argtypes = [C.c_int]
    L.fmd_last_hip_error.restype = C.c_char_p
    L.fmd_device_count.restype = C.c_int
    L.fmd_dev_open_file.argtypes = [C.c_int, C.c_char_p, C.POINTER(vp)]
    L.fmd_dev_open_rld.argtypes = [C.c_int, vp, C.c_uint64, vp, C.POINTER(vp)]
    L.fmd_dev_open_rle6.argtypes = [C.c_int, vp, C.c_uint64, C.POINTER(vp)]
    L.fmd_dev_open_bwt.argtypes = [C.c_int, vp, C.c_uint64, C.POINTER(vp)]
    L.fmd_dev_open_bwt_dev.argtypes = [C.c_int, vp, C.c_uint64, C.POINTER(vp)]
    L.fmd_dev_close.restype = None; L.fmd_dev_close.argtypes = [vp]
    L.fmd_dev_trim.restype = C.c_uint64; L.fmd_dev_trim.argtypes = [vp]
    L.fmd_dev_info.argtypes = [vp, C.POINTER(Info)]
    L.fmd_dev_sync.argtypes = [vp, vp]
    L.fmd_rank1a_dev.argtypes = [vp, vp, sz, u64p, u64p, vp]
    L.fmd_rank2a_dev.argtypes = [vp, vp, sz, u64p, u64p, u64p, u64p]
    L.fmd_rank1a_batch.argtypes = [vp, sz, u64p, u64p, vp]
    L.fmd_rank2a_batch.argtypes = [vp, sz, u64p, u64p, u64p, u64p]
    L.fmd_extend_dev.argtypes = [vp, vp, sz, vp, vp, vp]
    L.fmd_extend_batch.argtypes = [vp, sz, vp, vp, vp]
    L.fmd_bsearch_dev.argtypes = [vp, vp, sz, vp, u64p, u64p, u64p, u64p]
    L.fmd_bsearch_batch.argtypes = [vp, sz, vp, u64p, u64p, u64p, u64p]
    L.fmd_retrieve_dev.argtypes = [vp, vp, sz, u64p, vp, C.c_uint32, vp, u64p]
    L.fmd_retrieve_batch.argtypes = [vp, sz, u64p, vp, C.c_uint32, vp, u64p]
    L.fmd_build_bwt.argtypes = [C.c_int, sz, vp, u64p, vp, C.POINTER(C.c_uint64)]
    L.fmd_build_bwt_dev.argtypes = [C.c_int, vp, sz, vp, u64p, C.c_uint64, C.c_uint32, C.c_int, C.POINTER(vp), C.POINTER(C.c_uint64)]
    L.fmd_dev_free.restype = None; L.fmd_dev_free.argtypes = [vp]
    L.fmd_builder_new.argtypes = [C.c_int, C.c_uint64, C.c_uint32, C.POINTER(vp)]
    L.fmd_builder_add_dev.argtypes = [vp, vp, C.c_uint64, vp]
    L.fmd_builder_finish.argtypes = [vp, C.POINTER(vp)]
    L.fmd_builder_free.restype = None; L.fmd_builder_free.argtypes = [vp]
    L.fmd_bwt_to_rle6.argtypes = [C.c_int, vp, C.c_uint64, C.POINTER(vp), C.POINTER(C.c_uint64)]
    L.fmd_host_free.restype = None; L.fmd_host_free.argtypes = [vp]
    L.fmd_dev_malloc.argtypes = [C.c_int, sz, C.POINTER(vp)]
    L.fmd_memcpy_h2d.argtypes = [vp, vp, sz, vp]
    L.fmd_memcpy_d2h.argtypes = [vp, vp, sz, vp]
    L.fmd_ovlp_work_bytes.restype = sz; L.fmd_ovlp_work_bytes.argtypes = [sz, C.c_uint32, C.c_int]
    L.fmd_ovlp_sorted_work_bytes.restype = sz; L.fmd_ovlp_sorted_work_bytes.argtypes = [sz, sz, C.c_uint32, C.c_int]
    L.fmd_ovlp_sorted_dev.argtypes = [vp, vp, sz, u64p, C.c_int, C.c_uint32, C.c_uint32, vp, vp, vp, C.c_uint32, vp, sz, sz]
    L.fmd_ovlp_dev.argtypes = [vp, vp, sz, u64p, C.c_int, C.c_uint32, C.c_uint32, vp, vp, vp, C.c_uint32, vp, sz]
    L.fmd_ovlp_batch.argtypes = [vp, sz, u64p, C.c_int, C.c_uint32, C.c_uint32, vp, vp, vp, C.c_uint32, C.c_int]
    L.fmd_ovlp_check_left_dev.argtypes = [vp, vp, sz, C.c_int, C.c_uint32, vp, vp, C.c_uint32, vp, sz]
    L.fmd_kmer_work_bytes.restype = sz; L.fmd_kmer_work_bytes.argtypes = [C.c_uint64]
    L.fmd_kmer_collect_dev.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, vp, sz, C.c_uint64, vp, vp, vp, vp]
    L.fmd_kmer_collect_part_dev.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, sz, C.c_uint64, vp, vp, vp, vp]
    L.fmd_kmer_collect.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(C.c_uint64), vp]
    L.fmd_kmer_collect_seeds.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(C.c_uint64), vp]
    L.fmd_smem_work_bytes.restype = sz; L.fmd_smem_work_bytes.argtypes = [sz, C.c_uint32]
    L.fmd_smem_dev.argtypes = [vp, vp, sz, vp, u64p, C.c_int, C.c_uint32, C.c_uint32, vp, vp, vp, sz]
    L.fmd_smem_batch.argtypes = [vp, sz, vp, u64p, C.c_int, C.c_uint32, C.c_uint32, vp, vp]
    L.fmd_dev_export_bwt.argtypes = [vp, C.c_uint64, C.c_uint64, vp]
    L.fmd_dev_check_rank.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.fmd_dev_build_pairs.argtypes = [vp, C.POINTER(C.c_int)]
    L.fmd_dev_check_pairs.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.fmd_reach_dev.argtypes = [vp, vp, sz, vp, vp]
    L.fmd_reach_batch.argtypes = [vp, sz, vp, vp]
    L.fmd_smem_win_dev.argtypes = [vp, vp, sz, vp, vp, C.c_int, C.c_uint32, C.c_uint32, vp, vp, vp, sz]
    L.fmd_smem_win_batch.argtypes = [vp, sz, vp, C.c_uint64, vp, C.c_int, C.c_uint32, C.c_uint32, vp, vp]
    L.fmd_seqinfo_dev.argtypes = [vp, vp, sz, u64p, C.c_uint32, vp, vp, C.c_uint32, vp, sz]
    L.fmd_seqinfo_batch.argtypes = [vp, sz, u64p, C.c_uint32, vp]
    L.fmd_probe_gather.argtypes = [C.c_int, C.c_uint64, C.c_uint32, C.c_uint64, C.c_int, C.POINTER(C.c_float)]
    L.fmd_dev_line_count.argtypes = [vp, vp, C.c_int, C.POINTER(C.c_int)]
    L.fmd_dev_line_count3.argtypes = [vp, vp, C.c_int, C.POINTER(C.c_int)]
    L.fmd_ovlp_pack_max_bytes.restype = sz; L.fmd_ovlp_pack_max_bytes.argtypes = [sz, C.c_uint32, C.c_uint32]
    L.fmd_ovlp_pack_work_bytes.restype = sz; L.fmd_ovlp_pack_work_bytes.argtypes = [sz]
    L.fmd_ectab_build_dev.argtypes = [C.c_int, vp, C.c_int, C.c_int, C.c_uint64, vp, vp, vp, C.POINTER(vp)]
    L.fmd_ectab_build.argtypes = [C.c_int, C.c_int, C.c_int, C.c_uint64, vp, vp, vp, C.POINTER(vp)]
    L.fmd_ectab_free.restype = None; L.fmd_ectab_free.argtypes = [vp]
    L.fmd_ecfix_work_bytes.restype = sz; L.fmd_ecfix_work_bytes.argtypes = [vp, sz, C.c_uint32]
    L.fmd_ecfix_dev.argtypes = [vp, vp, sz, vp, vp, u64p, C.c_int, C.c_uint32, vp, vp, sz]
    L.fmd_ecfix_batch.argtypes = [vp, sz, vp, vp, u64p, C.c_int, vp]
    L.fmd_ectab_line_count.argtypes = [vp, vp, C.c_int, C.POINTER(C.c_int)]
    L.fmd_ovlp_packed_batch.argtypes = [vp, vp, C.c_uint64, C.c_uint64, sz, C.c_int, C.c_uint32, C.c_uint32, C.c_int, vp, vp, C.c_uint32, vp]
    L.fmd_ovlp_link_dev.argtypes = [vp, vp, sz, vp, vp, C.c_uint32, vp, vp, vp, vp]
    L.fmd_ovlp_packed_table.argtypes = [vp, sz, C.c_int, C.c_uint32, C.c_uint32, vp, vp, C.c_uint32, vp, vp, vp, C.POINTER(vp), C.POINTER(C.c_uint64)]
    L.fmd_ovlp_packed_free.restype = None; L.fmd_ovlp_packed_free.argtypes = [vp, sz]
    L.fmd_table_alloc.restype = vp; L.fmd_table_alloc.argtypes = [sz]
    L.fmd_table_free.restype = None; L.fmd_table_free.argtypes = [vp]
    L.fmd_ovlp_pack_dev.argtypes = [vp, vp, sz, vp, vp, C.c_uint32, vp, C.c_uint32, vp, vp, vp, C.c_uint64, vp, sz]
    L.fmd_ovlp_two_pass_ok.argtypes = [vp, sz, C.c_int, C.c_uint32]
    L.fmd_ovlp_head_work_bytes.restype = sz; L.fmd_ovlp_head_work_bytes.argtypes = [sz]
    L.fmd_ovlp_head_dev.argtypes = [vp, vp, sz, u64p, C.c_int, C.c_uint32, vp, vp, vp, vp, vp, sz]
    L.fmd_ovlp_tail_dev.argtypes = [vp, vp, sz, vp, vp, C.c_int, C.c_uint32, C.c_uint32, vp, vp, vp, C.c_uint32, vp, sz]
    L.fmd_ovlp_pack_rows_dev.argtypes = [vp, vp, sz, vp, vp, C.c_uint64, C.c_uint64, vp, vp, C.c_uint32, vp, C.c_uint32, vp, vp, vp, vp, C.c_uint64, vp, sz]
    L.fmd_ovlp_side_work_bytes.restype = sz; L.fmd_ovlp_side_work_bytes.argtypes = [sz, C.c_uint32, C.c_int]
    L.fmd_ovlp_rerun_overflow_dev.argtypes = [vp, vp, sz, vp, vp, C.c_int, C.c_uint32, C.c_uint32, C.c_uint64, vp, vp, vp, vp, vp, C.c_uint32, vp, sz, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.fmd_comm_rccl_unique_id.argtypes = [vp]
    L.fmd_comm_rccl_init.argtypes = [C.c_int, C.c_int, C.c_int, vp, C.POINTER(vp)]
    L.fmd_comm_rccl_version.restype = C.c_int
    L.fmd_comm_free.restype = None; L.fmd_comm_free.argtypes = [vp]
    L.fmd_ovlp_dist_new.argtypes = [vp, vp, vp, C.POINTER(vp)]
    L.fmd_ovlp_dist_step.argtypes = [vp, vp, vp]
    L.fmd_ovlp_dist_table.argtypes = [vp, vp]
    L.fmd_ovlp_dist_local.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(C.c_uint32)]
    L.fmd_ovlp_dist_free.restype = None; L.fmd_ovlp_dist_free.argtypes = [vp]
    return L


def _load(path):
    try:  # one process, one HIP runtime: if torch is going to be used, its bundled
        import torch  # noqa: F401  libamdhip64 must be the copy that gets loaded (same soname)
    except Exception:
        pass
    return _configure(C.CDLL(path))


def lib():
    """Load libfmdhip.so; raises FmdError (never falls back) when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise FmdError("libfmdhip.so is not built (%s); run `make` or __graft_entry__.build()" % LIB_PATH)
        _lib = _load(LIB_PATH)
    return _lib


def count_lib():
    """The instrumented build (measurement only: bench.py's device-byte model); None when it has not been built."""
    global _count_lib
    if _count_lib is None and os.path.exists(COUNT_LIB_PATH):
        _count_lib = _load(COUNT_LIB_PATH)
    return _count_lib


def check(rc):
    if rc != 0:
        L = lib()
        msg = L.fmd_strerror(rc).decode()
        if rc == -6:
            msg += ": " + L.fmd_last_hip_error().decode()
        raise FmdError("libfmdhip: %s (%d)" % (msg, rc))


def device_count():
    return lib().fmd_device_count()


def _ptr(a):
    return a.ctypes.data


def flatten_reads(seqs):
    """list/2-D array of nt6 reads -> (uint8 concatenation padded to 8 bytes, uint64 offsets[n+1])"""
    if isinstance(seqs, np.ndarray) and seqs.ndim == 2:
        n, ln = seqs.shape
        off = np.arange(n + 1, dtype=np.uint64) * np.uint64(ln)
        flat = np.ascontiguousarray(seqs, dtype=np.uint8).reshape(-1)
    else:
        lens = np.array([len(s) for s in seqs], dtype=np.uint64)
        off = np.zeros(len(seqs) + 1, dtype=np.uint64)
        np.cumsum(lens, out=off[1:])
        flat = np.concatenate([np.asarray(s, dtype=np.uint8) for s in seqs]) if len(seqs) else np.zeros(0, np.uint8)
    pad = (-len(flat)) % 8 + 8
    return np.concatenate([flat, np.zeros(pad, dtype=np.uint8)]), off


class DevIndex:
    """An FMD index resident in one GPU's HBM (fmd_dev_t)."""

    def __init__(self, handle):
        self.h = handle
        info = Info()
        check(lib().fmd_dev_info(self.h, C.byref(info)))
        self.cnt = np.array(info.cnt[:], dtype=np.uint64)
        self.mcnt = np.array(info.mcnt[:], dtype=np.uint64)
        self.n_blocks = int(info.n_blocks)
        self.hbm_bytes = int(info.hbm_bytes)
        self.device = int(info.device)
        self.n = int(self.mcnt[0])

    # ---- rld_restore (rld.c:288) and friends
    @classmethod
    def open(cls, fn, device=0):
        h = C.c_void_p()
        check(lib().fmd_dev_open_file(device, fn.encode(), C.byref(h)))
        return cls(h)

    @classmethod
    def from_bwt(cls, bwt, device=0):
        bwt = np.ascontiguousarray(bwt, dtype=np.uint8)
        h = C.c_void_p()
        check(lib().fmd_dev_open_bwt(device, _ptr(bwt), len(bwt), C.byref(h)))
        return cls(h)

    @classmethod
    def from_bwt_dev(cls, d_ptr, n, device=0):
        h = C.c_void_p()
        check(lib().fmd_dev_open_bwt_dev(device, d_ptr, n, C.byref(h)))
        return cls(h)

    @classmethod
    def from_rle6(cls, runs, device=0):
        runs = np.ascontiguousarray(runs, dtype=np.uint8)
        h = C.c_void_p()
        check(lib().fmd_dev_open_rle6(device, _ptr(runs), len(runs), C.byref(h)))
        return cls(h)

    def refresh_info(self):
        """cnt / mcnt / bytes again (the handle may have grown: fmd_dev_build_pairs)"""
        info = Info()
        check(lib().fmd_dev_info(self.h, C.byref(info)))
        self.hbm_bytes = int(info.hbm_bytes)

    def build_pairs(self):
        """the two-base blocks now (fmd_dev_build_pairs) -> True when the handle has them"""
        b = C.c_int(0)
        check(lib().fmd_dev_build_pairs(self.h, C.byref(b)))
        return bool(b.value)

    def check_pairs(self):
        """every row's pair step against two single LF steps -> (number of bad rows, the first one)"""
        bad, first = C.c_uint64(), C.c_uint64()
        check(lib().fmd_dev_check_pairs(self.h, C.byref(bad), C.byref(first)))
        return int(bad.value), int(first.value)

    def close(self):  # rld_destroy (rld.c:81)
        if self.h:
            lib().fmd_dev_close(self.h)
            self.h = None

    def sync(self, stream=None):
        check(lib().fmd_dev_sync(self.h, stream))

    # ---- host-array forms
    def rank1a(self, ks):
        ks = np.ascontiguousarray(ks, dtype=np.uint64)
        ok = np.zeros((len(ks), 6), dtype=np.uint64); sym = np.zeros(len(ks), dtype=np.int8)
        check(lib().fmd_rank1a_batch(self.h, len(ks), _ptr(ks), _ptr(ok), _ptr(sym)))
        return ok, sym

    def rank2a(self, ks, ls):
        ks = np.ascontiguousarray(ks, dtype=np.uint64); ls = np.ascontiguousarray(ls, dtype=np.uint64)
        ok = np.zeros((len(ks), 6), dtype=np.uint64); ol = np.zeros((len(ks), 6), dtype=np.uint64)
        check(lib().fmd_rank2a_batch(self.h, len(ks), _ptr(ks), _ptr(ls), _ptr(ok), _ptr(ol)))
        return ok, ol

    def extend(self, iks, is_back):
        iks = np.ascontiguousarray(iks, dtype=INTV_DT); is_back = np.ascontiguousarray(is_back, dtype=np.uint8)
        out = np.zeros((len(iks), 6), dtype=INTV_DT)
        check(lib().fmd_extend_batch(self.h, len(iks), _ptr(iks), _ptr(is_back), _ptr(out)))
        return out

    def backward_search(self, seqs):
        flat, off = flatten_reads(seqs)
        n = len(off) - 1
        cnt = np.zeros(n, dtype=np.uint64); beg = np.zeros(n, dtype=np.uint64); end = np.zeros(n, dtype=np.uint64)
        check(lib().fmd_bsearch_batch(self.h, n, _ptr(flat), _ptr(off), _ptr(cnt), _ptr(beg), _ptr(end)))
        return cnt, beg, end

    def retrieve(self, xs, stride=256):
        """fm_retrieve for each x, returned in READ order (the C ABI returns it reversed, as
        exact.c:59 does; this wrapper applies the callers' seq_reverse, unitig.c:285)."""
        xs = np.ascontiguousarray(xs, dtype=np.uint64)
        seqs = np.zeros((len(xs), stride), dtype=np.uint8); ln = np.zeros(len(xs), dtype=np.uint32)
        rank = np.zeros(len(xs), dtype=np.uint64)
        check(lib().fmd_retrieve_batch(self.h, len(xs), _ptr(xs), _ptr(seqs), stride, _ptr(ln), _ptr(rank)))
        out = np.zeros_like(seqs)
        for i in range(len(xs)):
            l = min(int(ln[i]), stride)
            out[i, :l] = seqs[i, :l][::-1]
        return out, ln.astype(np.int32), rank


def _ovlp(self, ids, min_match, max_len=100, max_nei=4, check_left=True):
    """Per-id overlap records (see include/fmd_hip.h): (rec[OVLP_DT], nei[n, max_nei, INTV_DT], seq[n, 2*max_len])."""
    ids = np.ascontiguousarray(ids, dtype=np.uint64)
    n = len(ids)
    stride = 2 * ((max_len + 3) // 4 * 4)
    rec = np.zeros(n, dtype=OVLP_DT); nei = np.zeros((n, max_nei), dtype=INTV_DT); seq = np.zeros((n, stride), dtype=np.uint8)
    check(lib().fmd_ovlp_batch(self.h, n, _ptr(ids), min_match, max_len, max_nei, _ptr(rec), _ptr(nei), _ptr(seq), stride, int(check_left)))
    return rec, nei, seq


DevIndex.overlap = _ovlp


def _ovlp_sorted(self, ids, min_match, max_len=100, max_nei=4, batch=0, check_left=False):
    """fmd_ovlp_sorted_dev over device copies: the records of ALL ids from one job (every strand 32 bases in, the strands sorted by
    the minimizer of those bases, the rest in that order, `batch` strands at a time).  Same arrays as overlap()."""
    L = lib()
    ids = np.ascontiguousarray(ids, dtype=np.uint64)
    n = len(ids)
    stride = 2 * ((max_len + 3) // 4 * 4)
    rec = np.zeros(n, dtype=OVLP_DT); nei = np.zeros((n, max_nei), dtype=INTV_DT); seq = np.zeros((n, stride), dtype=np.uint8)
    if n == 0:
        return rec, nei, seq
    wb = L.fmd_ovlp_sorted_work_bytes(n, batch, max_len, min_match)
    ptrs = []
    try:
        for b in (ids.nbytes, rec.nbytes, nei.nbytes, seq.nbytes, wb):
            p = C.c_void_p()
            check(L.fmd_dev_malloc(self.device, max(b, 16), C.byref(p)))
            ptrs.append(p)
        d_ids, d_rec, d_nei, d_seq, d_work = ptrs
        check(L.fmd_memcpy_h2d(d_ids, _ptr(ids), ids.nbytes, None))
        for d, a in ((d_rec, rec), (d_nei, nei), (d_seq, seq)):   # zeroes: rows of short / contained strands are not written
            check(L.fmd_memcpy_h2d(d, _ptr(a), a.nbytes, None))
        check(L.fmd_ovlp_sorted_dev(self.h, None, n, d_ids, min_match, max_len, max_nei, d_rec, d_nei, d_seq, stride, d_work, wb, batch))
        if check_left:
            m = batch if 0 < batch < n else n
            for o in range(0, n, m):
                c = min(m, n - o)
                check(L.fmd_ovlp_check_left_dev(self.h, None, c, min_match, max_len, C.c_void_p(d_rec.value + 64 * o), C.c_void_p(d_seq.value + stride * o), stride, d_work, wb))
        check(L.fmd_dev_sync(self.h, None))
        for d, a in ((d_rec, rec), (d_nei, nei), (d_seq, seq)):
            check(L.fmd_memcpy_d2h(_ptr(a), d, a.nbytes, None))
        return rec, nei, seq
    finally:
        for p in ptrs:
            L.fmd_dev_free(p)


DevIndex.overlap_sorted = _ovlp_sorted


def _ovlp_pack(self, rec, nei, seq):
    """fmd_ovlp_pack_dev over host copies of a finished batch: (prec[OVLP_DT], off[u64, n+1], var[u8])."""
    L = lib()
    n = len(rec)
    max_nei, stride = nei.shape[1], seq.shape[1]
    rec = np.ascontiguousarray(rec); nei = np.ascontiguousarray(nei); seq = np.ascontiguousarray(seq)
    cap = L.fmd_ovlp_pack_max_bytes(n, max_nei, stride)
    wb = L.fmd_ovlp_pack_work_bytes(n)
    sizes = [max(rec.nbytes, 16), max(nei.nbytes, 16), max(seq.nbytes, 16), max(rec.nbytes, 16), (n + 1) * 8, max(cap, 16), wb]
    ptrs = []
    try:
        for b in sizes:
            p = C.c_void_p()
            check(L.fmd_dev_malloc(self.device, b, C.byref(p)))
            ptrs.append(p)
        d_rec, d_nei, d_seq, d_prec, d_off, d_var, d_work = ptrs
        for d, a in ((d_rec, rec), (d_nei, nei), (d_seq, seq)):
            if a.nbytes:
                check(L.fmd_memcpy_h2d(d, _ptr(a), a.nbytes, None))
        check(L.fmd_ovlp_pack_dev(self.h, None, n, d_rec, d_nei, max_nei, d_seq, stride, d_prec, d_off, d_var, cap, d_work, wb))
        prec = np.zeros(n, dtype=OVLP_DT); off = np.zeros(n + 1, dtype=np.uint64)
        if n:
            check(L.fmd_memcpy_d2h(_ptr(prec), d_prec, prec.nbytes, None))
        check(L.fmd_memcpy_d2h(_ptr(off), d_off, off.nbytes, None))
        var = np.zeros(int(off[n]), dtype=np.uint8)
        if len(var):
            check(L.fmd_memcpy_d2h(_ptr(var), d_var, var.nbytes, None))
        return prec, off, var
    finally:
        for p in ptrs:
            L.fmd_dev_free(p)


DevIndex.overlap_pack = _ovlp_pack


def _kmer_collect(self, w, min_occ, suf_len=None):
    """fm6_traverse + ec_collect over all buckets: (bucket u32[], key u32[], val u8[], cnt[2])."""
    if suf_len is None:
        suf_len = w - 15 if w > 15 else 1   # compute_SUF (correct.c:319)
    b, k, v, n = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_uint64()
    cnt = (C.c_int64 * 2)()
    check(lib().fmd_kmer_collect(self.h, w, min_occ, suf_len, C.byref(b), C.byref(k), C.byref(v), C.byref(n), cnt))
    m = n.value

    def take(p, ct, dt):   # (ctypes.string_at stops at 2^31 bytes; a 5*10^10-symbol index has 1.7*10^9 solid k-mers)
        return np.ctypeslib.as_array((ct * max(m, 1)).from_address(p.value))[:m].astype(dt, copy=True) if m else np.zeros(0, dtype=dt)
    out = (take(b, C.c_uint32, np.uint32), take(k, C.c_uint32, np.uint32), take(v, C.c_uint8, np.uint8), [cnt[0], cnt[1]])
    for p in (b, k, v):
        lib().fmd_host_free(p)
    return out


DevIndex.kmer_collect = _kmer_collect


def _smem(self, seqs, self_match=0, max_mem=32):
    """fm6_smem for each read: list of INTV_DT arrays (one per read)."""
    flat, off = flatten_reads(seqs)
    n = len(off) - 1
    max_len = int(np.max(np.diff(off))) if n else 1
    mem = np.zeros((n, max_mem), dtype=INTV_DT); n_mem = np.zeros(n, dtype=np.uint32)
    check(lib().fmd_smem_batch(self.h, n, _ptr(flat), _ptr(off), self_match, max(max_len, 1), max_mem, _ptr(mem), _ptr(n_mem)))
    if (n_mem >> 31).any():
        raise FmdError("smem: capacity exceeded for %d reads (raise max_mem)" % int((n_mem >> 31).sum()))
    return [mem[i, :n_mem[i]].copy() for i in range(n)]


DevIndex.smem = _smem

SMEM_WIN_DT = np.dtype([("seq_off", "<u8"), ("seq_len", "<u4"), ("start", "<u4"), ("stop", "<u4"), ("reserved", "<u4")])


def _reach(self, seqs_zero_terminated):
    """len[p] = longest prefix of buffer[p..] (up to the next 0 byte) that occurs in the index."""
    buf = np.ascontiguousarray(seqs_zero_terminated, dtype=np.uint8)
    out = np.zeros(len(buf), dtype=np.uint32)
    check(lib().fmd_reach_batch(self.h, len(buf), _ptr(buf), _ptr(out)))
    return out


DevIndex.reach = _reach


def _smem_chain(self, seq, max_len=256, self_match=0, full_only=False):
    """fm6_smem of one long sequence, exactly as the reference walks it: reach -> chain positions ->
    one fm6_smem1_core item per position, results concatenated in chain order."""
    seq = np.ascontiguousarray(seq, dtype=np.uint8)
    L = len(seq)
    flat = np.zeros(L + 8, dtype=np.uint8); flat[:L] = seq
    reach = self.reach(flat[:L + 1])
    pos = []
    x = 0
    while x < L:
        pos.append(x)
        x += int(reach[x]) if reach[x] else 1
    n = len(pos)
    wins = np.zeros(n, dtype=SMEM_WIN_DT)
    wins["seq_len"] = L; wins["start"] = pos; wins["stop"] = np.array(pos) + 1; wins["reserved"] = 1 if full_only else 0
    max_mem = 2 * max_len + 2
    mem = np.zeros((n, max_mem), dtype=INTV_DT); n_mem = np.zeros(n, dtype=np.uint32)
    check(lib().fmd_smem_win_batch(self.h, n, _ptr(flat), L, _ptr(wins), self_match, max_len, max_mem, _ptr(mem), _ptr(n_mem)))
    if (n_mem >> 31).any():
        raise FmdError("smem_chain: capacity exceeded in %d calls" % int((n_mem >> 31).sum()))
    return np.concatenate([mem[i, :n_mem[i]] for i in range(n)]) if n else np.zeros(0, dtype=INTV_DT)


DevIndex.smem_chain = _smem_chain


def build_index_inplace(reads_2d, device=0, pieces=3):
    """The index of equal-length reads through the in-place builder (fmd_builder_*: packed text, BWT slices written
    straight into the device layout, reads appended in `pieces` calls) -> DevIndex."""
    L = lib()
    reads_2d = np.ascontiguousarray(reads_2d, dtype=np.uint8)
    n, ln = reads_2d.shape
    b = C.c_void_p()
    check(L.fmd_builder_new(device, n, ln, C.byref(b)))
    try:
        step = max(1, (n + pieces - 1) // pieces)
        for s in range(0, n, step):
            part = reads_2d[s:s + step]
            d = C.c_void_p()
            check(L.fmd_dev_malloc(device, part.nbytes + 16, C.byref(d)))
            try:
                check(L.fmd_memcpy_h2d(d, _ptr(part), part.nbytes, None))
                check(L.fmd_builder_add_dev(b, None, len(part), d))
                check(L.fmd_memcpy_d2h(_ptr(np.zeros(1, np.uint8)), d, 1, None))   # the add kernel has read the piece before it is freed
            finally:
                L.fmd_dev_free(d)
        h = C.c_void_p()
        check(L.fmd_builder_finish(b, C.byref(h)))
        b = None
        return DevIndex(h)
    finally:
        if b:
            L.fmd_builder_free(b)


def build_bwt(seqs, device=0):
    """`fermi build` BWT of a read collection (list or 2-D array of nt6 reads), computed on the GPU."""
    flat, off = flatten_reads(seqs)
    n = len(off) - 1
    bwt = np.zeros(2 * (int(off[n]) + n), dtype=np.uint8)
    n_sym = C.c_uint64(0)
    check(lib().fmd_build_bwt(device, n, _ptr(flat), _ptr(off), _ptr(bwt), C.byref(n_sym)))
    assert n_sym.value == len(bwt)
    return bwt


def probe_gather(ws_bytes, line_bytes, n_access, iters=3, device=0):
    ms = C.c_float(0)
    check(lib().fmd_probe_gather(device, ws_bytes, line_bytes, n_access, iters, C.byref(ms)))
    return ms.value
