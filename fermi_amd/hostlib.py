"""ctypes view of libfmdhost.so (fermi_amd/host/*.c): file-format helpers in plain C."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libfmdhost.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("libfmdhost.so is not built; run `make host`")
        L = C.CDLL(LIB_PATH)
        L.fmdh_write_rld_from_rle6.argtypes = [C.c_void_p, C.c_uint64, C.c_char_p]
        L.fmdh_write_rle6.argtypes = [C.c_void_p, C.c_uint64, C.c_char_p]
        L.fmdh_write_rld_from_bwt.argtypes = [C.c_void_p, C.c_uint64, C.c_char_p]
        L.fmdh_trim_palindrome.restype = C.c_uint32
        L.fmdh_trim_palindrome.argtypes = [C.c_void_p, C.c_uint32]
        class Shard(C.Structure):
            _fields_ = [("n", C.c_uint64), ("rec", C.c_void_p), ("off", C.c_void_p), ("chunk", C.POINTER(C.c_void_p)),
                        ("chunk_shift", C.c_uint32), ("max_nei", C.c_uint32), ("seq_stride", C.c_uint32)]
        class Table(C.Structure):
            _fields_ = [("n", C.c_uint64), ("n_shards", C.c_int), ("shard", C.POINTER(Shard)), ("side_of", C.c_void_p), ("side", Shard),
                        ("row_of", C.c_void_p), ("link", C.c_void_p)]
        L.fmdh_ovlp_table_link.argtypes = [C.POINTER(Table), C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
        L.Shard = Shard
        L.Table = Table
        L.fmdh_unitig_walk.argtypes = [C.POINTER(Table), C.c_uint64, C.c_int, C.c_void_p, C.c_void_p]
        L.fmdh_slim_stats.argtypes = [C.POINTER(Table), C.POINTER(C.c_uint64)]
        L.fmdh_unitig.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_int), C.c_int, C.c_char_p, C.c_void_p]
        class EcOpt(C.Structure):
            _fields_ = [("w", C.c_int), ("min_occ", C.c_int), ("keep_bad", C.c_int), ("is_paired", C.c_int), ("trim_l", C.c_int),
                        ("step", C.c_int), ("max_corr", C.c_float)]
        L.EcOpt = EcOpt
        L.fmdh_correct_reads.argtypes = [C.POINTER(EcOpt), C.c_int, C.c_int, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_char_p, C.c_void_p]
        L.fmdh_correct_kmer.argtypes = [C.c_uint64]
        class RemapOpt(C.Structure):
            _fields_ = [("skip", C.c_int), ("min_pcv", C.c_int), ("max_dist", C.c_int)]
        L.RemapOpt = RemapOpt
        L.fmdh_remap_new.restype = C.c_void_p
        L.fmdh_remap_new.argtypes = [C.POINTER(RemapOpt), C.c_void_p, C.c_uint64]
        L.fmdh_remap_contig.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        L.fmdh_remap_finish.argtypes = [C.c_void_p, C.c_void_p]
        L.fmdh_remap.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.POINTER(RemapOpt), C.c_char_p, C.c_void_p]
        class PPart(C.Structure):
            _fields_ = [("seq", C.c_void_p), ("qual", C.c_void_p), ("len", C.c_void_p), ("n", C.c_size_t), ("bytes", C.c_size_t), ("has_qual", C.c_int), ("bad", C.c_int),
                        ("m_bytes", C.c_size_t), ("m_n", C.c_size_t)]
        L.PPart = PPart
        L.fmdh_seq_open.restype = C.c_void_p; L.fmdh_seq_open.argtypes = [C.c_char_p]
        L.fmdh_seq_read.argtypes = [C.c_void_p]
        L.fmdh_seq_bases.restype = C.c_void_p; L.fmdh_seq_bases.argtypes = [C.c_void_p]
        L.fmdh_seq_qual.restype = C.c_void_p; L.fmdh_seq_qual.argtypes = [C.c_void_p]
        L.fmdh_seq_close.restype = None; L.fmdh_seq_close.argtypes = [C.c_void_p]
        L.fmdh_pseq_open.restype = C.c_void_p; L.fmdh_pseq_open.argtypes = [C.c_char_p, C.c_int, C.c_size_t]
        L.fmdh_pseq_next.argtypes = [C.c_void_p, C.POINTER(C.POINTER(PPart)), C.POINTER(C.c_int)]
        L.fmdh_pseq_close.restype = None; L.fmdh_pseq_close.argtypes = [C.c_void_p]
        L.fmdh_api_unitig.argtypes = [C.c_int, C.c_int, C.c_int64, C.c_void_p, C.c_void_p]
        L.fmdh_api_correct.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int64, C.c_void_p, C.c_void_p]
        L.fmdh_api_seqlen.argtypes = [C.c_int64, C.c_void_p, C.c_double]
        _lib = L
    return _lib


def _chk(rc, what):
    if rc != 0:
        raise OSError(-rc, "%s failed: %s" % (what, os.strerror(-rc)))


def write_rld_from_rle6(runs, path):
    runs = np.ascontiguousarray(runs, dtype=np.uint8)
    _chk(lib().fmdh_write_rld_from_rle6(runs.ctypes.data, len(runs), path.encode()), "write_rld_from_rle6")


def write_rld_from_rle6_ptr(ptr, n, path):
    _chk(lib().fmdh_write_rld_from_rle6(ptr, n, path.encode()), "write_rld_from_rle6")


def write_rle6(runs, path):
    runs = np.ascontiguousarray(runs, dtype=np.uint8)
    _chk(lib().fmdh_write_rle6(runs.ctypes.data, len(runs), path.encode()), "write_rle6")


def write_rld_from_bwt(bwt, path):
    bwt = np.ascontiguousarray(bwt, dtype=np.uint8)
    _chk(lib().fmdh_write_rld_from_bwt(bwt.ctypes.data, len(bwt), path.encode()), "write_rld_from_bwt")


def trim_palindrome(seq):
    seq = np.ascontiguousarray(seq, dtype=np.uint8)
    return int(lib().fmdh_trim_palindrome(seq.ctypes.data, len(seq)))


_libc = C.CDLL(None)
_libc.fopen.restype = C.c_void_p
_libc.fopen.argtypes = [C.c_char_p, C.c_char_p]
_libc.fclose.argtypes = [C.c_void_p]
_libc.free.argtypes = [C.c_void_p]
_libc.free.restype = None


def unitig_walk(shards, n_seq, min_match, out_path, sorted_map=None, max_nei=4, seq_stride=256, link=0, resolve=None, stats=None):
    """Replay the `fermi unitig -t1` walk over a packed per-id overlap table kept in len(shards) shards -- id i = row
    i // N of shard i % N; each shard = (prec[OVLP_DT], off[u64, n+1], var[u8]) as fmd_ovlp_pack_dev writes them.
    Writes MAG records to out_path."""
    L = lib()
    keep = []
    arr = (L.Shard * len(shards))()
    n_total = 0
    for k, (prec, off, var) in enumerate(shards):
        prec = np.ascontiguousarray(prec); off = np.ascontiguousarray(off, dtype=np.uint64)
        var = np.ascontiguousarray(var, dtype=np.uint8) if len(var) else np.zeros(8, np.uint8)
        chunk = (C.c_void_p * 1)(var.ctypes.data)
        keep += [prec, off, var, chunk]
        arr[k] = L.Shard(len(prec), prec.ctypes.data, off.ctypes.data, chunk, 40, max_nei, seq_stride)   # one chunk: shift beyond any row count
        n_total += len(prec)
    t = L.Table(n_total, len(shards), arr, None, L.Shard(), None, None)
    undecided = None
    if link:   # the parallel link pass of the product path (rec.reserved of decided edges is written in place)
        und, n_und = C.c_void_p(), C.c_uint64()
        _chk(L.fmdh_ovlp_table_link(C.byref(t), link, C.byref(und), C.byref(n_und)), "table_link")
        undecided = np.ctypeslib.as_array(C.cast(und, C.POINTER(C.c_uint64)), (n_und.value,)).copy() if n_und.value else np.zeros(0, np.uint64)
        _libc.free(und)
        if resolve is not None and len(undecided):   # the product runs fmd_ovlp_check_left on these ids (ovlp_table.c)
            vals = resolve(undecided)
            n_sh = len(shards)
            for i, v in zip(undecided, vals):
                keep[4 * int(i % n_sh)]["reserved"][int(i // n_sh)] = v
    if stats is not None:   # what the walk's own table (host/slim_table.c) makes of these rows
        st = (C.c_uint64 * 6)()
        _chk(L.fmdh_slim_stats(C.byref(t), st), "slim_stats")
        stats.update(bytes=st[0], big=st[1], ext_in_var=st[2], plain=st[3], own_seq=st[4], undecided=st[5])
    fp = _libc.fopen(out_path.encode(), b"wb")
    try:
        sm = None if sorted_map is None else np.ascontiguousarray(sorted_map, dtype=np.uint64)
        _chk(L.fmdh_unitig_walk(C.byref(t), n_seq, min_match, None if sm is None else sm.ctypes.data, fp), "unitig_walk")
    finally:
        _libc.fclose(fp)
        _libc.free(t.row_of); _libc.free(t.link)
    return undecided


class DistRoot:
    """fmdh_dist_root_t: the root of an N-process overlap job (fermi_amd.dist.DistJob with host_table = 2) keeps no packed table -- every piece is folded into
    the rows `unitig` walks as it arrives.  sink / ctx go into the job (row_sink, sink_ctx); feed() is the same entry for callers that hold pieces themselves."""

    def __init__(self, n_seq, max_len=128):
        L = lib()
        L.fmdh_dist_root_new.restype = C.c_void_p
        L.fmdh_dist_root_new.argtypes = [C.c_uint64, C.c_uint32]
        L.fmdh_dist_root_sink.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]
        L.fmdh_dist_root_rows.restype = C.c_uint64
        L.fmdh_dist_root_rows.argtypes = [C.c_void_p]
        L.fmdh_dist_root_finish.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]
        L.fmdh_dist_root_free.argtypes = [C.c_void_p]
        L.fmdh_dist_root_free.restype = None
        L.fmdh_slim_free.argtypes = [C.c_void_p]
        L.fmdh_slim_bytes.restype = C.c_uint64
        L.fmdh_slim_bytes.argtypes = [C.c_void_p]
        L.fmdh_unitig_walk_slim.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        self.L, self.n_seq = L, n_seq
        self.ctx = L.fmdh_dist_root_new(n_seq, max_len)
        if not self.ctx:
            raise MemoryError("fmdh_dist_root_new")
        self.sink = C.cast(L.fmdh_dist_root_sink, C.c_void_p).value
        self.slim = None

    def feed(self, ids, prec, off, var, max_nei):
        ids = np.ascontiguousarray(ids, dtype=np.uint32); prec = np.ascontiguousarray(prec); off = np.ascontiguousarray(off, dtype=np.uint64)
        var = np.ascontiguousarray(var, dtype=np.uint8) if len(var) else np.zeros(8, np.uint8)
        _chk(self.L.fmdh_dist_root_sink(self.ctx, len(ids), ids.ctypes.data, prec.ctypes.data, off.ctypes.data, var.ctypes.data, max_nei), "dist_root_sink")

    def rows(self):
        return int(self.L.fmdh_dist_root_rows(self.ctx))

    def finish(self, min_match, dev_handle=None):
        """-> bytes of the table; the rows that exceeded a capacity and the edges lfork leaves open go through the GPU behind dev_handle (an api.DevIndex.h)"""
        slim = C.c_void_p()
        ctx, self.ctx = self.ctx, None
        if self.L.fmdh_dist_root_finish(ctx, dev_handle, min_match, C.byref(slim)):
            raise RuntimeError("fmdh_dist_root_finish failed")
        self.slim = slim
        return int(self.L.fmdh_slim_bytes(slim))

    def walk(self, min_match, out_path):
        fp = _libc.fopen(out_path.encode(), b"wb")
        try:
            _chk(self.L.fmdh_unitig_walk_slim(self.slim, self.n_seq, min_match, None, fp, 0), "unitig_walk_slim")
        finally:
            _libc.fclose(fp)

    def close(self):
        if self.ctx:
            self.L.fmdh_dist_root_free(self.ctx); self.ctx = None
        if self.slim:
            self.L.fmdh_slim_free(self.slim); self.slim = None


def unitig(fmd_path, min_match, out_path, devices=(0,)):
    """`fermi-amd unitig -l min_match -g d0,d1,.. fmd_path > out_path` (GPU)."""
    L = lib()
    dev = (C.c_int * len(devices))(*devices)
    fp = _libc.fopen(out_path.encode(), b"wb")
    try:
        rc = L.fmdh_unitig(fmd_path.encode(), len(devices), dev, min_match, None, fp)
    finally:
        _libc.fclose(fp)
    if rc:
        raise RuntimeError("fmdh_unitig failed")


def slim_build(fmd_path, min_match, devices=(0,)):
    """The table `fermi-amd unitig -l min_match -g d0,d1,..` walks, built and freed again: one index replica + one host thread per GPU, rows of ids
    i = g (mod G) streamed over each GPU's own PCIe link into the slim table, host threads link them.  -> dict of seconds / bytes (fmdh_slim_last_build)."""
    L = lib()
    dev = (C.c_int * len(devices))(*devices)
    slim, n_seq = C.c_void_p(), C.c_uint64()
    L.fmdh_slim_build.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
    L.fmdh_slim_free.argtypes = [C.c_void_p]
    rc = L.fmdh_slim_build(fmd_path.encode(), len(devices), dev, min_match, C.byref(slim), C.byref(n_seq))
    if rc:
        raise RuntimeError("fmdh_slim_build failed")
    t = (C.c_double * 4)()
    L.fmdh_slim_last_build(t)
    L.fmdh_slim_free(slim)
    return {"n_seq": int(n_seq.value), "index_load_s": t[0], "rows_s": t[1], "build_s": t[2], "table_bytes": t[3]}


def correct_reads(w, min_occ, bucket, key, val, fq_path, out_path, step=5, max_corr=0.3, device=0):
    """ec_fix phase of `fermi correct` against a harvested solid-k-mer table (GPU correction pass + host marking/printing)."""
    L = lib()
    opt = L.EcOpt(w, min_occ, 0, 0, 0, step, max_corr)
    bucket = np.ascontiguousarray(bucket, dtype=np.uint32); key = np.ascontiguousarray(key, dtype=np.uint32)
    val = np.ascontiguousarray(val, dtype=np.uint8)
    fp = _libc.fopen(out_path.encode(), b"wb")
    try:
        rc = L.fmdh_correct_reads(C.byref(opt), device, w - 15 if w > 15 else 1, len(key), bucket.ctypes.data, key.ctypes.data, val.ctypes.data,
                                  fq_path.encode(), fp)
    finally:
        _libc.fclose(fp)
    if rc:
        raise RuntimeError("fmdh_correct_reads failed")


def remap_contigs(contigs, mems, n_seq, out_path, err_path, skip=50, min_pcv=0, max_dist=1000, sorted_map=None):
    """The host part of `fermi remap` (paircov + printing) over precomputed SMEM lists.
    contigs: list of (name, comment or None, nt6 array); mems: one INTV_DT array per contig, sorted by start."""
    L = lib()
    opt = L.RemapOpt(skip, min_pcv, max_dist)
    sm = None if sorted_map is None else np.ascontiguousarray(sorted_map, dtype=np.uint64)
    st = L.fmdh_remap_new(C.byref(opt), None if sm is None else sm.ctypes.data, n_seq)
    fp = _libc.fopen(out_path.encode(), b"wb"); fe = _libc.fopen(err_path.encode(), b"wb")
    try:
        for (name, comment, seq), m in zip(contigs, mems):
            buf = np.zeros(len(seq) + 1, dtype=np.uint8); buf[:len(seq)] = seq
            m = np.ascontiguousarray(m)
            L.fmdh_remap_contig(st, name.encode(), None if comment is None else comment.encode(), len(seq), buf.ctypes.data,
                                m.ctypes.data, len(m), fp)
        L.fmdh_remap_finish(st, fe)
    finally:
        _libc.fclose(fp); _libc.fclose(fe)


def _read_buffer(reads):
    """list of byte strings -> one buffer with a NUL after every read (what fm6_api_readseq returns, seq.c:385-408)"""
    return np.frombuffer(b"".join(r + b"\0" for r in reads), dtype=np.uint8).copy()


def api_unitig(reads, min_match, out_path, device=0):
    """fm6_api_unitig (unitig.c:413-434) over reads held in memory: list of ASCII byte strings -> MAG text in out_path."""
    L = lib()
    buf = _read_buffer(reads)
    fp = _libc.fopen(out_path.encode(), b"wb")
    try:
        if L.fmdh_api_unitig(device, min_match, len(buf), buf.ctypes.data, fp) != 0:
            raise RuntimeError("fmdh_api_unitig failed")
    finally:
        _libc.fclose(fp)


def api_correct(reads, quals, kmer, step=5, device=0):
    """fm6_api_correct (correct.c:464-511): -> (corrected reads, qualities) as lists of byte strings (upper case = kept, lower = corrected)."""
    L = lib()
    s, q = _read_buffer(reads), _read_buffer(quals)
    if L.fmdh_api_correct(device, kmer, step, len(s), s.ctypes.data, q.ctypes.data) != 0:
        raise RuntimeError("fmdh_api_correct failed")
    return s.tobytes().split(b"\0")[:-1], q.tobytes().split(b"\0")[:-1]


def read_records_serial(path):
    """[(bases, qualities or None)] as fmdh_seq_read gives them (seqio.c: the reader `build`, `correct`, `remap` use)."""
    L = lib()
    io = L.fmdh_seq_open(path.encode())
    out = []
    while True:
        n = L.fmdh_seq_read(io)
        if n < 0:
            break
        q = L.fmdh_seq_qual(io)
        out.append((C.string_at(L.fmdh_seq_bases(io), n), C.string_at(q, n) if q else None))
    L.fmdh_seq_close(io)
    return out, n


def read_records_parallel(path, threads, span):
    """the same records through fmdh_pseq_* (seqpar.c); None if the input cannot be mapped (gzip, stdin)."""
    L = lib()
    r = L.fmdh_pseq_open(path.encode(), threads, span)
    if not r:
        return None
    out = []
    parts, n_parts = C.POINTER(L.PPart)(), C.c_int()
    while L.fmdh_pseq_next(r, C.byref(parts), C.byref(n_parts)) == 1:
        for k in range(n_parts.value):
            p = parts[k]
            if p.n == 0:   # a cut guess that ran into the end of its span: no records, len never allocated
                continue
            lens = np.ctypeslib.as_array(C.cast(p.len, C.POINTER(C.c_uint32)), (max(p.n, 1),))[: p.n]
            seq = C.string_at(p.seq, p.bytes) if p.bytes else b""
            qual = C.string_at(p.qual, p.bytes) if p.bytes else b""
            o = 0
            for ln in lens:
                ln = int(ln)
                q = qual[o:o + ln]
                out.append((seq[o:o + ln], q if p.has_qual and (ln == 0 or any(q)) else None))
                o += ln
    L.fmdh_pseq_close(r)
    return out
