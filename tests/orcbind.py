"""ctypes bindings to oracle/liboracle.so (our CPU restatement; test infrastructure only)."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
ORC_SO = os.path.join(ROOT, "oracle", "liboracle.so")
INTV_DT = np.dtype([("x", "<u8", 3), ("info", "<u8")])


class IntvV(C.Structure):
    _fields_ = [("n", C.c_size_t), ("m", C.c_size_t), ("a", C.c_void_p)]


class StrT(C.Structure):
    _fields_ = [("n", C.c_size_t), ("m", C.c_size_t), ("s", C.c_void_p)]


class SolidT(C.Structure):
    _fields_ = [("n", C.c_size_t), ("m", C.c_size_t), ("key", C.c_void_p), ("val", C.c_void_p), ("cnt", C.c_int64 * 2)]


class Counters(C.Structure):
    _fields_ = [("rank1a", C.c_uint64), ("rank2a", C.c_uint64), ("rank2a_spill", C.c_uint64)]


class RldT(C.Structure):
    _fields_ = [("n_words", C.c_uint64), ("w", C.c_void_p), ("n_frames", C.c_uint64), ("frame", C.c_void_p),
                ("ibits", C.c_int), ("cnt", C.c_uint64 * 7), ("mcnt", C.c_uint64 * 7)]


_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "oracle"])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(ORC_SO):
            build()
        L = C.CDLL(ORC_SO)
        P = C.POINTER(RldT)
        L.orc_rld_load.restype = P; L.orc_rld_load.argtypes = [C.c_char_p]
        L.orc_rld_from_bwt.restype = P; L.orc_rld_from_bwt.argtypes = [C.c_void_p, C.c_uint64]
        L.orc_rld_free.argtypes = [P]
        L.orc_rld_dump.argtypes = [P, C.c_char_p]
        L.orc_rld_decode_all.restype = C.c_uint64; L.orc_rld_decode_all.argtypes = [P, C.c_void_p]
        L.orc_counters_read.restype = Counters
        L.orc_rank1a_batch.argtypes = [P, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_rank2a_batch.argtypes = [P, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_extend_batch.argtypes = [P, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_retrieve_batch.argtypes = [P, C.c_size_t, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_backward_search_batch.argtypes = [P, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.orc_traverse.restype = C.c_void_p; L.orc_traverse.argtypes = [P, C.c_int]
        L.orc_smem.restype = C.c_int; L.orc_smem.argtypes = [P, C.c_int, C.c_void_p, C.POINTER(IntvV), C.c_int]
        L.orc_smem_batch.argtypes = [P, C.c_size_t, C.c_int, C.c_void_p, C.c_int, C.c_uint32, C.c_void_p, C.c_void_p, C.c_int]
        L.orc_ec_range.argtypes = [P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_is_contained.restype = C.c_int
        L.orc_is_contained.argtypes = [P, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.POINTER(IntvV)]
        L.orc_get_nei.restype = C.c_int
        L.orc_get_nei.argtypes = [P, C.c_int, C.c_int, C.POINTER(StrT), C.POINTER(IntvV), C.POINTER(IntvV), C.POINTER(IntvV)]
        L.orc_overlap_batch.argtypes = [P, C.c_size_t, C.c_void_p, C.c_int, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_int]
        L.orc_ec_collect.argtypes = [P, C.c_int, C.c_int, C.c_int, C.c_void_p, C.POINTER(SolidT)]
        L.orc_ectab_new.restype = C.c_void_p
        L.orc_ectab_new.argtypes = [C.c_int, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_ectab_free.argtypes = [C.c_void_p]
        L.orc_ecfix_batch.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        _lib = L
    return _lib


_libc = C.CDLL(None)
_libc.malloc.restype = C.c_void_p
_libc.malloc.argtypes = [C.c_size_t]
_libc.free.argtypes = [C.c_void_p]


def _take(v, dt=INTV_DT, itemsize=32):
    out = np.zeros(v.n, dtype=dt)
    if v.n:
        C.memmove(out.ctypes.data, v.a, v.n * itemsize)
    return out


class OrcIndex:
    def __init__(self, fn=None, bwt=None):
        self.L = lib()
        if fn is not None:
            self.e = self.L.orc_rld_load(fn.encode())
        else:
            bwt = np.ascontiguousarray(bwt, dtype=np.uint8)
            self.e = self.L.orc_rld_from_bwt(bwt.ctypes.data, len(bwt))
        if not self.e:
            raise IOError("oracle load failed")
        self.cnt = np.array(self.e.contents.cnt[:], dtype=np.uint64)
        self.mcnt = np.array(self.e.contents.mcnt[:], dtype=np.uint64)
        self.n = int(self.mcnt[0])

    def close(self):
        if self.e:
            self.L.orc_rld_free(self.e)
            self.e = None

    def dump(self, fn):
        return self.L.orc_rld_dump(self.e, fn.encode())

    def counters(self):
        c = self.L.orc_counters_read()
        return {"rank1a": c.rank1a, "rank2a": c.rank2a, "rank2a_spill": c.rank2a_spill}

    def decode_all(self):
        out = np.zeros(self.n, dtype=np.uint8)
        n = self.L.orc_rld_decode_all(self.e, out.ctypes.data)
        assert n == self.n, (n, self.n)
        return out

    def rank1a(self, ks):
        ks = np.ascontiguousarray(ks, dtype=np.uint64)
        ok = np.zeros((len(ks), 6), dtype=np.uint64); sym = np.zeros(len(ks), dtype=np.int8)
        self.L.orc_rank1a_batch(self.e, len(ks), ks.ctypes.data, ok.ctypes.data, sym.ctypes.data)
        return ok, sym

    def rank2a(self, ks, ls):
        ks = np.ascontiguousarray(ks, dtype=np.uint64); ls = np.ascontiguousarray(ls, dtype=np.uint64)
        ok = np.zeros((len(ks), 6), dtype=np.uint64); ol = np.zeros((len(ks), 6), dtype=np.uint64)
        self.L.orc_rank2a_batch(self.e, len(ks), ks.ctypes.data, ls.ctypes.data, ok.ctypes.data, ol.ctypes.data)
        return ok, ol

    def extend(self, iks, is_back):
        iks = np.ascontiguousarray(iks, dtype=INTV_DT); is_back = np.ascontiguousarray(is_back, dtype=np.uint8)
        out = np.zeros((len(iks), 6), dtype=INTV_DT)
        self.L.orc_extend_batch(self.e, len(iks), iks.ctypes.data, is_back.ctypes.data, out.ctypes.data)
        return out

    def backward_search(self, seqs, n_threads=1):
        seqs = np.ascontiguousarray(seqs, dtype=np.uint8)
        n, ln = seqs.shape
        cnt = np.zeros(n, dtype=np.uint64); beg = np.zeros(n, dtype=np.uint64); end = np.zeros(n, dtype=np.uint64)
        self.L.orc_backward_search_batch(self.e, n, ln, seqs.ctypes.data, cnt.ctypes.data, beg.ctypes.data, end.ctypes.data, n_threads)
        return cnt, beg, end

    def retrieve(self, xs, stride=256):
        xs = np.ascontiguousarray(xs, dtype=np.uint64)
        seqs = np.zeros((len(xs), stride), dtype=np.uint8); ln = np.zeros(len(xs), dtype=np.int32); rank = np.zeros(len(xs), dtype=np.uint64)
        self.L.orc_retrieve_batch(self.e, len(xs), xs.ctypes.data, seqs.ctypes.data, stride, ln.ctypes.data, rank.ctypes.data)
        return seqs, ln, rank

    def traverse(self, depth):
        p = self.L.orc_traverse(self.e, depth)
        n = 1 << (2 * depth)
        out = np.zeros(n, dtype=INTV_DT)
        C.memmove(out.ctypes.data, p, n * 32)
        _libc.free(p)
        return out

    def smem(self, seq, self_match=0):
        seq = np.ascontiguousarray(seq, dtype=np.uint8)
        v = IntvV(0, 0, None)
        self.L.orc_smem(self.e, len(seq), seq.ctypes.data, C.byref(v), self_match)
        out = _take(v)
        _libc.free(v.a)
        return out

    def overlap(self, seq_id, min_match):
        """Same record as refbind.RefIndex.overlap, computed by the oracle."""
        seqs, ln, rank = self.retrieve([seq_id], stride=4096)
        L0 = int(ln[0])
        rec = {"id": int(seq_id), "rank": int(rank[0]), "len": L0}
        if L0 <= min_match:
            rec["status"] = -1
            return rec
        intv = np.zeros(1, dtype=INTV_DT)
        a0, a1, nei = IntvV(0, 0, None), IntvV(0, 0, None), IntvV(0, 0, None)
        s = seqs[0, :L0].copy()
        ret = self.L.orc_is_contained(self.e, min_match, s.ctypes.data, L0, intv.ctypes.data, C.byref(a0))
        rec["intv"] = tuple(int(v) for v in intv[0]["x"])
        rec["contained"] = int(ret)
        rec["n_ovlp"] = int(a0.n)
        rec["rbeg"] = -1; rec["nei"] = []; rec["ext"] = b""
        if ret >= 0 and a0.n:
            cap = L0 + 4096
            buf = _libc.malloc(cap)
            C.memmove(buf, s.ctypes.data, L0)
            st = StrT(L0, cap, buf)
            rbeg = self.L.orc_get_nei(self.e, min_match, 0, C.byref(st), C.byref(nei), C.byref(a0), C.byref(a1))
            rec["rbeg"] = int(rbeg)
            nv = _take(nei)
            rec["nei"] = [(int(r["x"][0]), int(r["x"][1]), int(r["x"][2]), int(r["info"])) for r in nv]
            rec["ext"] = C.string_at(st.s, st.n)[L0:]
            _libc.free(st.s)
        for p in (a0.a, a1.a, nei.a):
            if p:
                _libc.free(p)
        return rec

    def overlap_batch(self, ids, min_match, max_len=100, max_nei=4, n_threads=1, check_left=True):
        """Array form of overlap(): same (rec, nei, seq) triple as fermi_amd.api.DevIndex.overlap."""
        from fermi_amd.api import OVLP_DT
        ids = np.ascontiguousarray(ids, dtype=np.uint64)
        n = len(ids)
        stride = 2 * ((max_len + 3) // 4 * 4)
        rec = np.zeros(n, dtype=OVLP_DT); nei = np.zeros((n, max_nei), dtype=INTV_DT); seq = np.zeros((n, stride), dtype=np.uint8)
        self.L.orc_overlap_batch(self.e, n, ids.ctypes.data, min_match, max_nei, rec.ctypes.data, nei.ctypes.data,
                                 seq.ctypes.data, stride, n_threads, int(check_left))
        return rec, nei, seq

    def smem_batch(self, seqs, self_match=0, max_mem=32, n_threads=1):
        """fm6_smem over an (n, len) uint8 array: (mem[n, max_mem] INTV_DT, n_mem[n]) as fmd_smem_dev writes them."""
        seqs = np.ascontiguousarray(seqs, dtype=np.uint8)
        n, ln = seqs.shape
        mem = np.zeros((n, max_mem), dtype=INTV_DT); n_mem = np.zeros(n, dtype=np.uint32)
        self.L.orc_smem_batch(self.e, n, ln, seqs.ctypes.data, self_match, max_mem, mem.ctypes.data, n_mem.ctypes.data, n_threads)
        return mem, n_mem

    def ec_range(self, w, min_occ, suf_len, b0, b1, n_threads=1):
        """ec_collect over suffix buckets [b0, b1): (bucket, key, val, seconds)."""
        pb, pk, pv = C.c_void_p(), C.c_void_p(), C.c_void_p()
        n, secs = C.c_uint64(), C.c_double()
        self.L.orc_ec_range(self.e, w, min_occ, suf_len, b0, b1, n_threads, C.byref(pb), C.byref(pk), C.byref(pv), C.byref(n), C.byref(secs))
        m = n.value
        B = np.zeros(m, dtype=np.uint32); K = np.zeros(m, dtype=np.uint32); V = np.zeros(m, dtype=np.uint8)
        if m:
            C.memmove(B.ctypes.data, pb, m * 4); C.memmove(K.ctypes.data, pk, m * 4); C.memmove(V.ctypes.data, pv, m)
        for p in (pb, pk, pv):
            _libc.free(p)
        return B, K, V, secs.value

    def ec_collect(self, w, min_occ, suf_len):
        top = self.traverse(suf_len)
        out = []
        cnt = [0, 0]
        for b in range(len(top)):
            so = SolidT(0, 0, None, None, (C.c_int64 * 2)(0, 0))
            self.L.orc_ec_collect(self.e, w, min_occ, suf_len, top[b:b + 1].ctypes.data, C.byref(so))
            k = np.zeros(so.n, dtype=np.uint32); v = np.zeros(so.n, dtype=np.uint8)
            if so.n:
                C.memmove(k.ctypes.data, so.key, so.n * 4); C.memmove(v.ctypes.data, so.val, so.n)
                _libc.free(so.key); _libc.free(so.val)
            out.append((k, v))
            cnt[0] += so.cnt[0]; cnt[1] += so.cnt[1]
        return out, cnt


def ec_fix(w, bucket, key, val, seqs_nt6, quals, step=5):
    """ec_fix (correct.c:232-246) of a list of reads against a (bucket, key, val) table: lists of corrected nt6 arrays
    and qualities (phred + 33 bytes) and the info array -- the contract of fmd_ecfix_batch (include/fmd_hip.h)."""
    L = lib()
    suf_len = w - 15 if w > 15 else 1
    bucket = np.ascontiguousarray(bucket, dtype=np.uint32); key = np.ascontiguousarray(key, dtype=np.uint32); val = np.ascontiguousarray(val, dtype=np.uint8)
    t = L.orc_ectab_new(suf_len, len(key), bucket.ctypes.data, key.ctypes.data, val.ctypes.data)
    assert t
    n = len(seqs_nt6)
    off = np.zeros(n + 1, dtype=np.uint64)
    np.cumsum([len(x) for x in seqs_nt6], out=off[1:])
    s = np.concatenate([np.asarray(x, dtype=np.uint8) for x in seqs_nt6] + [np.zeros(8, np.uint8)])
    q = np.concatenate([np.asarray(x, dtype=np.uint8) for x in quals] + [np.zeros(8, np.uint8)])
    info = np.zeros(n, dtype=np.int32)
    L.orc_ecfix_batch(t, w, step, n, s.ctypes.data, q.ctypes.data, off.ctypes.data, info.ctypes.data)
    L.orc_ectab_free(t)
    return s[: int(off[n])], q[: int(off[n])], off, info
