"""ctypes bindings to the COMPILED REFERENCE (oracle/_ref/libfermi_ref.so, built by
`make -C oracle ref` from /root/reference in place).  Test infrastructure only: used to pin
the oracle and to generate tests/golden/ vectors (tests/golden/make_golden.py)."""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libfermi_ref.so")
REF_BIN = os.path.join(ROOT, "oracle", "_ref", "fermi")


def available():
    return os.path.exists(REF_SO) and os.path.exists(REF_BIN)


class RldT(C.Structure):  # field layout of rld_t (rld.h:20-39), needed to read cnt/mcnt
    _fields_ = [("asize", C.c_uint8), ("asize1", C.c_uint8), ("abits", C.c_int8), ("sbits", C.c_int8),
                ("ibits", C.c_int8), ("offset0", C.c_int8 * 2), ("ssize", C.c_int), ("n", C.c_int),
                ("n_bytes", C.c_uint64), ("z", C.c_void_p), ("cnt", C.POINTER(C.c_uint64)),
                ("mcnt", C.POINTER(C.c_uint64)), ("n_frames", C.c_uint64), ("frame", C.POINTER(C.c_uint64)),
                ("fd", C.c_int), ("mem", C.c_void_p)]


class Intv(C.Structure):  # fmintv_t (fermi.h:13-16)
    _fields_ = [("x", C.c_uint64 * 3), ("info", C.c_uint64)]


class IntvV(C.Structure):  # fmintv_v (fermi.h:21)
    _fields_ = [("n", C.c_size_t), ("m", C.c_size_t), ("a", C.POINTER(Intv))]


class KString(C.Structure):  # kstring_t (fermi.h:44-48)
    _fields_ = [("l", C.c_uint32), ("m", C.c_uint32), ("s", C.c_void_p)]


class I32V(C.Structure):  # fm32s_v
    _fields_ = [("n", C.c_size_t), ("m", C.c_size_t), ("a", C.c_void_p)]


INTV_DT = np.dtype([("x", "<u8", 3), ("info", "<u8")])

_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(REF_SO)
        L.rld_restore.restype = C.POINTER(RldT)
        L.rld_restore.argtypes = [C.c_char_p]
        L.rld_destroy.argtypes = [C.POINTER(RldT)]
        L.rld_rank1a.restype = C.c_int
        L.rld_rank1a.argtypes = [C.POINTER(RldT), C.c_uint64, C.c_void_p]
        L.rld_rank2a.restype = None
        L.rld_rank2a.argtypes = [C.POINTER(RldT), C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p]
        L.fm6_extend.argtypes = [C.POINTER(RldT), C.c_void_p, C.c_void_p, C.c_int]
        L.fm6_extend0.argtypes = [C.POINTER(RldT), C.c_void_p, C.c_void_p, C.c_int]
        L.fm_backward_search.restype = C.c_uint64
        L.fm_backward_search.argtypes = [C.POINTER(RldT), C.c_int, C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.fm_retrieve.restype = C.c_int64
        L.fm_retrieve.argtypes = [C.POINTER(RldT), C.c_uint64, C.POINTER(KString)]
        L.fm6_smem.restype = C.c_int
        L.fm6_smem.argtypes = [C.POINTER(RldT), C.c_int, C.c_void_p, C.POINTER(IntvV), C.c_int]
        L.fm6_traverse.restype = C.POINTER(Intv)
        L.fm6_traverse.argtypes = [C.POINTER(RldT), C.c_int]
        L.fm6_is_contained.restype = C.c_int
        L.fm6_is_contained.argtypes = [C.POINTER(RldT), C.c_int, C.POINTER(KString), C.POINTER(Intv), C.POINTER(IntvV)]
        L.fm6_get_nei.restype = C.c_int
        L.fm6_get_nei.argtypes = [C.POINTER(RldT), C.c_int, C.c_int, C.POINTER(KString), C.POINTER(IntvV),
                                  C.POINTER(IntvV), C.POINTER(IntvV), C.POINTER(I32V), C.c_void_p, C.c_void_p]
        L.seq_reverse.argtypes = [C.c_int, C.c_void_p]
        _lib = L
    return _lib


class RefIndex:
    """A reference rld_t loaded with rld_restore (rld.c:288)."""

    def __init__(self, fn):
        self.L = lib()
        self.e = self.L.rld_restore(fn.encode())
        if not self.e:
            raise IOError("rld_restore failed: " + fn)
        self.cnt = np.array([self.e.contents.cnt[i] for i in range(7)], dtype=np.uint64)
        self.mcnt = np.array([self.e.contents.mcnt[i] for i in range(7)], dtype=np.uint64)

    def close(self):
        if self.e:
            self.L.rld_destroy(self.e)
            self.e = None

    def rank1a(self, ks):
        ks = np.asarray(ks, dtype=np.uint64)
        ok = np.zeros((len(ks), 6), dtype=np.uint64)
        sym = np.zeros(len(ks), dtype=np.int8)
        buf = (C.c_uint64 * 6)()
        for i, k in enumerate(ks):
            sym[i] = self.L.rld_rank1a(self.e, int(k), buf)
            ok[i] = buf[:]
        return ok, sym

    def rank2a(self, ks, ls):
        ok = np.zeros((len(ks), 6), dtype=np.uint64)
        ol = np.zeros((len(ks), 6), dtype=np.uint64)
        a, b = (C.c_uint64 * 6)(), (C.c_uint64 * 6)()
        for i, (k, l) in enumerate(zip(ks, ls)):
            self.L.rld_rank2a(self.e, int(k), int(l), a, b)
            ok[i] = a[:]
            ol[i] = b[:]
        return ok, ol

    def extend(self, iks, is_back):
        iks = np.ascontiguousarray(iks, dtype=INTV_DT)
        out = np.zeros((len(iks), 6), dtype=INTV_DT)
        for i in range(len(iks)):
            self.L.fm6_extend(self.e, iks[i:i + 1].ctypes.data, out[i].ctypes.data, int(is_back[i]))
        return out

    def backward_search(self, seqs):
        n = len(seqs)
        cnt = np.zeros(n, dtype=np.uint64); beg = np.zeros(n, dtype=np.uint64); end = np.zeros(n, dtype=np.uint64)
        b, e_ = C.c_uint64(0), C.c_uint64(0)
        for i, s in enumerate(seqs):
            s = np.ascontiguousarray(s, dtype=np.uint8)
            b.value = 0; e_.value = 0
            cnt[i] = self.L.fm_backward_search(self.e, len(s), s.ctypes.data, C.byref(b), C.byref(e_))
            beg[i], end[i] = b.value, e_.value
        return cnt, beg, end

    def retrieve(self, x):
        """fm_retrieve (exact.c:59): returns (sequence in read order, rank)."""
        ks = KString(0, 0, None)
        k = self.L.fm_retrieve(self.e, int(x), C.byref(ks))
        s = np.frombuffer(C.string_at(ks.s, ks.l), dtype=np.uint8).copy()[::-1].copy()
        C.CDLL(None).free(C.c_void_p(ks.s))
        return s, k

    def smem(self, seq, self_match=0):
        v = IntvV(0, 0, None)
        seq = np.ascontiguousarray(seq, dtype=np.uint8)
        n = self.L.fm6_smem(self.e, len(seq), seq.ctypes.data, C.byref(v), self_match)
        out = np.zeros(n, dtype=INTV_DT)
        if n:
            C.memmove(out.ctypes.data, v.a, n * 32)
        C.CDLL(None).free(v.a)
        return out

    def traverse(self, depth):
        p = self.L.fm6_traverse(self.e, depth)
        n = 1 << (2 * depth)
        out = np.zeros(n, dtype=INTV_DT)
        C.memmove(out.ctypes.data, p, n * 32)
        C.CDLL(None).free(p)
        return out

    def overlap(self, seq_id, min_match):
        """Per-read overlap record: fm_retrieve -> seq_reverse ->
        fm6_is_contained -> fm6_get_nei(used=NULL).  Returns a dict."""
        libc = C.CDLL(None)
        libc.malloc.restype = C.c_void_p
        ks = KString(0, 0, None)
        rank = self.L.fm_retrieve(self.e, int(seq_id), C.byref(ks))
        self.L.seq_reverse(ks.l, ks.s)
        rec = {"id": int(seq_id), "rank": int(rank), "len": int(ks.l)}
        if ks.l <= min_match:
            rec["status"] = -1  # too short (unitig.c:288)
            libc.free(C.c_void_p(ks.s))
            return rec
        intv = Intv()
        a0, a1, nei = IntvV(0, 0, None), IntvV(0, 0, None), IntvV(0, 0, None)
        cat = I32V(0, 0, None)
        ret = self.L.fm6_is_contained(self.e, min_match, C.byref(ks), C.byref(intv), C.byref(a0))
        rec["intv"] = (intv.x[0], intv.x[1], intv.x[2])
        rec["contained"] = int(ret)
        rec["n_ovlp"] = int(a0.n)
        rec["rbeg"] = -1
        rec["nei"] = []
        rec["ext"] = b""
        if ret >= 0 and a0.n:
            rbeg = self.L.fm6_get_nei(self.e, min_match, 0, C.byref(ks), C.byref(nei), C.byref(a0), C.byref(a1),
                                      C.byref(cat), None, None)
            rec["rbeg"] = int(rbeg)
            rec["nei"] = [(nei.a[i].x[0], nei.a[i].x[1], nei.a[i].x[2], nei.a[i].info) for i in range(nei.n)]
            rec["ext"] = C.string_at(ks.s, ks.l)[rec["len"]:]
        for p in (a0.a, a1.a, nei.a):
            if p:
                libc.free(p)
        if cat.a:
            libc.free(C.c_void_p(cat.a))
        libc.free(C.c_void_p(ks.s))
        return rec
