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
