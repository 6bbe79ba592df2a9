"""CPU: the C-ABI library loads, exports every symbol include/fmd_hip.h declares, and fails
loudly (no CPU fallback) when there is no GPU.  No compute calls here."""
import os
import re

import numpy as np
import pytest

from fermi_amd import api

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "fmd_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    hdr = re.sub(r"static inline[^{]*\{.*?\n\}", "", hdr, flags=re.S)  # header-only helpers are not exports
    return sorted(set(re.findall(r"\b(fmd_[a-z0-9_]+)\s*\(", hdr)))


def test_header_and_binding_lists_agree():
    assert _declared_symbols() == sorted(api.ABI_SYMBOLS)


def test_library_exports_every_declared_symbol():
    L = api.lib()
    for s in _declared_symbols():
        assert hasattr(L, s), s


def test_strerror_covers_codes():
    L = api.lib()
    for code in range(0, -8, -1):
        assert L.fmd_strerror(code)


def test_no_cpu_fallback_without_gpu():
    if api.device_count() > 0:
        pytest.skip("a GPU is present")
    bwt = np.array([1, 2, 3, 0], dtype=np.uint8)
    with pytest.raises(api.FmdError, match="no usable HIP device"):
        api.DevIndex.from_bwt(bwt)
    with pytest.raises(api.FmdError):
        api.DevIndex.open(os.path.join(ROOT, "tests", "golden", "tiny.fmd"))


def test_product_does_not_reference_oracle():
    """The product tree must never import, link or dlopen anything under oracle/."""
    bad = []
    for base in ("fermi_amd", "include"):
        for dp, _, fns in os.walk(os.path.join(ROOT, base)):
            for fn in fns:
                if fn.endswith((".py", ".c", ".h", ".hip", ".cpp")):
                    txt = open(os.path.join(dp, fn), errors="ignore").read()
                    if re.search(r"liboracle|orcbind|oracle/|fmd_oracle|libfermi_ref", txt):
                        bad.append(os.path.join(dp, fn))
    assert not bad, bad


def test_no_kernel_spills_to_scratch():
    """Every gfx950 kernel of the built library keeps its registers: no VGPR spills, no private segment (tools/kernel_resources.py reads
    the code-object metadata of build/*.o).  A spilling k_ovl_nei_fast returned wrong neighbours for a few strands in 10^7 when it ran
    beside the walk on a second stream (round 2), and a spilling kernel is never at the occupancy its launch bound names."""
    import glob, sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import kernel_resources as kr
    objs = sorted(glob.glob(os.path.join(ROOT, "build", "*.o")))
    if not objs or not os.path.exists(kr.LLVM + "/clang-offload-bundler"):
        pytest.skip("no build/*.o here (the library was built elsewhere) or no LLVM tools")
    seen = 0
    for o in objs:
        for k in kr.kernels_of(o):
            seen += 1
            name = kr.demangle(k["name"])
            assert int(k.get("vgpr_spill_count", 0)) == 0, (os.path.basename(o), name, k)
            if "rocprim" not in name and int(k.get("private_segment_fixed_size", 0)) != 0:   # (the library sort's kernels keep small private arrays; ours have none)
                # a private segment in the metadata and not one instruction that reaches it: SGPR spill slots that were placed in VGPR lanes (k_ovl_walk<WALK_HEADP>,
                # the opt-in two-base head: 106 SGPRs).  Anything that does reach scratch fails here.
                assert kr.scratch_instructions(o, k["name"]) == 0, (os.path.basename(o), name, k)
    assert seen > 40


def test_table_memory_as_file_pages(tmp_path, monkeypatch):
    """fmd_table_alloc: anonymous memory by default; with FMD_TABLE_DIR, blocks of FMD_TABLE_DIR_MIN bytes and more are pages of an
    unlinked file in that directory (the streamed form of `unitig`'s table), smaller ones stay malloc'ed; fmd_table_free takes both."""
    import ctypes as C
    L = api.lib()

    def mapped_files():
        return [ln for ln in open("/proc/self/maps") if str(tmp_path) in ln]

    a = L.fmd_table_alloc(1 << 20)
    assert a and not mapped_files()
    C.memset(a, 7, 1 << 20)
    L.fmd_table_free(a)
    monkeypatch.setenv("FMD_TABLE_DIR", str(tmp_path))
    monkeypatch.setenv("FMD_TABLE_DIR_MIN", str(1 << 16))
    small, big = L.fmd_table_alloc(1000), L.fmd_table_alloc((1 << 20) + 123)
    assert small and big
    m = mapped_files()
    assert len(m) == 1 and "fmdtab." in m[0] and "(deleted)" in m[0] and not os.listdir(tmp_path)   # unlinked: nothing is left behind, whatever happens to the process
    buf = (C.c_uint8 * ((1 << 20) + 123)).from_address(big)
    buf[0] = 1; buf[(1 << 20) + 122] = 2
    assert buf[0] == 1 and buf[(1 << 20) + 122] == 2 and buf[5000] == 0
    L.fmd_table_free(big); L.fmd_table_free(small); L.fmd_table_free(None)
    assert not mapped_files()
    monkeypatch.setenv("FMD_TABLE_DIR", str(tmp_path / "missing"))                   # no such directory: anonymous memory, a warning, no failure
    c = L.fmd_table_alloc(1 << 20)
    assert c and not mapped_files()
    L.fmd_table_free(c)
