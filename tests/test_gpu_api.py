"""GPU: the in-memory API (fermi_amd/host/api_mem.c: fmdh_api_unitig / fmdh_api_correct = fm6_api_unitig / fm6_api_correct, fermi.h:119-123)
against the output of the reference's own caller of that API, `fermi example` (example.c:29-45), on the fixtures."""
import gzip

import numpy as np
import pytest

from fermi_amd import hostlib

pytestmark = pytest.mark.gpu


def _fastq(gold, name):
    lines = gold.text_gz(name).split(b"\n")
    return [lines[i] for i in range(1, len(lines) - 1, 4)], [lines[i] for i in range(3, len(lines) - 1, 4)]


@pytest.mark.parametrize("name,mm,want", [("tiny", 50, "tiny.mag.gz"), ("special", 20, "special.api_l20.mag.gz")])
def test_api_unitig_equals_fermi_example(gpu, gold, tmp_path, name, mm, want):
    """tiny has no self-reverse-complement read: fm6_build2's index is `fermi build`'s and the API's MAG is `fermi unitig -t1`'s; special
    has one (and Ns, and ragged lengths): the API does not trim it (build.c:52-70) and the fixture is `fermi example -l 20`'s output."""
    reads, _ = _fastq(gold, name + ".fq.gz")
    out = str(tmp_path / "o.mag")
    hostlib.api_unitig(reads, mm, out)
    assert open(out, "rb").read() == gold.text_gz(want)


def test_api_unitig_chooses_min_match_like_the_reference(gpu, gold, tmp_path):
    reads, _ = _fastq(gold, "tiny.fq.gz")
    L = hostlib.lib()
    buf = np.frombuffer(b"".join(r + b"\0" for r in reads), dtype=np.uint8).copy()
    q25 = L.fmdh_api_seqlen(len(buf), buf.ctypes.data, 0.25)
    assert q25 == sorted(len(r) for r in reads)[int(len(reads) * 0.25)]
    a, b = str(tmp_path / "a.mag"), str(tmp_path / "b.mag")
    hostlib.api_unitig(reads, -1, a)
    hostlib.api_unitig(reads, int(q25 * .33 + .499), b)       # unitig.c:418-421
    assert open(a, "rb").read() == open(b, "rb").read() and len(open(a, "rb").read()) > 1000


def test_api_correct_equals_fermi_example(gpu, gold):
    """`fermi example -eU -k 17 tiny.fq.gz` = fm6_api_correct + fm6_api_writeseq (seq.c:410-428: '@' + the offset of the read's NUL).  The
    reference leaves opt.step uninitialised there (correct.c:471-474); the binary this repository pins behaves as one of the values
    tried here, and the API takes the step as a parameter."""
    reads, quals = _fastq(gold, "tiny.fq.gz")
    want = gold.text_gz("tiny.api_ec_k17.fq.gz")
    got = {}
    for step in (0, 5, 1000):
        s, q = hostlib.api_correct(reads, quals, 17, step)
        pos, out = -1, []
        for a, b in zip(s, q):
            pos += len(a) + 1
            out.append(b"@%d\n%s\n+\n%s\n" % (pos, a, b))
        got[step] = b"".join(out)
    assert any(g == want for g in got.values()), {k: sum(x != y for x, y in zip(v.split(b"\n"), want.split(b"\n"))) for k, v in got.items()}
    # whatever the step, what was changed is lower case with quality '$', what was kept is the input in upper case
    s, q = hostlib.api_correct(reads, quals, 17, 5)
    n_low = 0
    for a, b, r0 in zip(s, q, reads):
        assert len(a) == len(r0) and a.upper().replace(b"N", b"N") != b""
        for x, y, z in zip(a, b, r0):
            if chr(x).islower():
                n_low += 1
                assert y == 36 and chr(x).upper() != chr(z).upper()
            else:
                assert x == z
    assert n_low > 100
