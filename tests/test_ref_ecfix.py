"""The CPU leg of bench.py's ec_fix line, pinned where /root/reference was compiled (oracle/_ref): the reference's own static ec_fix through
oracle/ref_ec_harness.c:refec_fix, its tables filled from the golden solid table of tiny.fmd, must print `fermi correct -t1`'s FASTQ; and bench.py's
numpy form of the marking rule (correct.c:247-252) must agree with it.  No GPU."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import bench  # noqa: E402
from benchlegs import ecfix as ecfix_leg  # noqa: E402
import orcbind  # noqa: E402
from test_oracle_golden import _fastq_records  # noqa: E402


def _fixed_len(gold, L=100):
    recs = [r for r in _fastq_records(gold.text_gz("tiny.fq.gz"))]
    ids = [i for i, r in enumerate(recs) if len(r[1]) == L]
    nt6 = np.stack([bench.NT6_OF_ASCII[np.frombuffer(recs[i][1], dtype=np.uint8)] for i in ids])
    q = np.stack([np.frombuffer(recs[i][2], dtype=np.uint8) for i in ids])
    return ids, nt6, q


def _sorted_trip(v):
    o = np.argsort(v["w17_o3_bucket"], kind="stable")
    return (np.ascontiguousarray(v["w17_o3_bucket"][o], dtype=np.uint32), np.ascontiguousarray(v["w17_o3_key"][o], dtype=np.uint32), np.ascontiguousarray(v["w17_o3_val"][o], dtype=np.uint8))


def _kept_records(text):
    return {int(r[0].lstrip(b"@").split(b"_")[0]): (r[0].lstrip(b"@"), r[1], r[2]) for r in _fastq_records(text)}


@pytest.mark.parametrize("threads", [1, 3])
def test_refec_fix_prints_fermi_correct(gold, oracle_lib, threads, monkeypatch):
    if bench.ref_ec_lib() is None:
        pytest.skip("oracle/_ref/libref_ec.so not built here")
    monkeypatch.setattr(ecfix_leg, "usable_cpus", lambda: threads)
    ids, nt6, q = _fixed_len(gold)
    (txt, q2, info), _, _, lpr, kind, cores = bench.cpu_ecfix(17, 2, 5, _sorted_trip(gold.npz("tiny_solid.npz")), nt6, q)
    assert kind == "reference" and cores == threads and lpr > 20
    want = _kept_records(gold.text_gz("tiny.ec.fq.gz"))
    n_kept = 0
    for k, i in enumerate(ids):
        bad = info[k] >> 16 & 1
        assert (i in want) == (not bad), i                       # the filter of correct.c:412 (keep_bad = 0)
        if not bad:
            name, s, ql = want[i]
            assert name == b"%d_%d_%d" % (i, info[k] & 0xffff, info[k] >> 18) and s == txt[k].tobytes() and ql == q2[k].tobytes(), i
            n_kept += 1
    assert n_kept == len(want) and n_kept >= 2000   # (every record of the golden output has been compared)


def test_marking_rule_in_numpy_equals_the_reference(gold, oracle_lib):
    """the oracle's ec_fix (nt6 bases, qualities and info BEFORE the marking: the contract of fmd_ecfix_dev) + bench.mark_corrected == refec_fix (after it)"""
    if bench.ref_ec_lib() is None:
        pytest.skip("oracle/_ref/libref_ec.so not built here")
    ids, nt6, q = _fixed_len(gold)
    rng = np.random.default_rng(3)
    extra = nt6[:400].copy()                                      # reads the filter rejects: other genomes, reads with a third of their bases changed, Ns
    extra[:100] = rng.integers(1, 5, size=(100, nt6.shape[1]))
    for r in extra[100:300]:
        r[rng.choice(len(r), 35, replace=False)] = rng.integers(1, 5, size=35)
    extra[300:, ::17] = 5
    nt6 = np.concatenate([nt6, extra]); q = np.concatenate([q, q[:400]])
    v = gold.npz("tiny_solid.npz")
    (txt, q2, info), _, _, _, _, _ = bench.cpu_ecfix(17, 2, 5, _sorted_trip(v), nt6, q)
    s, qq, off, inf = orcbind.ec_fix(17, v["w17_o3_bucket"], v["w17_o3_key"], v["w17_o3_val"], list(nt6), list(q))
    m_txt, m_q, m_inf = bench.mark_corrected(nt6, s.reshape(nt6.shape), qq.reshape(nt6.shape), inf)
    assert np.array_equal(m_txt, txt) and np.array_equal(m_q, q2) and np.array_equal(m_inf, info)
    assert (m_txt >= ord("a")).sum() > 1000 and (m_inf >> 16 & 1).sum() > 100
