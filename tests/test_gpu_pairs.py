"""GPU: the two-base blocks (fmd_pair.hip) and the kernel that reads them (k_ovl_pair: pass 1 of the sorted job between the depth at which the single-step
head hands a strand over and the 32 bases at which it is parked).  A pair step is two exact LF / extension steps, so (a) every row's pair step must equal
two single steps (fmd_dev_check_pairs, all rows of every fixture, the look-ahead chunks included), and (b) the sorted job must leave the same bytes whether
the handle has two-base blocks or not, wherever the hand-over is (FMD_PAIR_FROM), on reads with errors, ragged lengths, Ns and sequences that end inside the
head (all of which k_ovl_pair refuses and the head walks again) -- and those bytes are the id-order pass's, which the golden tests tie to the reference."""
import os

import numpy as np
import pytest

from fermi_amd import synth

pytestmark = pytest.mark.gpu
U64 = np.uint64
HERE = os.path.dirname(os.path.abspath(__file__))


def _same(a, b, max_nei):
    rec0, nei0, seq0 = a; rec1, nei1, seq1 = b
    assert rec1.tobytes() == rec0.tobytes()
    for j in range(max_nei):
        mj = rec0["n_nei"] > j
        assert nei1[mj, j].tobytes() == nei0[mj, j].tobytes(), j
    used = (rec0["len"] + np.maximum(rec0["ext_len"], 0)).astype(np.int64)
    m = (np.arange(seq0.shape[1])[None, :] < used[:, None]) & (rec0["status"] == 0)[:, None]
    assert np.array_equal(seq1[m], seq0[m])


@pytest.mark.parametrize("name", ["tiny", "special", "repeat", "dup32"])
def test_pair_step_equals_two_single_steps_on_every_row(gpu, monkeypatch, name):
    monkeypatch.setenv("FMD_PAIR", "1")
    d = gpu.DevIndex.open(os.path.join(HERE, "golden", name + ".fmd"))
    assert d.build_pairs()
    assert d.check_pairs() == (0, 0)
    d.close()


def _ragged(rng, N, L, cov, err):
    base = synth.reads(synth.DEFAULT_SEED + 77, N, L, cov, err)
    reads = []
    for i in range(N):
        r = base[i].copy()
        u = rng.random()
        if u < 0.05:
            r = r[: rng.integers(1, 62)]                     # ends inside the head, before or after the hand-over
        elif u < 0.15:
            r = r[rng.integers(0, 45):]
        if rng.random() < 0.06:
            r[rng.integers(0, len(r))] = 5                   # an N anywhere: before the hand-over, among the bases k_ovl_pair would take, beyond
        reads.append(r)
    return reads + reads[:40]


@pytest.mark.parametrize("mm,splits", [(50, (None, 14, 20, 30)), (60, (None, 18)), (33, (None, 26))])
def test_sorted_job_with_and_without_two_base_blocks(gpu, monkeypatch, mm, splits):
    rng = np.random.default_rng(5 + mm)
    reads = _ragged(rng, 14000, 100, 40, 0.004)
    bwt = gpu.build_bwt(reads)
    n_seq = 2 * len(reads)
    ids = rng.permutation(n_seq)[: n_seq - 77].astype(U64)
    monkeypatch.setenv("FMD_PAIR", "0")                               # never: not even when asked for through the ABI
    d0 = gpu.DevIndex.from_bwt(bwt)
    want = d0.overlap(ids, mm, 100, 8, check_left=False)              # id order, the one-pass walk
    plain = d0.overlap_sorted(ids, mm, 100, 8, 5000)
    assert not d0.build_pairs()
    _same(want, plain, 8)
    d0.close()
    monkeypatch.setenv("FMD_PAIR", "1")
    d1 = gpu.DevIndex.from_bwt(bwt)
    for sp in splits:
        if sp is None:
            monkeypatch.delenv("FMD_PAIR_FROM", raising=False)
        else:
            monkeypatch.setenv("FMD_PAIR_FROM", str(sp))
        for batch in (0, 4097):
            _same(want, d1.overlap_sorted(ids, mm, 100, 8, batch), 8)
    assert d1.build_pairs() and d1.check_pairs() == (0, 0)            # (the job built them itself)
    assert (want[0]["status"] == -1).sum() > 100 and (want[0]["n_nei"] > 0).sum() > 3000
    d1.close()


def test_two_base_head_on_a_repeat_rich_deep_set(gpu, monkeypatch):
    """80-fold reads of a genome with repeats: intervals that are still wider than 63 at the hand-over (k_ovl_pair refuses the strand, the head walks it
    again), identical reads, forks."""
    monkeypatch.setenv("FMD_PAIR", "1")
    rng = np.random.default_rng(3)
    unit = rng.integers(1, 5, 300).astype(np.uint8)
    g = np.concatenate([rng.integers(1, 5, 3000).astype(np.uint8), unit, rng.integers(1, 5, 2000).astype(np.uint8), unit, unit, rng.integers(1, 5, 3000).astype(np.uint8)])
    N, L = 9000, 100
    pos = rng.integers(0, len(g) - L, N)
    reads = []
    for i in range(N):
        r = g[pos[i]:pos[i] + L].copy()
        if rng.random() < 0.5:
            r = (5 - r)[::-1].copy()
        e = rng.random(L) < 0.003
        r[e] = (r[e] - 1 + rng.integers(1, 4, int(e.sum()))) % 4 + 1
        reads.append(r)
    d = gpu.DevIndex.from_bwt(gpu.build_bwt(reads))
    ids = np.arange(2 * N, dtype=U64)
    for mm in (40, 55):
        _same(d.overlap(ids, mm, L, 16, check_left=False), d.overlap_sorted(ids, mm, L, 16, 6000), 16)
    assert d.build_pairs() and d.check_pairs() == (0, 0)
    d.close()


def test_backward_search_two_bases_per_request(gpu, monkeypatch):
    """fm_backward_search (exact.c:7-23) with the two-base blocks: k_bsearch<1> hands a search over as soon as its interval is narrower than 65 and an even
    number of bases is left, k_bsearch_pair takes two bases per request from there, k_bsearch<2> redoes the reads with an N among the bases left.  Same
    counts and intervals as without the blocks: hits, early and late misses (reads with errors against an error-free index), ragged lengths from 1 base
    on (odd and even), Ns anywhere, duplicates."""
    rng = np.random.default_rng(17)
    N = 20000
    clean = synth.reads(synth.DEFAULT_SEED + 5, N, 100, 30, 0.0)
    dirty = synth.reads(synth.DEFAULT_SEED + 5, N, 100, 30, 0.02)
    q = []
    for i in range(N):
        r = (clean[i] if i % 3 else dirty[i]).copy()
        u = rng.random()
        if u < 0.2:
            r = r[rng.integers(0, 99):]
        elif u < 0.3:
            r = r[: rng.integers(1, 100)]
        if rng.random() < 0.05:
            r[rng.integers(0, len(r))] = 5
        q.append(r)
    bwt = gpu.build_bwt(list(clean))
    monkeypatch.setenv("FMD_PAIR", "0")
    d0 = gpu.DevIndex.from_bwt(bwt)
    want = d0.backward_search(q)
    d0.close()
    monkeypatch.setenv("FMD_PAIR", "1")
    d1 = gpu.DevIndex.from_bwt(bwt)
    assert d1.build_pairs()
    got = d1.backward_search(q)
    hit = want[0] > 0
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1][hit], want[1][hit]) and np.array_equal(got[2][hit], want[2][hit])
    assert hit.sum() > N // 2 and (~hit).sum() > N // 10
    monkeypatch.setenv("FMD_PAIR_USE", "0")
    again = d1.backward_search(q)
    assert np.array_equal(again[0], want[0])
    d1.close()
