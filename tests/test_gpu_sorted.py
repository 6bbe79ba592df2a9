"""GPU: the sorted overlap job (fmd_ovlp_sorted_dev: every strand 32 bases in, minimizer sort, the rest in that order, batch by
batch) leaves the bytes fmd_ovlp_dev leaves in id order -- records incl. lfork, neighbours, sequences + appended bases -- and
what the oracle computes, whatever the batch size, on reads with errors (forks: the general group kernels and the lane-per-strand
kernel write through the slot -> row map too), ragged lengths, sequences that end inside the first pass, and Ns."""
import numpy as np
import pytest

import orcbind
from fermi_amd import synth

pytestmark = pytest.mark.gpu
U64 = np.uint64


def _same(a, b, max_nei):
    rec0, nei0, seq0 = a; rec1, nei1, seq1 = b
    assert rec1.tobytes() == rec0.tobytes()
    for j in range(max_nei):
        mj = rec0["n_nei"] > j
        assert nei1[mj, j].tobytes() == nei0[mj, j].tobytes(), j
    used = (rec0["len"] + np.maximum(rec0["ext_len"], 0)).astype(np.int64)
    m = (np.arange(seq0.shape[1])[None, :] < used[:, None]) & (rec0["status"] == 0)[:, None]
    assert np.array_equal(seq1[m], seq0[m])


@pytest.mark.parametrize("L,cov,mm,err,N,batch", [(100, 30, 50, 0.0, 30000, 0), (100, 30, 50, 0.0, 30000, 7777), (100, 30, 50, 0.01, 30000, 25001),
                                                  (100, 60, 40, 0.003, 10000, 4096), (151, 12, 33, 0.02, 8000, 5000), (100, 80, 45, 0.005, 6000, 12000)])
def test_sorted_job_equals_id_order_and_oracle(gpu, oracle_lib, monkeypatch, L, cov, mm, err, N, batch):
    reads = synth.reads(synth.DEFAULT_SEED + 3 * L + cov, N, L, cov, err)
    bwt = gpu.build_bwt(reads)
    d = gpu.DevIndex.from_bwt(bwt)
    ids = np.arange(2 * N - 5, dtype=U64)
    want = d.overlap(ids, mm, L, 8, check_left=False)              # fmd_ovlp_batch: chunks in id order, the one-pass walk
    got = d.overlap_sorted(ids, mm, L, 8, batch)
    _same(want, got, 8)
    assert (got[0]["n_nei"] > 0).sum() > N // 4
    monkeypatch.setenv("FMD_OVLP_SORT", "0")                        # the same entry point with the sort switched off: id order inside
    _same(want, d.overlap_sorted(ids, mm, L, 8, batch), 8)
    monkeypatch.delenv("FMD_OVLP_SORT")
    monkeypatch.setenv("FMD_OVLP_FAST", "0")                        # every strand through the general group kernels
    _same(want, d.overlap_sorted(ids, mm, L, 8, batch), 8)
    monkeypatch.delenv("FMD_OVLP_FAST")
    o = orcbind.OrcIndex(bwt=bwt)
    sub = np.sort(np.random.default_rng(L + cov).choice(len(ids), 3000, replace=False)).astype(U64)
    wrec, wnei, _ = o.overlap_batch(sub, mm, L, 8, 4, check_left=False)
    si = sub.astype(np.int64)
    ok = (got[0]["flags"][si] & gpu.OVLP_F_OVERFLOW) == 0
    assert ok.sum() > 2900
    for f in ("rank", "k", "len", "status", "n_ovlp", "rbeg", "ext_len", "n_nei"):
        assert np.array_equal(got[0][f][si][ok], wrec[f][ok]), f
    for j in range(8):
        mj = ok & (wrec["n_nei"] > j)
        assert got[1][si][mj, j].tobytes() == wnei[mj, j].tobytes(), j
    d.close(); o.close()


def test_sorted_job_ragged_short_and_ambiguous_reads(gpu, oracle_lib):
    """Sequences that end inside the first pass (shorter than 32 bases: their records are written there), sequences just past it,
    Ns inside and outside the first 32 bases (no minimizer over an N; an N among the last 12 bases: no tail-table entry),
    duplicates and an arbitrary subset of ids in arbitrary order."""
    rng = np.random.default_rng(99)
    N = 12000
    base = synth.reads(synth.DEFAULT_SEED + 41, N, 100, 40, 0.004)
    reads = []
    for i in range(N):
        r = base[i].copy()
        u = rng.random()
        if u < 0.06:
            r = r[: rng.integers(1, 40)]                     # ends inside (or just behind) the head
        elif u < 0.16:
            r = r[rng.integers(0, 45):]                      # ragged
        if rng.random() < 0.05:
            r[rng.integers(0, len(r))] = 5                   # an N anywhere
        if rng.random() < 0.02 and len(r) > 8:
            r[len(r) - 1 - rng.integers(0, 8)] = 5           # an N among the last bases
        reads.append(r)
    reads += reads[:50]                                      # duplicates
    bwt = gpu.build_bwt(reads)
    d = gpu.DevIndex.from_bwt(bwt)
    n_seq = 2 * len(reads)
    ids = rng.permutation(n_seq)[: n_seq - 123].astype(U64)
    for mm, batch in ((50, 0), (32, 3001), (60, 9000)):
        want = d.overlap(ids, mm, 100, 8, check_left=False)
        got = d.overlap_sorted(ids, mm, 100, 8, batch)
        _same(want, got, 8)
    assert (want[0]["status"] == -1).sum() > 100 and (want[0]["len"] < 32).sum() > 100
    o = orcbind.OrcIndex(bwt=bwt)
    wrec, wnei, _ = o.overlap_batch(ids[:4000], 60, 100, 8, 4, check_left=False)
    ok = (got[0]["flags"][:4000] & gpu.OVLP_F_OVERFLOW) == 0
    for f in ("rank", "k", "len", "status", "n_ovlp", "rbeg", "ext_len", "n_nei"):
        assert np.array_equal(got[0][f][:4000][ok], wrec[f][ok]), f
    d.close(); o.close()


def test_sorted_job_below_the_split_takes_id_order(gpu):
    """min_match < 32: candidates could be pushed inside the first pass, so the job runs in id order (same bytes by construction);
    and the work-area contract: a work area too small for the job is an argument error, not a crash."""
    N = 5000
    reads = synth.reads(synth.DEFAULT_SEED + 8, N, 80, 30, 0.0)
    d = gpu.DevIndex.from_bwt(gpu.build_bwt(reads))
    ids = np.arange(2 * N, dtype=U64)
    _same(d.overlap(ids, 25, 80, 8, check_left=False), d.overlap_sorted(ids, 25, 80, 8, 3000), 8)
    L = gpu.lib()
    assert L.fmd_ovlp_sorted_work_bytes(10**6, 10**5, 100, 50) > L.fmd_ovlp_work_bytes(10**5, 100, 50) + 64 * 10**6
    assert L.fmd_ovlp_sorted_dev(d.h, None, 1000, 1, 50, 100, 4, 1, 1, 1, 200, 1, 4096, 0) == gpu.FMD_E_ARG
    d.close()


@pytest.mark.parametrize("name,mm", [("tiny", 50), ("repeat", 20)])
@pytest.mark.parametrize("devs", [(0,), (0, 0, 0)])
def test_unitig_from_a_sorted_table_equals_fermi_unitig_t1(gpu, gold, tmp_path, monkeypatch, name, mm, devs):
    """The product's table (fmd_ovlp_packed_table / fmd_ovlp_packed_batch) takes large tables from one sorted job and packs the
    chunks from it in id order; forced on the small fixtures here (FMD_PACKED_SORT_MIN), one GPU and three replicas (strided ids):
    the MAG is `fermi unitig -t1`'s, byte for byte.  (-l20 is below the split: the job runs in id order inside.)"""
    from fermi_amd import hostlib
    monkeypatch.setenv("FMD_PACKED_SORT_MIN", "1")
    out = str(tmp_path / "o.mag")
    hostlib.unitig(gold.path(name + ".fmd"), mm, out, devices=devs)
    assert open(out, "rb").read() == gold.text_gz(name + ".mag.gz")


def test_packed_rows_from_a_sorted_job_equal_chunked_rows(gpu, monkeypatch):
    """fmd_ovlp_packed_batch with explicit ids, with and without the sorted job behind it, small chunks: same records, offsets, bytes."""
    import ctypes as C
    N = 20000
    reads = synth.reads(synth.DEFAULT_SEED + 12, N, 100, 30, 0.005)
    d = gpu.DevIndex.from_bwt(gpu.build_bwt(reads))
    L = gpu.lib()
    ids = np.random.default_rng(4).permutation(2 * N)[: 2 * N - 77].astype(U64)
    n, shift = len(ids), 13
    nch = (n + (1 << shift) - 1) >> shift
    res = []
    for force in (False, True):
        if force:
            monkeypatch.setenv("FMD_PACKED_SORT_MIN", "1")
        rec = np.zeros(n, dtype=gpu.OVLP_DT); off = np.zeros(n, dtype=U64)
        chunks = (C.c_void_p * nch)()
        gpu.check(L.fmd_ovlp_packed_batch(d.h, ids.ctypes.data, 0, 1, n, 50, 100, 4, 1, rec.ctypes.data, off.ctypes.data, shift, chunks))
        var = []
        for c in range(nch):
            lo, hi = c << shift, min(n, (c + 1) << shift)
            last = hi - 1
            var.append(C.string_at(chunks[c], int(off[last])) if hi > lo else b"")   # (up to the start of the chunk's last row: its length follows from its record)
        L.fmd_ovlp_packed_free(chunks, nch)
        res.append((rec.tobytes(), off.tobytes(), var))
    assert res[0][0] == res[1][0] and res[0][1] == res[1][1] and res[0][2] == res[1][2]
    d.close()


@pytest.mark.parametrize("devs", [(0,), (0, 0)])
def test_unitig_with_the_table_in_file_pages(gpu, gold, tmp_path, monkeypatch, devs):
    """FMD_TABLE_DIR: chunks, records, offsets, row map and links of `unitig`'s table are pages of unlinked files (fmd_table_alloc), the
    copies from the GPU land in them unpinned where the runtime will not register file pages; the MAG is `fermi unitig -t1`'s."""
    from fermi_amd import hostlib
    d = tmp_path / "pages"; d.mkdir()
    monkeypatch.setenv("FMD_TABLE_DIR", str(d))
    monkeypatch.setenv("FMD_TABLE_DIR_MIN", "1")
    out = str(tmp_path / "o.mag")
    hostlib.unitig(gold.path("tiny.fmd"), 50, out, devices=devs)
    assert open(out, "rb").read() == gold.text_gz("tiny.mag.gz")
    assert not list(d.iterdir())
