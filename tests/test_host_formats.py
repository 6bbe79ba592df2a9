"""CPU: the product's own .fmd writers (fermi_amd/host/rld_writer.c) are byte-identical to the
files `fermi build` / `fermi ropebwt` wrote (golden), incl. 32-bit block headers."""
import subprocess
import os

import numpy as np
import pytest

import orcbind
from fermi_amd import hostlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module", autouse=True)
def _build_host():
    subprocess.check_call(["make", "-s", "-C", ROOT, "host"])


def test_rle6_to_rld_is_fermi_build_output(gold, tmp_path):
    runs = np.frombuffer(open(gold.path("tiny.rle.fmd"), "rb").read()[4:], dtype=np.uint8)
    out = str(tmp_path / "t.fmd")
    hostlib.write_rld_from_rle6(runs, out)
    assert open(out, "rb").read() == open(gold.path("tiny.fmd"), "rb").read()
    out2 = str(tmp_path / "t.rle.fmd")
    hostlib.write_rle6(runs, out2)
    assert open(out2, "rb").read() == open(gold.path("tiny.rle.fmd"), "rb").read()


@pytest.mark.parametrize("name", ["tiny", "special", "dup32"])
def test_bwt_to_rld_roundtrip(oracle_lib, gold, tmp_path, name):
    o = orcbind.OrcIndex(gold.path(name + ".fmd"))
    out = str(tmp_path / "b.fmd")
    hostlib.write_rld_from_bwt(o.decode_all(), out)
    assert open(out, "rb").read() == open(gold.path(name + ".fmd"), "rb").read()
    o.close()


def test_trim_palindrome():
    nt = {"A": 1, "C": 2, "G": 3, "T": 4, "N": 5}
    f = lambda s: hostlib.trim_palindrome(np.array([nt[c] for c in s], dtype=np.uint8))
    assert f("AACCGGTT") == 7          # even length, own reverse complement (cmd.c:457-463)
    assert f("AACCGGTA") == 8
    assert f("ACGTA") == 5             # odd length never trimmed
    assert f("ANNT") == 4


def _packed_shards(rec, nei, seq, n_shards):
    """The oracle's fixed-stride table as the packed shards the GPUs produce: id i = row i // N of shard i % N."""
    import packref
    return [packref.pack_rows(rec[g::n_shards], nei[g::n_shards], seq[g::n_shards], nei.shape[1]) for g in range(n_shards)]


@pytest.mark.parametrize("name,mm", [("tiny", 50), ("repeat", 20), ("special", 20)])
@pytest.mark.parametrize("n_shards", [1, 2, 3])
def test_unitig_walk_reproduces_fermi_unitig_t1(oracle_lib, gold, tmp_path, name, mm, n_shards):
    """Host walk (fermi_amd/host/unitig_walk.c) over the packed per-read table == `fermi unitig -t1` output, byte for
    byte, however many shards (GPUs) the table was computed in.  The table here comes from the oracle (CPU); the GPU
    tests feed the same walk from fmd_ovlp_packed_batch."""
    o = orcbind.OrcIndex(gold.path(name + ".fmd"))
    n_seq = int(o.mcnt[1])
    rec, nei, seq = o.overlap_batch(np.arange(n_seq, dtype=np.uint64), mm, max_len=100, max_nei=8, n_threads=4)
    out = str(tmp_path / "o.mag")
    hostlib.unitig_walk(_packed_shards(rec, nei, seq, n_shards), n_seq, mm, out, max_nei=8, seq_stride=seq.shape[1])
    got = open(out, "rb").read()
    want = gold.text_gz(name + ".mag.gz")
    assert got == want
    o.close()


@pytest.mark.parametrize("name,mm", [("tiny", 50), ("repeat", 20), ("special", 20)])
@pytest.mark.parametrize("weaken", [0, 3])
def test_unitig_walk_with_check_left_decided_by_the_link_pass(oracle_lib, gold, tmp_path, name, mm, weaken):
    """The product path: check_left_simple is NOT run per row (rec.reserved = 2); fmdh_ovlp_table_link decides every edge
    from the lfork of the neighbour's reverse strand (include/fmd_hip.h) on several threads, hands back the rows it cannot
    decide, and the walk steps through the link array.  With the oracle's exact lfork nothing is left undecided; with
    every `weaken`-th lfork blanked the rest goes to the exact answer.  MAG == `fermi unitig -t1` either way."""
    o = orcbind.OrcIndex(gold.path(name + ".fmd"))
    n_seq = int(o.mcnt[1])
    rec, nei, seq = o.overlap_batch(np.arange(n_seq, dtype=np.uint64), mm, max_len=100, max_nei=8, n_threads=4)
    exact = rec["reserved"].copy()
    rec["reserved"] = 2
    if weaken:
        rec["lfork"][::weaken] = 0
    out = str(tmp_path / "o.mag")
    und = hostlib.unitig_walk(_packed_shards(rec, nei, seq, 2), n_seq, mm, out, max_nei=8, seq_stride=seq.shape[1], link=3,
                              resolve=lambda ids: exact[ids.astype(np.int64)])
    assert (len(und) == 0) if not weaken else (len(und) > 0)
    assert open(out, "rb").read() == gold.text_gz(name + ".mag.gz")
    o.close()


@pytest.mark.parametrize("n_shards", [1, 3])
def test_unitig_walk_prefetch_hints_change_nothing(oracle_lib, gold, tmp_path, monkeypatch, n_shards):
    """The walk's skip list over the links (unitig_walk.c: jump[] by pointer doubling, prefetches 8 rows ahead) only issues hints:
    the MAG is the reference's with it (default, linked table) and without it (FMD_WALK_NO_JUMP=1), one shard or three."""
    o = orcbind.OrcIndex(gold.path("tiny.fmd"))
    n_seq = int(o.mcnt[1])
    rec, nei, seq = o.overlap_batch(np.arange(n_seq, dtype=np.uint64), 50, max_len=100, max_nei=8, n_threads=4)
    exact = rec["reserved"].copy()
    rec["reserved"] = 2
    outs = []
    monkeypatch.setenv("FMD_WALK_LONG", "1")
    monkeypatch.setenv("FMD_FAR_CHECK", "1")      # the skip list by pointer doubling (one temporary array) == by following eight links per row (none: large tables)
    for no_jump in (False, True, "chase"):
        if no_jump is True:
            monkeypatch.setenv("FMD_WALK_NO_JUMP", "1")
        if no_jump == "chase":
            monkeypatch.delenv("FMD_WALK_NO_JUMP"); monkeypatch.setenv("FMD_FAR_CHASE", "1")
        out = str(tmp_path / ("o%s.mag" % no_jump))
        hostlib.unitig_walk(_packed_shards(rec.copy(), nei, seq, n_shards), n_seq, 50, out, max_nei=8, seq_stride=seq.shape[1], link=2,
                            resolve=lambda ids: exact[ids.astype(np.int64)])
        outs.append(open(out, "rb").read())
    assert outs[0] == outs[1] == outs[2] == gold.text_gz("tiny.mag.gz")
    o.close()


@pytest.mark.parametrize("name,mm", [("tiny", 50), ("repeat", 20), ("special", 20), ("circle", 40)])
@pytest.mark.parametrize("threads", [1, 4])
def test_one_line_per_plain_step_changes_nothing(oracle_lib, gold, tmp_path, monkeypatch, name, mm, threads):
    """hop[] (unitig_walk.c: the 29 bytes a plain step of the walk needs, one 32-byte entry per row, built from the linked table by all host threads)
    against the general code that reads the record, the link, the packed row and the reverse strand's record (FMD_WALK_NO_HOP=1): the same MAG, the
    reference's, from the sequential loop and from the speculative chunks -- on the fixtures with forks, contained reads, bases other than A/C/G/T
    (rows that are not plain steps) and unitigs that close on themselves."""
    monkeypatch.setenv("FMD_WALK_THREADS", str(threads))
    monkeypatch.setenv("FMD_WALK_CHUNK", "16")
    monkeypatch.setenv("FMD_WALK_LONG", "1")   # (the walk builds hop[] and the skip list where a sample of the links shows long walks: here, always)
    monkeypatch.setenv("FMD_FAR_CHECK", "1")   # (and checks its two constructions of the skip list against each other: loops, forks, ends)
    o = orcbind.OrcIndex(gold.path(name + ".fmd"))
    n_seq = int(o.mcnt[1])
    rec, nei, seq = o.overlap_batch(np.arange(n_seq, dtype=np.uint64), mm, max_len=100, max_nei=8, n_threads=4)
    outs = []
    for no_hop in (False, True):
        if no_hop:
            monkeypatch.setenv("FMD_WALK_NO_HOP", "1")
        out = str(tmp_path / ("o%d.mag" % no_hop))
        hostlib.unitig_walk(_packed_shards(rec.copy(), nei, seq, 2), n_seq, mm, out, max_nei=8, seq_stride=seq.shape[1], link=3)
        outs.append(open(out, "rb").read())
    assert outs[0] == outs[1] == gold.text_gz(name + ".mag.gz")
    o.close()


def _random_bwt(rng, n, mean_run, long_runs):
    """nt6 symbols with geometric run lengths (mean `mean_run`) and a few very long runs (>= 2^15: the 32-bit block headers)"""
    out, tot = [], 0
    while tot < n:
        k = int(rng.integers(2000, 8000))
        lens = rng.geometric(1.0 / mean_run, k)
        syms = rng.integers(0, 6, k).astype(np.uint8)
        out.append(np.repeat(syms, lens)); tot += int(lens.sum())
        if long_runs and rng.random() < 0.3:
            L = int(rng.integers(1 << 15, 1 << 18))
            out.append(np.full(L, int(rng.integers(0, 6)), dtype=np.uint8)); tot += L
    return np.concatenate(out)[:n]


def _to_rle6(bwt):
    """the byte stream the GPU run-length pass emits: len << 3 | sym, len <= 31 (long runs in pieces)"""
    edge = np.flatnonzero(np.diff(bwt)) + 1
    start = np.concatenate(([0], edge)); ln = np.diff(np.concatenate((start, [len(bwt)])))
    reps = (ln + 30) // 31
    sym = np.repeat(bwt[start], reps)
    piece = np.full(int(reps.sum()), 31, dtype=np.int64)
    last = np.cumsum(reps) - 1
    piece[last] = ln - 31 * (reps - 1)
    return (piece.astype(np.uint8) << 3 | sym).astype(np.uint8)


@pytest.mark.parametrize("mean_run,long_runs,chunk_words", [(1.3, False, 0), (6, True, 0), (40, True, 0), (2, False, 1024), (9, True, 4096), (3, True, 64)])
def test_parallel_rld_encoder_writes_the_bytes_of_the_sequential_one(tmp_path, monkeypatch, mean_run, long_runs, chunk_words):
    """rld_writer.c: speculative slices + stitch (several host threads) against the one-thread encoder, from a byte BWT and from an
    RLE\\6 stream; short and long runs (16- and 32-bit block headers); with the chunk length of the format shrunk so that the
    shortened last block of a chunk (rld.h:66) falls inside every slice (both encoders then write the same non-standard file)."""
    rng = np.random.default_rng(int(mean_run * 10) + chunk_words)
    bwt = _random_bwt(rng, 3_000_000, mean_run, long_runs)
    rle = _to_rle6(bwt)
    if chunk_words:
        monkeypatch.setenv("FMD_RLD_TEST_HOOKS", "1")
        monkeypatch.setenv("FMD_RLD_TEST_CHUNK_WORDS", str(chunk_words))
    want = {}
    for threads in (1, 3, 16, 61):
        monkeypatch.setenv("FMD_RLD_THREADS", str(threads))
        for kind, arr, fn in (("bwt", bwt, hostlib.write_rld_from_bwt), ("rle6", rle, hostlib.write_rld_from_rle6)):
            path = str(tmp_path / ("%s_%d.fmd" % (kind, threads)))
            fn(arr, path)
            got = open(path, "rb").read()
            os.remove(path)
            if threads == 1:
                want[kind] = got
            else:
                assert got == want[kind], (kind, threads)
    assert want["bwt"] == want["rle6"]


def test_correct_kmer_rule():
    assert hostlib.lib().fmdh_correct_kmer(404000) == 17       # correct.c:313-318


@pytest.mark.parametrize("name,mm", [("tiny", 50), ("repeat", 20), ("special", 20)])
def test_unitig_walk_with_rank_file(oracle_lib, gold, tmp_path, name, mm):
    """`unitig -r rank`: the used bitmap is indexed through the seqsort map (unitig.c:22-29, 282)."""
    o = orcbind.OrcIndex(gold.path(name + ".fmd"))
    n_seq = int(o.mcnt[1])
    rec, nei, seq = o.overlap_batch(np.arange(n_seq, dtype=np.uint64), mm, max_len=100, max_nei=8, n_threads=4)
    sm = np.fromfile(gold.path(name + ".rank"), dtype=np.uint64)
    assert len(sm) == n_seq
    out = str(tmp_path / "o.mag")
    hostlib.unitig_walk(_packed_shards(rec, nei, seq, 2), n_seq, mm, out, sorted_map=sm, max_nei=8, seq_stride=seq.shape[1])
    assert open(out, "rb").read() == gold.text_gz(name + ".r.mag.gz")
    o.close()


def _read_contigs(path):
    """(name, comment, nt6) per record, with kseq's header split (name = up to the first white space)."""
    import gzip
    lines = gzip.open(path, "rt").read().split("\n")
    tab = np.full(256, 5, dtype=np.uint8)
    for i, c in enumerate("$ACGTN"):
        tab[ord(c)] = i; tab[ord(c.lower())] = i
    out = []
    for h, s in zip(lines[0::4], lines[1::4]):
        if not h:
            continue
        h = h[1:]
        k = next((i for i, c in enumerate(h) if c.isspace()), len(h))
        out.append((h[:k], h[k + 1:] if k < len(h) and h[k + 1:] else None, tab[np.frombuffer(s.encode(), dtype=np.uint8)]))
    return out


@pytest.mark.parametrize("mode,kw", [("u", dict()), ("p", dict(skip=20, max_dist=600, rank=True)),
                                     ("c", dict(skip=20, max_dist=600, min_pcv=2, rank=True)), ("d", dict(max_dist=310, min_pcv=1, rank=True))])
def test_remap_host_part_reproduces_fermi_remap(oracle_lib, gold, tmp_path, mode, kw):
    """paircov + pair table + mask_pcv + printers (fermi_amd/host/remap_cmd.c) over the SMEM chain of
    each contig == `fermi remap -t1` bytes, incl. the UR:Z order (bucket order of the pair table, which
    lives across contigs) and the insert-size line on stderr."""
    import json
    o = orcbind.OrcIndex(gold.path("pairs.fmd"))
    contigs = _read_contigs(gold.path("pairs_contigs.fq.gz"))
    mems = [o.smem(c[2], 0) for c in contigs]
    kw = dict(kw)
    sm = np.fromfile(gold.path("pairs.rank"), dtype=np.uint64) if kw.pop("rank", False) else None
    out, err = str(tmp_path / "o"), str(tmp_path / "e")
    hostlib.remap_contigs(contigs, mems, int(o.mcnt[1]), out, err, sorted_map=sm, **kw)
    o.close()
    assert open(out, "rb").read() == gold.text_gz("pairs.remap_%s.gz" % mode)
    assert open(err).read().strip() == json.load(open(gold.path("pairs.remap_stderr.json")))[mode][0]


def test_remap_pair_table_restarts_like_the_reference_batches(oracle_lib, gold, tmp_path, monkeypatch):
    """The reference hands contigs to paircov_all in batches of >= 2^28 bases and every batch starts with a new pair
    table (smem.c:237, :380).  The product keeps one table across its own GPU batches and must start it afresh at the
    same places: forced here after every contig (FMD_REMAP_TABLE_BASES=1).  The table is empty between contigs, so only
    the ORDER of the UR:Z entries may depend on its history: everything else, and the entries as a set, stay golden."""
    o = orcbind.OrcIndex(gold.path("pairs.fmd"))
    contigs = _read_contigs(gold.path("pairs_contigs.fq.gz"))
    mems = [o.smem(c[2], 0) for c in contigs]
    sm = np.fromfile(gold.path("pairs.rank"), dtype=np.uint64)
    monkeypatch.setenv("FMD_REMAP_TABLE_BASES", "1")
    out, err = str(tmp_path / "o.txt"), str(tmp_path / "e.txt")
    hostlib.remap_contigs(contigs, mems, int(o.mcnt[1]), out, err, sorted_map=sm, skip=20, max_dist=600)
    def canon(blob):
        res = []
        for line in blob.split(b"\n"):
            if b"\tUR:Z:" in line:
                head, tag = line.split(b"\tUR:Z:")
                line = head + b"\tUR:Z:" + b";".join(sorted(tag.rstrip(b";").split(b";")))
            res.append(line)
        return res
    got = open(out, "rb").read()
    assert b"UR:Z:" in got and canon(got) == canon(gold.text_gz("pairs.remap_p.gz"))
    o.close()


@pytest.mark.parametrize("name,mm", [("tiny", 50), ("repeat", 20), ("special", 20)])
@pytest.mark.parametrize("threads,chunk", [(1, 0), (3, 16), (16, 64), (8, 2048), (4, -32)])
def test_parallel_walk_is_the_t1_walk(oracle_lib, gold, tmp_path, monkeypatch, name, mm, threads, chunk):
    """fmdh_unitig_walk with N threads (speculative chunks of seeds, committed in seed order: unitig_walk.c) writes the bytes of
    `fermi unitig -t1` whatever the number of threads and the chunk size -- chunks of 16 seeds make the fixtures hundreds of windows in
    which walks of one window do meet (the repeat fixture: forks, bend marks, reads used by an earlier seed of the same window)."""
    from fermi_amd import hostlib
    monkeypatch.setenv("FMD_WALK_THREADS", str(threads))
    monkeypatch.setenv("FMD_TIMING", "1")
    if chunk:
        monkeypatch.setenv("FMD_WALK_CHUNK", str(abs(chunk)))
    if chunk < 0:   # walks of more than three reads are "too long to speculate on": the rest of their chunk waits for the commit
        monkeypatch.setenv("FMD_WALK_SPEC_STEPS", "3")
    o = orcbind.OrcIndex(gold.path(name + ".fmd"))
    n_seq = int(o.mcnt[1])
    rec, nei, seq = o.overlap_batch(np.arange(n_seq, dtype=np.uint64), mm, max_len=100, max_nei=8, n_threads=4)
    o.close()
    out = str(tmp_path / "o.mag")
    hostlib.unitig_walk(_packed_shards(rec, nei, seq, 2), n_seq, mm, out, max_nei=8, seq_stride=seq.shape[1])
    assert open(out, "rb").read() == gold.text_gz(name + ".mag.gz")


def test_parallel_fastq_reader_gives_the_serial_reader_s_records(gold, tmp_path):
    """seqpar.c: a plain file cut at guessed record starts, every piece parsed by its own reader, a piece kept only if the piece before it
    ended exactly where it starts -- four-line FASTQ in parallel, anything else (multi-line records, FASTA, '@' opening a quality line, CRLF,
    a truncated last record) through the verification's fall-back: the records are fmdh_seq_read's in every case."""
    from fermi_amd import hostlib
    tiny = gold.text_gz("tiny.fq.gz")
    rng = np.random.default_rng(3)
    files = {"tiny.fq": tiny, "special.fq": gold.text_gz("special.fq.gz")}
    recs = tiny.split(b"\n")
    # qualities that start with '@' and with '+', CRLF line ends
    lines = list(recs)
    for i in range(3, len(lines) - 1, 4):
        if rng.random() < 0.3:
            lines[i] = (b"@" if rng.random() < 0.5 else b"+") + lines[i][1:]
    files["at_quals.fq"] = b"\n".join(lines)
    files["crlf.fq"] = tiny.replace(b"\n", b"\r\n")
    # multi-line FASTQ (sequence and quality wrapped at 60) and FASTA
    ml, fa = [], []
    for i in range(0, len(recs) - 1, 4):
        s, q = recs[i + 1], recs[i + 3]
        ml += [recs[i]] + [s[j:j + 60] for j in range(0, len(s), 60)] + [b"+"] + [q[j:j + 60] for j in range(0, len(q), 60)]
        fa += [b">" + recs[i][1:]] + [s[j:j + 70] for j in range(0, len(s), 70)]
    files["multiline.fq"] = b"\n".join(ml) + b"\n"
    files["reads.fa"] = b"\n".join(fa) + b"\n"
    files["truncated.fq"] = tiny[: len(tiny) - 37]
    files["no_final_newline.fq"] = tiny.rstrip(b"\n")
    for name, data in files.items():
        path = str(tmp_path / name)
        open(path, "wb").write(data)
        want, _ = hostlib.read_records_serial(path)
        assert len(want) > 100
        for threads, span in ((2, 1 << 20), (7, 3 << 20), (16, 0), (5, 1 << 16)):
            got = hostlib.read_records_parallel(path, threads, span)
            assert got is not None and len(got) == len(want), (name, threads, span, len(got), len(want))
            assert got == want, (name, threads, span)
    gz = str(tmp_path / "tiny.fq.gz")
    open(gz, "wb").write(open(gold.path("tiny.fq.gz"), "rb").read())
    assert hostlib.read_records_parallel(gz, 4, 0) is None          # gzip: one zlib stream, one reader


@pytest.mark.parametrize("threads,chunk,spec", [(1, 0, 0), (4, 16, 0), (3, 16, 3), (16, 2048, 0)])
def test_walks_that_close_on_themselves(oracle_lib, gold, tmp_path, monkeypatch, threads, chunk, spec):
    """circle: two circular genomes tiled by reads (unitigs that come round to their own first read, unitig.c:247: the one piece of its own history a walk
    needs) and a linear one, ids shuffled; `fermi unitig -l40 -t1`'s MAG from the sequential loop and from the speculative chunks, with walks that are
    given up after three reads and run at commit among them (tests/golden/make_golden_circle.py made the fixture with the reference binary)."""
    from fermi_amd import hostlib
    monkeypatch.setenv("FMD_WALK_THREADS", str(threads))
    if chunk:
        monkeypatch.setenv("FMD_WALK_CHUNK", str(chunk))
    if spec:
        monkeypatch.setenv("FMD_WALK_SPEC_STEPS", str(spec))
    o = orcbind.OrcIndex(gold.path("circle.fmd"))
    n_seq = int(o.mcnt[1])
    rec, nei, seq = o.overlap_batch(np.arange(n_seq, dtype=np.uint64), 40, max_len=100, max_nei=8, n_threads=4)
    o.close()
    assert (rec["status"] == 0).sum() > 2000
    out = str(tmp_path / "o.mag")
    hostlib.unitig_walk(_packed_shards(rec, nei, seq, 2), n_seq, 40, out, max_nei=8, seq_stride=seq.shape[1], link=3)
    got = open(out, "rb").read()
    assert got == gold.text_gz("circle.mag.gz") and got.count(b"\n@") + 1 == 3


@pytest.mark.parametrize("name", ["tiny", "repeat", "special", "circle", "pairs", "dup32", "palin"])
def test_what_a_line_of_the_slim_table_leaves_to_the_other_strand(oracle_lib, gold, name):
    """The walk's 32-byte line (fmdh_wrec_t) keeps k[0] and ONE byte of the rank.  What it leaves out is in the record of the read's other strand, row id ^ 1:
    the bi-interval of `$read$` is (interval, interval of the reverse complement, size) -- k[1] of a row is k[0] of its other strand, k[2] is shared, and the
    two strands are short / not short together -- and the rank fm_retrieve returns (exact.c:59-70) is one of the k[2] sentinels of the interval."""
    o = orcbind.OrcIndex(gold.path(name + ".fmd"))
    n_seq = int(o.mcnt[1])
    assert n_seq % 2 == 0
    for mm in (10, 20, 30, 50):
        rec, _, _ = o.overlap_batch(np.arange(n_seq, dtype=np.uint64), mm, max_len=300, max_nei=16, n_threads=4)
        ok = (rec["status"] != -1) & ((rec["flags"] & 2) == 0)
        assert (ok[0::2] == ok[1::2]).all()
        k, m = rec["k"], ok[0::2]
        assert m.any()
        assert (k[0::2, 1] == k[1::2, 0])[m].all() and (k[0::2, 0] == k[1::2, 1])[m].all() and (k[0::2, 2] == k[1::2, 2])[m].all()
        assert ((rec["rank"] >= k[:, 0]) & (rec["rank"] < k[:, 0] + k[:, 2]))[ok].all()
        assert (rec["len"][0::2] == rec["len"][1::2]).all()
    o.close()


@pytest.mark.parametrize("name,mm", [("tiny", 50), ("repeat", 20), ("special", 20), ("circle", 40), ("palin", 40)])
@pytest.mark.parametrize("link", [0, 3])
def test_slim_table_keeps_what_the_walk_reads(oracle_lib, gold, tmp_path, monkeypatch, name, mm, link):
    """The walk runs over its own table (host/slim_table.c): 32 bytes per row + a short variable part instead of the packed rows (64-byte record, 32-byte
    neighbours, both strands' bases).  Same MAG as the reference from both of its neighbour forms (link = 0: host threads link the slim rows, 10-byte
    entries; link = 3: the links come with the table, one neighbour = its overlap alone), with the bases of a read kept once (odd rows take the reverse
    complement of the even row's), and with records kept whole (W_BIG) where a field outgrows the line -- forced here by a narrow k[2] (FMD_SLIM_BIG_K2=1:
    every interval of two or more identical reads)."""
    o = orcbind.OrcIndex(gold.path(name + ".fmd"))
    n_seq = int(o.mcnt[1])
    rec, nei, seq = o.overlap_batch(np.arange(n_seq, dtype=np.uint64), mm, max_len=100, max_nei=8, n_threads=4)
    want = gold.text_gz(name + ".mag.gz")
    fat_bytes = None
    for big in (False, True):
        if big:
            monkeypatch.setenv("FMD_SLIM_BIG_K2", "1")
        st = {}
        out = str(tmp_path / ("o%d.mag" % big))
        shards = _packed_shards(rec.copy(), nei, seq, 2)
        hostlib.unitig_walk(shards, n_seq, mm, out, max_nei=8, seq_stride=seq.shape[1], link=link, stats=st)
        assert open(out, "rb").read() == want
        ok = int(((rec["status"] == 0) & ((rec["flags"] & 2) == 0)).sum())
        if not big:
            fat_bytes = sum(p.nbytes + 8 * len(p) + len(v) for p, _, v in shards) + 12 * n_seq      # records, offsets, packed parts, row map + links
            assert st["big"] == 0 and st["undecided"] == 0
            assert st["own_seq"] == int(((rec["status"] == 0) & ((rec["flags"] & 2) == 0))[0::2].sum())       # the even rows only
            assert st["plain"] > 0 and st["bytes"] < 0.5 * fat_bytes, (st, fat_bytes)
        else:
            dup = int(((rec["k"][:, 2] > 1) & (rec["status"] != -1) & ((rec["flags"] & 2) == 0)).sum())
            assert st["big"] == dup and (dup > 0 or name not in ("tiny", "repeat")), (st, dup)
        assert st["own_seq"] <= ok
    o.close()


@pytest.mark.parametrize("name,mm", [("tiny", 50), ("repeat", 20), ("special", 20)])
@pytest.mark.parametrize("n_pieces", [1, 7])
def test_rows_folded_as_a_root_receives_them_walk_to_the_reference_mag(oracle_lib, gold, tmp_path, name, mm, n_pieces):
    """The root of an N-process job keeps no packed table (fmd_ovlp_dist_cfg_t.host_table = 2): every piece of every peer goes through fmdh_dist_root_sink into
    the slim rows -- ids in ANY order (a piece is a stretch of the sorted key order), bases with the even row of a read only -- and after the last one
    fmdh_dist_root_finish links them.  Here the pieces come from the oracle's table cut at random; the MAG must be `fermi unitig -t1`'s, byte for byte, and
    the table as small as the one the streamed chunks make."""
    import packref
    o = orcbind.OrcIndex(gold.path(name + ".fmd"))
    n_seq = int(o.mcnt[1])
    rec, nei, seq = o.overlap_batch(np.arange(n_seq, dtype=np.uint64), mm, max_len=100, max_nei=8, n_threads=4)
    o.close()
    order = np.random.default_rng(11 + n_pieces).permutation(n_seq)
    root = hostlib.DistRoot(n_seq, 128)
    try:
        for part in np.array_split(order, n_pieces):
            if len(part):
                prec, off, var = packref.pack_rows(rec[part], nei[part], seq[part], nei.shape[1])
                root.feed(part.astype(np.uint32), prec, off, var, nei.shape[1])
        assert root.rows() == n_seq
        nbytes = root.finish(mm)          # (no row exceeded a capacity, no edge is left to the exact kernel in these sets: no GPU needed)
        out = str(tmp_path / "o.mag")
        root.walk(mm, out)
        assert open(out, "rb").read() == gold.text_gz(name + ".mag.gz")
        st = {}
        hostlib.unitig_walk(_packed_shards(rec.copy(), nei, seq, 1), n_seq, mm, str(tmp_path / "p.mag"), max_nei=8, seq_stride=seq.shape[1], stats=st)
        assert nbytes <= st["bytes"] + 8 * n_seq + 4096, (nbytes, st)      # (rows by id start on multiples of 8 in the growing area)
    finally:
        root.close()


def test_a_root_that_misses_rows_says_so(oracle_lib, gold):
    o = orcbind.OrcIndex(gold.path("tiny.fmd"))
    n_seq = int(o.mcnt[1])
    o.close()
    root = hostlib.DistRoot(n_seq, 128)
    with pytest.raises(RuntimeError):
        root.finish(50)
    root.close()
