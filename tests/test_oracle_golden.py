"""CPU: the oracle (oracle/*.c) against the golden vectors dumped from the compiled reference
(tests/golden/make_golden.py).  This is what pins the oracle."""
import hashlib
import json
import os

import numpy as np
import pytest

import orcbind

U64 = np.uint64


def test_manifest_matches_files(gold):
    man = json.load(open(gold.path("MANIFEST.json")))
    for fn, meta in man["files"].items():
        if fn.endswith(".fmd") or fn.endswith(".npz"):
            assert os.path.getsize(gold.path(fn)) == meta["bytes"], fn
            assert hashlib.md5(open(gold.path(fn), "rb").read()).hexdigest() == meta["md5"], fn


@pytest.mark.parametrize("name", ["tiny", "special", "dup32"])
def test_rank1a_golden(oracle_lib, gold, name):
    v = gold.npz(name + "_vectors.npz")
    e = orcbind.OrcIndex(gold.path(name + ".fmd"))
    ok, sym = e.rank1a(v["rank1a_k"])
    assert np.array_equal(ok, v["rank1a_ok"])
    assert np.array_equal(sym, v["rank1a_sym"])
    e.close()


@pytest.mark.parametrize("name", ["tiny", "special", "dup32"])
def test_chkbwt_equivalent(oracle_lib, gold, name):
    """`fermi chkbwt -r` (cmd.c:90-105): rank1a at EVERY position equals the running counts of
    the sequentially decoded BWT; marginal counts match the header (cmd.c:108-115)."""
    e = orcbind.OrcIndex(gold.path(name + ".fmd"))
    bwt = e.decode_all()
    step = 1 if e.n < 600000 else 97
    ks = np.arange(0, e.n, step, dtype=U64)
    ok, sym = e.rank1a(ks)
    onehot = np.zeros((e.n, 6), dtype=np.int64)
    onehot[np.arange(e.n), bwt] = 1
    run = np.cumsum(onehot, axis=0)[::step]
    assert np.array_equal(ok.astype(np.int64), run)
    assert np.array_equal(sym, bwt[::step].astype(np.int8))
    assert np.array_equal(np.bincount(bwt, minlength=6).astype(U64), e.mcnt[1:])
    e.close()


def test_bwt_md5_matches_chkbwt_p(oracle_lib, gold):
    e = orcbind.OrcIndex(gold.path("tiny.fmd"))
    txt = np.frombuffer(b"$ACGTN", dtype=np.uint8)[e.decode_all()].tobytes() + b"\n"
    man = json.load(open(gold.path("MANIFEST.json")))
    assert hashlib.md5(txt).hexdigest() == man["tiny_bwt_md5"]
    e.close()


def test_rank2a_golden(tiny_oracle, gold):
    v = gold.npz("tiny_vectors.npz")
    ok, ol = tiny_oracle.rank2a(v["rank2a_k"], v["rank2a_l"])
    assert np.array_equal(ok, v["rank2a_ok"]) and np.array_equal(ol, v["rank2a_ol"])


def test_extend_golden(tiny_oracle, gold):
    v = gold.npz("tiny_vectors.npz")
    ik = v["ext_ik"].copy().view(orcbind.INTV_DT).reshape(-1)
    out = tiny_oracle.extend(ik, v["ext_back"])
    assert np.array_equal(out.view(U64).reshape(-1, 24), v["ext_ok"])


def test_backward_search_golden(tiny_oracle, gold):
    v = gold.npz("tiny_vectors.npz")
    cnt, beg, end = tiny_oracle.backward_search(v["bs_reads"])
    assert np.array_equal(cnt, v["bs_cnt"])
    hit = cnt > 0
    assert hit.sum() > 100 and (~hit).sum() > 100
    assert np.array_equal(beg[hit], v["bs_beg"][hit]) and np.array_equal(end[hit], v["bs_end"][hit])
    for i in range(len(v["bs_short_len"])):  # ragged short queries
        L = int(v["bs_short_len"][i])
        c, b, e = tiny_oracle.backward_search(v["bs_short"][i:i + 1, :L])
        assert c[0] == v["bs_short_cnt"][i]
        if c[0]:
            assert b[0] == v["bs_short_beg"][i] and e[0] == v["bs_short_end"][i]


@pytest.mark.parametrize("name,stride", [("tiny", 128), ("special", 64)])
def test_retrieve_golden(oracle_lib, gold, name, stride):
    v = gold.npz(name + "_vectors.npz")
    e = orcbind.OrcIndex(gold.path(name + ".fmd"))
    seqs, ln, rank = e.retrieve(v["ret_x"], stride=stride)
    assert np.array_equal(ln, v["ret_len"]) and np.array_equal(rank, v["ret_rank"])
    assert np.array_equal(seqs, v["ret_seq"])
    e.close()


def test_unpack_matches_cli(tiny_oracle, gold):
    """`fermi unpack` (cmd.c:122-171) prints every sequence via fm_retrieve."""
    lines = [l for l in gold.text_gz("tiny.unpack.gz").split(b"\n") if l]
    n_seq = int(tiny_oracle.mcnt[1])
    assert len(lines) == n_seq
    seqs, ln, rank = tiny_oracle.retrieve(np.arange(n_seq, dtype=U64), stride=128)
    tab = np.frombuffer(b"$ACGTN", dtype=np.uint8)
    mine = [tab[seqs[i, :ln[i]]].tobytes() + b"\t%d" % rank[i] for i in range(n_seq)]
    assert mine == lines


def test_traverse_golden(tiny_oracle, gold):
    v = gold.npz("tiny_vectors.npz")
    assert np.array_equal(tiny_oracle.traverse(3).view(U64).reshape(-1, 4), v["traverse3"])


@pytest.mark.parametrize("sm", [0, 1])
def test_smem_golden(tiny_oracle, gold, sm):
    v = gold.npz("tiny_vectors.npz")
    off = v["smem%d_off" % sm]
    for i, q in enumerate(v["smem%d_reads" % sm]):
        m = tiny_oracle.smem(q, sm).view(U64).reshape(-1, 4)
        assert np.array_equal(m, v["smem%d_mem" % sm][off[i]:off[i + 1]]), i


def test_exact_cli_text(tiny_oracle, gold):
    """`fermi exact` output (cmd.c:320-327, smem.c:412-418) rebuilt from oracle SMEMs."""
    reads = gold.fastq_nt6("tiny.fq.gz")[:200]
    want = gold.text_gz("tiny.exact.gz").split(b"//\n")[:200]
    mcnt1 = int(tiny_oracle.mcnt[1])
    for i, q in enumerate(reads):
        m = tiny_oracle.smem(q, 0)
        out = [b"SQ\tr%d\t%d\t%d" % (i, len(q), len(m))]
        for r in m:
            info = int(r["info"])
            out.append(b"EM\t%d\t%d\t%d\t%s%s" % (info >> 32 & 0x3fffffff, info & 0x3fffffff, min(int(r["x"][2]), 0xffffffff),
                                                    b"OT"[info >> 63:(info >> 63) + 1], b"OT"[int(r["x"][1]) < mcnt1:][:1]))
        assert b"\n".join(out) + b"\n" == want[i], i


@pytest.mark.parametrize("name,key", [("tiny", "l50"), ("tiny", "l30"), ("special", "l20"), ("repeat", "l20"), ("repeat", "l35")])
def test_overlap_records_golden(oracle_lib, gold, name, key):
    """Per-read overlap records = fm_retrieve + fm6_is_contained + fm6_get_nei (unitig.c:77,93)."""
    recs = gold.json_gz(name + "_overlap.json.gz")[key]
    e = orcbind.OrcIndex(gold.path(name + ".fmd"))
    mm = int(key[1:])
    n_nei = 0
    for want in recs:
        got = e.overlap(want["id"], mm)
        got["ext"] = got.get("ext", b"").hex()
        if "intv" in got:
            got["intv"] = list(got["intv"])
        got["nei"] = [list(x) for x in got.get("nei", [])] if "nei" in got else None
        if got["nei"] is None:
            del got["nei"]
        assert got == want, (got, want)
        n_nei += len(want.get("nei", []))
    assert n_nei > 0
    e.close()


def test_rle6_and_encoder_roundtrip(oracle_lib, gold, tmp_path):
    """RLE\\6 stream (ropebwt output) re-encodes to the byte-identical RLD\\2 file `fermi build`
    wrote; encoder + dump are byte-exact, incl. 32-bit headers (dup32)."""
    want = open(gold.path("tiny.fmd"), "rb").read()
    e = orcbind.OrcIndex(gold.path("tiny.rle.fmd"))
    e.dump(str(tmp_path / "a.fmd"))
    assert open(tmp_path / "a.fmd", "rb").read() == want
    e.close()
    for name in ("tiny", "special", "dup32"):
        e = orcbind.OrcIndex(gold.path(name + ".fmd"))
        e2 = orcbind.OrcIndex(bwt=e.decode_all())
        e2.dump(str(tmp_path / "b.fmd"))
        assert open(tmp_path / "b.fmd", "rb").read() == open(gold.path(name + ".fmd"), "rb").read(), name
        e.close(); e2.close()


@pytest.mark.parametrize("w,mo", [(17, 3), (21, 3), (23, 2)])
def test_ec_collect_golden(tiny_oracle, gold, w, mo):
    """The `solid` k-mer tables (correct.c:35-87) as a sorted multiset of (bucket, key, val)."""
    v = gold.npz("tiny_solid.npz")
    tag = "w%d_o%d" % (w, mo)
    sl = w - 15 if w > 15 else 1
    out, cnt = tiny_oracle.ec_collect(w, mo, sl)
    B = np.concatenate([np.full(len(k), i, dtype=np.uint32) for i, (k, _) in enumerate(out)])
    K = np.concatenate([k for k, _ in out]); V = np.concatenate([x for _, x in out])
    o = np.lexsort([V, K, B])
    assert np.array_equal(B[o], v[tag + "_bucket"]) and np.array_equal(K[o], v[tag + "_key"]) and np.array_equal(V[o], v[tag + "_val"])
    assert list(cnt) == list(v[tag + "_cnt"])


def _fastq_records(blob):
    lines = blob.split(b"\n")
    return [(lines[i], lines[i + 1], lines[i + 3]) for i in range(0, len(lines) - 1, 4)]


def finish_correct(reads_ascii, s, q, off, info, max_corr=0.3):
    """What the host does after ec_fix (correct.c:247-252, 412-425; unpaired, keep_bad = 0): FASTQ text."""
    nt6 = np.full(256, 5, dtype=np.uint8)
    for ch, v in zip(b"ACGTacgt", [1, 2, 3, 4, 1, 2, 3, 4]):
        nt6[ch] = v
    out = []
    for i, a in enumerate(reads_ascii):
        a = np.frombuffer(a, dtype=np.uint8)
        cs, cq = s[int(off[i]):int(off[i + 1])], q[int(off[i]):int(off[i + 1])].copy()
        same = nt6[a] == cs
        text = np.where(same, np.frombuffer(bytes(a).upper(), dtype=np.uint8), np.frombuffer(b"$acgtn", dtype=np.uint8)[cs])
        lower = (text >= ord("a")) & (text <= ord("z"))
        cq[lower] = 36
        inf = int(info[i])
        if len(a) and lower.sum() / len(a) > max_corr:
            inf |= 1 << 16
        if inf >> 18 <= 10:
            inf |= 1 << 16
        if not (inf >> 16 & 1):
            out.append(b"@%d_%d_%d\n%s\n+\n%s\n" % (i, inf & 0xffff, inf >> 18, text.tobytes(), cq.tobytes()))
    return b"".join(out)


def test_oracle_ec_fix_reproduces_fermi_correct(oracle_lib, gold):
    """oracle/ecfix_oracle.c (the checker of the GPU correction pass) over the golden solid table of tiny.fmd
    (k = 17, -O3) + the host's marking/filter rule == `fermi correct -t1` output, byte for byte."""
    import gzip
    v = gold.npz("tiny_solid.npz")
    recs = _fastq_records(gold.text_gz("tiny.fq.gz"))
    reads = [r[1] for r in recs]
    nt6 = gold.fastq_nt6("tiny.fq.gz")
    quals = [np.frombuffer(r[2], dtype=np.uint8) for r in recs]
    s, q, off, info = orcbind.ec_fix(17, v["w17_o3_bucket"], v["w17_o3_key"], v["w17_o3_val"], nt6, quals)
    assert finish_correct(reads, s, q, off, info) == gold.text_gz("tiny.ec.fq.gz")


def test_generator_rule_reproduces_the_reference_mag(tmp_path):
    """tools/mag_vs_generator.py states what `unitig -l50` prints for error-free synthetic reads from the generator alone (runs of start positions at most
    L - 50 apart: sequence, coverage string, number of reads, no neighbours) -- the exact check of `unitig` at sizes the reference cannot run (config 5,
    tests/test_gpu_cfg5.py).  Here the rule is held against a MAG the REFERENCE printed (tests/golden/gen_rule_20k.mag.gz, made by make_gen_rule.py:
    20 000 reads at 8-fold coverage, 346 unitigs, 8 of them single reads), and against MAGs it must reject."""
    import gzip, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import mag_vs_generator as mg
    mag = gzip.open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "gen_rule_20k.mag.gz")).read()
    p = str(tmp_path / "r.mag")
    open(p, "wb").write(mag)
    log = []
    assert mg.check(p, 20000, 50, 100, 8, log=log.append), log
    assert "346 runs expected, 346 matched exactly" in log[-1]
    recs = mag.split(b"\n@")
    bad = []
    bad.append(b"\n@".join(recs[:100] + recs[101:]))                                             # a unitig missing
    r = recs[5].split(b"\n"); r[3] = r[3][:40] + bytes([r[3][40] + 1]) + r[3][41:]               # one coverage character off by one
    bad.append(b"\n@".join(recs[:5] + [b"\n".join(r)] + recs[6:]))
    r = recs[7].split(b"\n"); r[1] = r[1][:-1] + (b"A" if r[1][-1:] != b"A" else b"C")             # one base
    bad.append(b"\n@".join(recs[:7] + [b"\n".join(r)] + recs[8:]))
    r = recs[9].split(b"\n"); h = r[0].split(b"\t"); h[1] = b"%d" % (int(h[1]) + 1); r[0] = b"\t".join(h)   # nsr
    bad.append(b"\n@".join(recs[:9] + [b"\n".join(r)] + recs[10:]))
    for m in bad:
        open(p, "wb").write(m)
        assert not mg.check(p, 20000, 50, 100, 8, log=lambda s: None)
