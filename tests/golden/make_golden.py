#!/usr/bin/env python3
"""Generate the golden fixtures in this directory FROM THE COMPILED REFERENCE.

Run in the build container only (needs oracle/_ref, i.e. /root/reference):
    make -C oracle ref && python tests/golden/make_golden.py
Everything written here is DATA: inputs and the reference's outputs on them (index files as
`fermi build` / `fermi ropebwt` wrote them, function results dumped through ctypes, CLI text).
"""
import gzip
import hashlib
import json
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import refbind  # noqa: E402
from fermi_amd import synth  # noqa: E402

FERMI = refbind.REF_BIN
TMP = "/tmp/fmd_golden"
os.makedirs(TMP, exist_ok=True)


def run(cmd, out=None):
    with open(out, "wb") if out else open(os.devnull, "wb") as fo:
        subprocess.check_call(cmd, stdout=fo, stderr=subprocess.DEVNULL)


def write_fq(reads_ascii, path):
    with open(path, "w") as f:
        for i, s in enumerate(reads_ascii):
            f.write("@r%d\n%s\n+\n%s\n" % (i, s, "I" * len(s)))


def md5(path):
    return hashlib.md5(open(path, "rb").read()).hexdigest()


def ascii_reads(nt6):
    tab = np.frombuffer(b"$ACGTN", dtype=np.uint8)
    return [tab[r].tobytes().decode() for r in nt6]


def overlap_table(R, n_seq, min_match, ids):
    recs = []
    for i in ids:
        r = R.overlap(int(i), min_match)
        r["ext"] = r.get("ext", b"").hex()
        recs.append(r)
    return recs


def vectors_for(R, reads_nt6, miss_nt6, seed, n_rank=4000, n_ext=3000):
    rng = np.random.default_rng(seed)
    n = int(R.mcnt[0])
    v = {}
    ks = np.unique(np.concatenate([rng.integers(0, n, n_rank), [0, 1, n - 1, n - 2],
                                   np.arange(0, n, max(1, n // 257))])).astype(np.uint64)
    ks = np.concatenate([ks, [np.uint64(0xFFFFFFFFFFFFFFFF)]]).astype(np.uint64)
    v["rank1a_k"] = ks
    v["rank1a_ok"], v["rank1a_sym"] = R.rank1a(ks)
    k2 = rng.integers(0, n, n_rank).astype(np.uint64)
    l2 = np.minimum(k2 + rng.integers(0, 700, n_rank).astype(np.uint64), np.uint64(n - 1))
    k2[:40] = np.uint64(0xFFFFFFFFFFFFFFFF)
    v["rank2a_k"], v["rank2a_l"] = k2, l2
    v["rank2a_ok"], v["rank2a_ol"] = R.rank2a(k2, l2)
    # extend: walk random reads so that the intervals are realistic, both directions, plus x==0 rows
    iks, backs = [], []
    for t in range(n_ext):
        rd = reads_nt6[rng.integers(0, len(reads_nt6))] if t % 5 else miss_nt6[rng.integers(0, len(miss_nt6))]
        L = int(rng.integers(1, 40))
        p = int(rng.integers(0, len(rd) - L))
        ik = np.zeros(1, dtype=refbind.INTV_DT)
        c = int(rd[p + L - 1])
        ik["x"][0] = (R.cnt[c], R.cnt[5 - c if 1 <= c <= 4 else c], R.cnt[c + 1] - R.cnt[c])
        ok_alive = True
        for j in range(p + L - 2, p - 1, -1):
            o = R.extend(ik, [1])[0]
            if o[int(rd[j])]["x"][2] == 0:
                break
            ik[0] = o[int(rd[j])]
        iks.append(ik[0].copy()); backs.append(int(rng.integers(0, 2)))
    # whole-index interval and sentinel interval exercise k == -1
    for c in range(6):
        ik = np.zeros(1, dtype=refbind.INTV_DT)
        ik["x"][0] = (R.cnt[c], R.cnt[5 - c if 1 <= c <= 4 else c], R.cnt[c + 1] - R.cnt[c])
        iks.append(ik[0].copy()); backs.append(1)
        iks.append(ik[0].copy()); backs.append(0)
    iks = np.array(iks, dtype=refbind.INTV_DT); backs = np.array(backs, dtype=np.uint8)
    v["ext_ik"], v["ext_back"] = iks.view(np.uint64).reshape(-1, 4), backs
    out = R.extend(iks, backs)
    out["info"] = 0  # fm6_extend leaves info unspecified
    v["ext_ok"] = out.view(np.uint64).reshape(-1, 24)
    # backward search
    q = np.concatenate([reads_nt6[:300], miss_nt6[:300]])
    v["bs_reads"] = q
    v["bs_cnt"], v["bs_beg"], v["bs_end"] = R.backward_search(list(q))
    short = [q[i][int(rng.integers(0, 60)):][:int(rng.integers(1, 40))] for i in range(200)]
    v["bs_short"] = np.array([np.pad(s, (0, 40 - len(s))) for s in short], dtype=np.uint8)
    v["bs_short_len"] = np.array([len(s) for s in short], dtype=np.int32)
    v["bs_short_cnt"], v["bs_short_beg"], v["bs_short_end"] = R.backward_search(short)
    # retrieve
    xs = np.unique(np.concatenate([rng.integers(0, int(R.mcnt[1]), 300), [0, 1, int(R.mcnt[1]) - 1]])).astype(np.uint64)
    v["ret_x"] = xs
    seqs = np.zeros((len(xs), 128), dtype=np.uint8); lens = np.zeros(len(xs), dtype=np.int32); rk = np.zeros(len(xs), dtype=np.uint64)
    for i, x in enumerate(xs):
        s, k = R.retrieve(int(x))
        seqs[i, :len(s)] = s; lens[i] = len(s); rk[i] = k
    v["ret_seq"], v["ret_len"], v["ret_rank"] = seqs, lens, rk
    v["traverse3"] = R.traverse(3).view(np.uint64).reshape(-1, 4)
    # SMEM (fm6_smem) on reads and on noisy reads, both self_match settings
    for sm in (0, 1):
        rows, offs = [], [0]
        qq = np.concatenate([reads_nt6[:80], miss_nt6[:80]])
        for s in qq:
            m = R.smem(s, sm)
            rows.append(m.view(np.uint64).reshape(-1, 4)); offs.append(offs[-1] + len(m))
        v["smem%d_reads" % sm] = qq
        v["smem%d_mem" % sm] = np.concatenate(rows) if rows else np.zeros((0, 4), np.uint64)
        v["smem%d_off" % sm] = np.array(offs, dtype=np.int64)
    return v


def main():
    man = {"reference": "lh3/fermi 1.1-r751-beta (oracle/_ref, reference Makefile flags)", "files": {}}

    # ---- fixture 1: tiny (2000 x 100 bp, 20x, 0.5 % substitutions) -------------------------
    N = 2000
    reads = synth.reads(synth.DEFAULT_SEED, N, coverage=20, err=0.005)
    miss = synth.reads(synth.DEFAULT_SEED, N, coverage=20, err=0.04)
    fq = os.path.join(TMP, "tiny.fq")
    write_fq(ascii_reads(reads), fq)
    with gzip.open(os.path.join(HERE, "tiny.fq.gz"), "wb", 9) as f:
        f.write(open(fq, "rb").read())
    run([FERMI, "build", "-fo", os.path.join(HERE, "tiny.fmd"), fq])
    subprocess.check_call("%s ropebwt -a bcr -v0 -btNf %s/t.tmp %s > %s" % (FERMI, TMP, fq, os.path.join(HERE, "tiny.rle.fmd")),
                          shell=True, stderr=subprocess.DEVNULL)
    run([FERMI, "chkbwt", "-p", os.path.join(HERE, "tiny.fmd")], os.path.join(TMP, "tiny.bwt.txt"))
    man["tiny_bwt_md5"] = md5(os.path.join(TMP, "tiny.bwt.txt"))
    R = refbind.RefIndex(os.path.join(HERE, "tiny.fmd"))
    v = vectors_for(R, reads, miss, 11)
    np.savez_compressed(os.path.join(HERE, "tiny_vectors.npz"), **v)
    # per-read overlap records for every sequence id, -l50 and a sample at -l30
    ov = {"l50": overlap_table(R, int(R.mcnt[1]), 50, range(int(R.mcnt[1]))),
          "l30": overlap_table(R, int(R.mcnt[1]), 30, range(0, int(R.mcnt[1]), 9))}
    with gzip.open(os.path.join(HERE, "tiny_overlap.json.gz"), "wt") as f:
        json.dump(ov, f)
    R.close()
    # CLI outputs
    run([FERMI, "unitig", "-l50", "-t1", os.path.join(HERE, "tiny.fmd")], os.path.join(TMP, "tiny.mag"))
    run([FERMI, "exact", os.path.join(HERE, "tiny.fmd"), fq], os.path.join(TMP, "tiny.exact"))
    run([FERMI, "exact", "-s", os.path.join(HERE, "tiny.fmd"), fq], os.path.join(TMP, "tiny.exact_s"))
    run([FERMI, "correct", "-t1", os.path.join(HERE, "tiny.fmd"), fq], os.path.join(TMP, "tiny.ec.fq"))
    run([FERMI, "unpack", os.path.join(HERE, "tiny.fmd")], os.path.join(TMP, "tiny.unpack"))
    for nm in ("tiny.mag", "tiny.exact", "tiny.exact_s", "tiny.ec.fq", "tiny.unpack"):
        with gzip.open(os.path.join(HERE, nm + ".gz"), "wb", 9) as f:
            f.write(open(os.path.join(TMP, nm), "rb").read())

    # ---- fixture 2: special (N bases, an even-length self-reverse-complement read, ragged lengths)
    rng = np.random.default_rng(5)
    sp = ascii_reads(synth.reads(77, 300, read_len=60, coverage=15, err=0.01))
    sp = [s[: int(rng.integers(35, 61))] for s in sp]
    for i in range(0, 300, 17):
        s = list(sp[i]); s[int(rng.integers(0, len(s)))] = "N"; sp[i] = "".join(s)
    sp.append("ACGTTGCAACGT" + "ACGTTGCAACGT"[::-1].translate(str.maketrans("ACGT", "TGCA")))  # palindrome, len 24
    sp.append("AACCGGTT")  # also its own reverse complement
    fq2 = os.path.join(TMP, "special.fq")
    write_fq(sp, fq2)
    with gzip.open(os.path.join(HERE, "special.fq.gz"), "wb", 9) as f:
        f.write(open(fq2, "rb").read())
    run([FERMI, "build", "-fo", os.path.join(HERE, "special.fmd"), fq2])
    R = refbind.RefIndex(os.path.join(HERE, "special.fmd"))
    sp_nt6 = [np.array([refbind_nt6(c) for c in s], dtype=np.uint8) for s in sp]
    v = {}
    n = int(R.mcnt[0])
    ks = np.arange(0, n, 3, dtype=np.uint64)
    v["rank1a_k"] = ks
    v["rank1a_ok"], v["rank1a_sym"] = R.rank1a(ks)
    xs = np.arange(int(R.mcnt[1]), dtype=np.uint64)
    seqs = np.zeros((len(xs), 64), dtype=np.uint8); lens = np.zeros(len(xs), dtype=np.int32); rk = np.zeros(len(xs), dtype=np.uint64)
    for i, x in enumerate(xs):
        s, k = R.retrieve(int(x)); seqs[i, :len(s)] = s; lens[i] = len(s); rk[i] = k
    v["ret_x"], v["ret_seq"], v["ret_len"], v["ret_rank"] = xs, seqs, lens, rk
    c, b, e = R.backward_search(sp_nt6)
    v["bs_cnt"], v["bs_beg"], v["bs_end"] = c, b, e
    np.savez_compressed(os.path.join(HERE, "special_vectors.npz"), **v)
    ov = {"l20": overlap_table(R, int(R.mcnt[1]), 20, range(int(R.mcnt[1])))}
    with gzip.open(os.path.join(HERE, "special_overlap.json.gz"), "wt") as f:
        json.dump(ov, f)
    R.close()

    # ---- fixture 3: dup32 -- one read 40 000x so that blocks hold >= 0x8000 symbols (rld.c:120)
    rd = ascii_reads(synth.reads(9, 1, read_len=100))[0]
    fq3 = os.path.join(TMP, "dup32.fq")
    write_fq([rd] * 40000, fq3)
    run([FERMI, "build", "-fo", os.path.join(HERE, "dup32.fmd"), fq3])
    R = refbind.RefIndex(os.path.join(HERE, "dup32.fmd"))
    n = int(R.mcnt[0])
    ks = np.unique(np.concatenate([np.random.default_rng(3).integers(0, n, 3000), np.arange(0, n, 40000), [n - 1]])).astype(np.uint64)
    ok, sym = R.rank1a(ks)
    np.savez_compressed(os.path.join(HERE, "dup32_vectors.npz"), read=np.frombuffer(rd.encode(), dtype=np.uint8),
                        rank1a_k=ks, rank1a_ok=ok, rank1a_sym=sym)
    R.close()

    # ---- fixture 4: repeat -- a genome with near-identical repeats and ragged reads: forks
    # (several irreducible neighbours), contained reads and the fake-fork fix-up (unitig.c:158-176)
    rng = np.random.default_rng(123)
    unit = rng.integers(1, 5, 400)

    def mut(u, k):
        u = u.copy(); idx = rng.choice(len(u), k, replace=False); u[idx] = 1 + (u[idx] + rng.integers(0, 3, k)) % 4
        return u
    g = np.concatenate([rng.integers(1, 5, 500), unit, rng.integers(1, 5, 300), mut(unit, 3), rng.integers(1, 5, 400),
                        mut(unit, 1), rng.integers(1, 5, 500), unit[:200], rng.integers(1, 5, 300)])
    rp = []
    for i in range(1500):
        L = int(rng.integers(40, 81)); p0 = int(rng.integers(0, len(g) - L)); r = g[p0:p0 + L]
        if rng.integers(0, 2):
            r = (5 - r)[::-1]
        rp.append(r.astype(np.uint8))
    fq4 = os.path.join(TMP, "repeat.fq")
    write_fq(ascii_reads(rp), fq4)
    with gzip.open(os.path.join(HERE, "repeat.fq.gz"), "wb", 9) as f:
        f.write(open(fq4, "rb").read())
    run([FERMI, "build", "-fo", os.path.join(HERE, "repeat.fmd"), fq4])
    R = refbind.RefIndex(os.path.join(HERE, "repeat.fmd"))
    ov = {"l20": overlap_table(R, int(R.mcnt[1]), 20, range(int(R.mcnt[1]))),
          "l35": overlap_table(R, int(R.mcnt[1]), 35, range(0, int(R.mcnt[1]), 3))}
    with gzip.open(os.path.join(HERE, "repeat_overlap.json.gz"), "wt") as f:
        json.dump(ov, f)
    R.close()
    run([FERMI, "unitig", "-l20", "-t1", os.path.join(HERE, "repeat.fmd")], os.path.join(TMP, "repeat.mag"))
    with gzip.open(os.path.join(HERE, "repeat.mag.gz"), "wb", 9) as f:
        f.write(open(os.path.join(TMP, "repeat.mag"), "rb").read())

    # ---- solid k-mer tables of `correct` phase 1 (correct.c:35-87, 341-356), dumped through
    # oracle/ref_ec_harness.c which compiles the reference's own correct.c
    import ctypes as C
    Lec = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref_ec.so"))
    Lec.refec_collect.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                  C.POINTER(C.c_void_p), C.POINTER(C.c_uint64), C.c_void_p]
    kv = {}
    for (w, mo) in [(17, 3), (21, 3), (23, 2)]:
        sl = w - 15 if w > 15 else 1
        b, k, v, n = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_uint64()
        cnt = (C.c_int64 * 2)()
        assert Lec.refec_collect(os.path.join(HERE, "tiny.fmd").encode(), w, mo, sl, C.byref(b), C.byref(k), C.byref(v), C.byref(n), cnt) == 0
        m = n.value
        B = np.frombuffer(C.string_at(b, m * 4), dtype=np.uint32); K = np.frombuffer(C.string_at(k, m * 4), dtype=np.uint32)
        V = np.frombuffer(C.string_at(v, m), dtype=np.uint8)
        o = np.lexsort([V, K, B])
        tag = "w%d_o%d" % (w, mo)
        kv[tag + "_bucket"], kv[tag + "_key"], kv[tag + "_val"] = B[o], K[o], V[o]
        kv[tag + "_cnt"] = np.array(list(cnt), dtype=np.int64)
    np.savez_compressed(os.path.join(HERE, "tiny_solid.npz"), **kv)

    # ---- seqsort + `unitig -r` (seqsort.c:37-70, unitig.c:22-29, 282): rank files and the MAGs they give
    for name, mm in (("tiny", 50), ("repeat", 20), ("special", 20)):
        rank = os.path.join(HERE, name + ".rank")
        with open(rank, "wb") as fo:
            subprocess.check_call([FERMI, "seqsort", os.path.join(HERE, name + ".fmd")], stdout=fo, stderr=subprocess.DEVNULL)
        run([FERMI, "unitig", "-l%d" % mm, "-t1", "-r", rank, os.path.join(HERE, name + ".fmd")], os.path.join(TMP, name + ".r.mag"))
        with gzip.open(os.path.join(HERE, name + ".r.mag.gz"), "wb", 9) as f:
            f.write(open(os.path.join(TMP, name + ".r.mag"), "rb").read())
    run([FERMI, "unitig", "-l20", "-t1", os.path.join(HERE, "special.fmd")], os.path.join(TMP, "special.mag"))
    with gzip.open(os.path.join(HERE, "special.mag.gz"), "wb", 9) as f:
        f.write(open(os.path.join(TMP, "special.mag"), "rb").read())

    for fn in sorted(os.listdir(HERE)):
        if fn.endswith((".fmd", ".gz", ".npz", ".rank")):
            man["files"][fn] = {"md5": md5(os.path.join(HERE, fn)), "bytes": os.path.getsize(os.path.join(HERE, fn))}
    with open(os.path.join(HERE, "MANIFEST.json"), "w") as f:
        json.dump(man, f, indent=1, sort_keys=True)
    print(json.dumps(man, indent=1))


def refbind_nt6(ch):
    return {"A": 1, "C": 2, "G": 3, "T": 4}.get(ch.upper(), 5)


if __name__ == "__main__":
    main()
