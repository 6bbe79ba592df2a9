"""Host helper + CPU test: a brute-force BWT of a read collection by sorting all suffixes, the
definition the index constructors are checked against (sentinels sort by sequence id:
ksa.c:54; text = read $ revcomp $ ..., cmd.c:457-469)."""
import numpy as np

import orcbind


def bwt_by_sorting(seqs_padded):
    """seqs_padded: uint8 [n_seq, L+1], each row = sequence then 0 ('$').  Equal-length rows."""
    n_seq, L1 = seqs_padded.shape
    # key of suffix (s, p) = row[p:] zero-padded to L1 symbols, then s
    pad = np.zeros((n_seq, 2 * L1), dtype=np.uint8)
    pad[:, :L1] = seqs_padded
    idx = np.arange(L1)
    keys = np.empty((n_seq * L1, L1), dtype=np.uint8)
    for p in range(L1):
        keys[p::L1] = pad[:, p:p + L1]
    sid = np.repeat(np.arange(n_seq), L1)
    pos = np.tile(idx, n_seq)
    order = np.lexsort([sid] + [keys[:, j] for j in range(L1 - 1, -1, -1)])
    prev = np.where(pos[order] == 0, 0, seqs_padded[sid[order], pos[order] - 1])
    return prev.astype(np.uint8)


def test_bruteforce_bwt_equals_fermi_build(oracle_lib, gold):
    reads = gold.fastq_nt6("tiny.fq.gz")
    r = np.array(reads, dtype=np.uint8)
    both = np.zeros((2 * len(r), 101), dtype=np.uint8)
    both[0::2, :100] = r
    both[1::2, :100] = (5 - r)[:, ::-1]
    bwt = bwt_by_sorting(both)
    o = orcbind.OrcIndex(gold.path("tiny.fmd"))
    assert np.array_equal(bwt, o.decode_all())
    o.close()
