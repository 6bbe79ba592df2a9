"""Host helper + CPU test: a brute-force BWT of a read collection by sorting all suffixes, the
definition the index constructors are checked against (sentinels sort by sequence id:
ksa.c:54; text = read $ revcomp $ ..., cmd.c:457-469)."""
import numpy as np
import pytest

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


def test_word_wide_sort_keys_equal_the_symbol_loops(tmp_path):
    """fermi_amd/csrc/fmd_keys.inc (the sort keys of the GPU index build, compiled here for the host): the word-wide forms -- four aligned 8-byte loads
    of a byte text, three of a 4-bit text, symbols packed to 3-bit fields in registers -- against the symbol-by-symbol loops, on every alignment, every
    count of symbols 1..21 and windows at the very end of the text (where the loops take over)."""
    import os, subprocess
    src = tmp_path / "keys_test.c"
    src.write_text(r"""
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "fmd_keys.inc"
int main(void)
{
    const uint64_t n = 1 << 16;
    uint8_t *t8 = (uint8_t *)aligned_alloc(64, n + 64), *t4 = (uint8_t *)aligned_alloc(64, n / 2 + 64);
    uint64_t x = 88172645463325252ull, bad = 0, wide8 = 0, wide4 = 0;
    memset(t8, 0xee, n + 64); memset(t4, 0xee, n / 2 + 64);     /* what lies behind the text must not matter */
    for (uint64_t i = 0; i < n; ++i) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; t8[i] = (uint8_t)(x % 6); }
    for (uint64_t i = 0; i < n; i += 2) t4[i >> 1] = (uint8_t)(t8[i] | t8[i + 1] << 4);
    for (uint64_t a = 0; a < n; ++a) {
        const uint32_t room = n - a < 21 ? (uint32_t)(n - a) : 21;
        for (uint32_t mm = 1; mm <= room; mm += (a & 63) < 40 || a + 64 > n ? 1 : 5) {
            const uint64_t w8 = key_bytes(t8, a, mm);
            bad += chunk_key(t8, n, a, mm) != w8;
            bad += key_nibbles(t4, a, mm) != w8;                /* the two texts hold the same symbols */
            bad += chunk_key4(t4, n, a, mm) != w8;
        }
        wide8 += a + 32 <= n; wide4 += a + 48 <= n;
    }
    bad += chunk_key(t8 + 1, n - 1, 5, 21) != key_bytes(t8 + 1, 5, 21);   /* an unaligned text takes the loop */
    printf("%llu %llu %llu\n", (unsigned long long)bad, (unsigned long long)wide8, (unsigned long long)wide4);
    return bad != 0;
}
""")
    exe = tmp_path / "keys_test"
    inc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "fermi_amd", "csrc")
    subprocess.check_call(["gcc", "-O2", "-Wall", "-I", inc, str(src), "-o", str(exe)])
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    bad, wide8, wide4 = (int(v) for v in out.stdout.split())
    assert out.returncode == 0 and bad == 0, out.stdout
    assert wide8 > 65000 and wide4 > 65000      # (nearly every window took the word-wide form)


def _chunk_keys(text, send_of, ids, ch):
    """numpy restatement of k_chunk_keys64 (fmd_build.hip): symbols [21 ch, 21 ch + 21) of the suffix at ids[i], up to and including its '$', 3 bits each"""
    r = send_of[ids] - ids                                   # distance to the '$' that closes the sequence
    o0 = 21 * ch
    keys = np.zeros(len(ids), dtype=np.uint64)
    for j in range(21):
        use = o0 + j <= r
        t = np.minimum(ids + o0 + j, len(text) - 1)
        keys |= np.where(use, text[t].astype(np.uint64), 0).astype(np.uint64) << np.uint64(3 * (20 - j))
    return keys, r


@pytest.mark.parametrize("ragged", [False, True])
def test_setting_the_ended_suffixes_aside_is_the_stable_sort(ragged):
    """The claim behind FMD_BUILD_PARTITION (fmd_build.hip, build_bucketed): in the LSD pass of chunk ch a suffix has key 0 exactly when it ended before the
    chunk (distance to its '$' <= 21 ch), so [those, in their order] + [the rest, stably sorted by key] IS the stable sort of all -- restated in numpy
    over one bucket-free text (reads with Ns, equal or ragged lengths up to 70: four chunks), pass by pass, and the final order against sorted suffixes."""
    rng = np.random.default_rng(7 + ragged)
    n_reads = 300
    lens = rng.integers(1, 71, n_reads) if ragged else np.full(n_reads, 64)
    parts, ends = [], []
    for ln in lens:
        r = rng.integers(1, 5, ln).astype(np.uint8)
        r[rng.random(ln) < 0.02] = 5
        for s in (r, np.where((r >= 1) & (r <= 4), 5 - r, r)[::-1]):
            parts += [s, np.zeros(1, dtype=np.uint8)]
            ends.append(sum(len(p) for p in parts) - 1)
    text = np.concatenate(parts)
    n = len(text)
    send_of = np.array(ends, dtype=np.int64)[np.searchsorted(np.array(ends), np.arange(n))]
    n_chunks = (int(lens.max()) + 1 + 20) // 21
    plain = np.arange(n, dtype=np.int64)
    aside = plain.copy()
    for ch in range(n_chunks - 1, -1, -1):
        k, _ = _chunk_keys(text, send_of, plain, ch)
        plain = plain[np.argsort(k, kind="stable")]
        k, r = _chunk_keys(text, send_of, aside, ch)
        ended = r <= 21 * ch
        assert np.array_equal(ended, k == 0) or ch == 0      # (chunk 0 of the whole text: the suffixes that ARE a '$' have key 0 too; a bucket has none)
        if ch == 0:
            ended = k == 0
        rest = aside[~ended]
        aside = np.concatenate([aside[ended], rest[np.argsort(k[~ended], kind="stable")]])
        assert np.array_equal(aside, plain), ch
    want = sorted(range(n), key=lambda t: (bytes(text[t:send_of[t] + 1]), t))      # a '$' ends the comparison; ties go by text order = sequence id
    assert plain.tolist() == want
