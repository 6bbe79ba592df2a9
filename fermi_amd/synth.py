"""Deterministic synthetic read generator: counter-based splitmix64 so that the
same reads can be produced by C (fermi_amd/host/synth.c), numpy (here) or a kernel.

rnd(seed, stream, i) = the i-th output of splitmix64 seeded with  seed ^ (stream * K)
  genome base i      : 1 + (rnd(seed,1,i) >> 62)                          (nt6: A=1 C=2 G=3 T=4)
  read r position    : rnd(seed,2,r) % (G - L + 1)
  read r strand      : rnd(seed,3,r) >> 63          (1 = reverse complement of the genome window)
  read r base j error: u = rnd(seed,4,r*L+j); substitute iff (u >> 32) < floor(err * 2^32),
                       new base = 1 + ((old - 1) + 1 + (u & 0xffffffff) % 3) % 4
Genome length G = N * L / coverage (integer division).
"""
import numpy as np

_GOLD = np.uint64(0x9E3779B97F4A7C15)
_K = np.uint64(0xD1B54A32D192ED03)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)
DEFAULT_SEED = 20260928


def rnd(seed, stream, idx):
    """Vectorised splitmix64 output number `idx` (array) of stream `stream`."""
    with np.errstate(over="ignore"):
        s = np.uint64(seed) ^ (np.uint64(stream) * _K)
        z = s + (np.asarray(idx, dtype=np.uint64) + np.uint64(1)) * _GOLD
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        return z ^ (z >> np.uint64(31))


def genome(seed, n_reads, read_len=100, coverage=30):
    g = n_reads * read_len // coverage
    if g < read_len:
        g = read_len
    return (1 + (rnd(seed, 1, np.arange(g, dtype=np.uint64)) >> np.uint64(62))).astype(np.uint8)


def reads(seed, n_reads, read_len=100, coverage=30, err=0.0, start=0, count=None, gen=None):
    """Return uint8 [count, read_len] nt6 reads `start .. start+count` of the n_reads-read set (gen: the genome from genome(),
    to generate a large set piece by piece without recomputing it)."""
    if count is None:
        count = n_reads - start
    if gen is None:
        gen = genome(seed, n_reads, read_len, coverage)
    G = gen.shape[0]
    r = np.arange(start, start + count, dtype=np.uint64)
    pos = (rnd(seed, 2, r) % np.uint64(G - read_len + 1)).astype(np.int64)
    strand = (rnd(seed, 3, r) >> np.uint64(63)).astype(bool)
    win = gen[pos[:, None] + np.arange(read_len)[None, :]]
    rc = (5 - win)[:, ::-1]
    out = np.where(strand[:, None], rc, win).astype(np.uint8)
    if err > 0:
        thr = np.uint64(int(err * 4294967296.0))
        u = rnd(seed, 4, (r[:, None] * np.uint64(read_len) + np.arange(read_len, dtype=np.uint64)[None, :]))
        hit = (u >> np.uint64(32)) < thr
        sub = (1 + ((out.astype(np.uint64) - 1) + 1 + (u & np.uint64(0xFFFFFFFF)) % np.uint64(3)) % 4).astype(np.uint8)
        out = np.where(hit, sub, out)
    return np.ascontiguousarray(out)


def to_fastq(reads_nt6, path, qual="I"):
    tab = np.frombuffer(b"$ACGTN", dtype=np.uint8)
    with open(path, "wb") as fp:
        for i, r in enumerate(reads_nt6):
            s = tab[r].tobytes()
            fp.write(b"@r%d\n%s\n+\n%s\n" % (i, s, qual.encode() * len(s)))


# ---- the same generator on a GPU (torch int64 arithmetic wraps like uint64; shifts are made logical) ----
def _i64(v):
    v &= 0xFFFFFFFFFFFFFFFF
    return v - (1 << 64) if v >= (1 << 63) else v


def _lsr(t, k):
    return (t >> k) & ((1 << (64 - k)) - 1)


def rnd_torch(seed, stream, idx):
    """rnd() on a torch int64 tensor of indices; returns the 64-bit outputs as int64 bit patterns."""
    s = _i64(int(seed) ^ (int(stream) * int(_K)))
    z = (idx + 1) * _i64(int(_GOLD)) + s
    z = (z ^ _lsr(z, 30)) * _i64(int(_M1))
    z = (z ^ _lsr(z, 27)) * _i64(int(_M2))
    return z ^ _lsr(z, 31)


def _umod(u, m):
    """(u as unsigned 64-bit) % m for 0 < m < 2^47, int64 tensors only: Horner over the four 16-bit limbs."""
    r = _lsr(u, 48) % m
    for sh in (32, 16, 0):
        r = (r * 65536 + (_lsr(u, sh) & 0xFFFF)) % m
    return r


def genome_torch(seed, n_reads, read_len=100, coverage=30, device="cuda"):
    import torch
    g = max(n_reads * read_len // coverage, read_len)
    gen = torch.empty(g, dtype=torch.uint8, device=device)
    for s in range(0, g, 1 << 26):
        c = min(1 << 26, g - s)
        gen[s:s + c] = (1 + _lsr(rnd_torch(seed, 1, torch.arange(s, s + c, dtype=torch.int64, device=device)), 62)).to(torch.uint8)
    return gen


def reads_torch(seed, n_reads, read_len=100, coverage=30, err=0.0, device="cuda", chunk=2_000_000, start=0, count=None, gen=None):
    """reads() computed on `device`: uint8 [count, read_len] tensor (reads start .. start+count of the n_reads-read set),
    identical to the numpy form.  gen: the genome from genome_torch(), to generate a large set piece by piece."""
    import torch
    if count is None:
        count = n_reads - start
    if gen is None:
        gen = genome_torch(seed, n_reads, read_len, coverage, device)
    span = gen.shape[0] - read_len + 1
    out = torch.empty((count, read_len), dtype=torch.uint8, device=device)
    ar = torch.arange(read_len, dtype=torch.int64, device=device)
    thr = int(err * 4294967296.0)
    for s0 in range(0, count, chunk):
        c = min(chunk, count - s0)
        s = start + s0
        r = torch.arange(s, s + c, dtype=torch.int64, device=device)
        pos = _umod(rnd_torch(seed, 2, r), span)
        strand = _lsr(rnd_torch(seed, 3, r), 63).bool()
        win = gen[pos[:, None] + ar[None, :]]
        rc = (5 - win).flip(1)
        o = torch.where(strand[:, None], rc, win)
        if err > 0:
            u = rnd_torch(seed, 4, r[:, None] * read_len + ar[None, :])
            hit = _lsr(u, 32) < thr
            sub = (1 + ((o.to(torch.int64) - 1) + 1 + (u & 0xFFFFFFFF) % 3) % 4).to(torch.uint8)
            o = torch.where(hit, sub, o)
        out[s0:s0 + c] = o
    return out


# ---- a repeat-rich genome and ragged reads (what a real read set looks like to fm6_get_nei: repeats make forks and wide intervals,
# substrings of other reads are contained, lengths differ) -- numpy only, deterministic from the seed -----------------------------
def repeat_genome(seed, g_len, repeat_frac=0.05, min_len=300, max_len=5000, min_copies=2, max_copies=50):
    """Random genome of g_len bases in which families of repeats (a segment of min_len..max_len bases pasted at min_copies..max_copies
    random places, every other copy reverse-complemented) cover about repeat_frac of the positions."""
    gen = (1 + (rnd(seed, 21, np.arange(g_len, dtype=np.uint64)) >> np.uint64(62))).astype(np.uint8)
    covered, f = 0, 0
    while covered < repeat_frac * g_len:
        ln = int(min_len + rnd(seed, 22, f) % np.uint64(max_len - min_len + 1))
        nc = int(min_copies + rnd(seed, 23, f) % np.uint64(max_copies - min_copies + 1))
        ln = min(ln, g_len // 4)
        seg = (1 + (rnd(seed, 24, np.uint64(f) * np.uint64(1 << 20) + np.arange(ln, dtype=np.uint64)) >> np.uint64(62))).astype(np.uint8)
        for c in range(nc):
            p = int(rnd(seed, 25, f * 64 + c) % np.uint64(g_len - ln + 1))
            gen[p:p + ln] = seg if c % 2 == 0 else (5 - seg)[::-1]
        covered += ln * nc
        f += 1
    return gen


def ragged_reads(seed, n_reads, gen, min_len=70, max_len=150, err=0.01, dup_frac=0.01, sub_frac=0.01):
    """n_reads reads of min_len..max_len bases at uniform positions and strands of `gen`, substitutions with probability err per
    base; dup_frac of the reads are exact copies of an earlier read, sub_frac are proper substrings of one (1..20 bases trimmed
    from each end: contained reads).  -> list of uint8 arrays (nt6)."""
    G = len(gen)
    r = np.arange(n_reads, dtype=np.uint64)
    ln = (min_len + rnd(seed, 31, r) % np.uint64(max_len - min_len + 1)).astype(np.int64)
    pos = (rnd(seed, 32, r) % (np.uint64(G) - ln.astype(np.uint64) + np.uint64(1))).astype(np.int64)
    strand = (rnd(seed, 33, r) >> np.uint64(63)).astype(bool)
    kind = rnd(seed, 34, r) % np.uint64(1000000)
    is_dup = kind < np.uint64(int(dup_frac * 1e6))
    is_sub = (~is_dup) & (kind < np.uint64(int((dup_frac + sub_frac) * 1e6)))
    thr = np.uint64(int(err * 4294967296.0))
    out = [None] * n_reads
    CH = 200000
    for s in range(0, n_reads, CH):
        e = min(n_reads, s + CH)
        idx = pos[s:e, None] + np.arange(max_len)[None, :]
        win = gen[np.minimum(idx, G - 1)]
        u = rnd(seed, 35, (r[s:e, None] * np.uint64(max_len) + np.arange(max_len, dtype=np.uint64)[None, :]))
        hit = (u >> np.uint64(32)) < thr
        sub = (1 + ((win.astype(np.uint64) - 1) + 1 + (u & np.uint64(0xFFFFFFFF)) % np.uint64(3)) % 4).astype(np.uint8)
        win = np.where(hit, sub, win).astype(np.uint8)
        for i in range(s, e):
            w = win[i - s, :ln[i]]
            out[i] = np.ascontiguousarray((5 - w)[::-1]) if strand[i] else w.copy()
    back = (1 + rnd(seed, 36, r) % np.uint64(1000)).astype(np.int64)
    t5 = (1 + rnd(seed, 37, r) % np.uint64(20)).astype(np.int64)
    t3 = (1 + rnd(seed, 38, r) % np.uint64(20)).astype(np.int64)
    for i in np.nonzero(is_dup | is_sub)[0]:
        src = out[max(0, int(i) - int(back[i]))] if i > 0 else out[0]
        if is_dup[i] or len(src) <= int(t5[i] + t3[i]) + 32:
            out[i] = src.copy()
        else:
            out[i] = src[int(t5[i]): len(src) - int(t3[i])].copy()
    return out
