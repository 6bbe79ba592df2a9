"""CPU, world_size 2, gloo: the N>1 path -- id sharding (start/step interleave) and the single gather of the
packed per-id rows on rank 0 -- on records the ORACLE computed from the golden index (not synthetic rows)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from fermi_amd import dist as fdist

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _oracle_rows(fmd, ids, min_match):
    import sys
    sys.path.insert(0, HERE)
    import orcbind
    import packref
    o = orcbind.OrcIndex(fmd)
    rec, nei, seq = o.overlap_batch(ids, min_match, 128, 4, 1, check_left=True)
    o.close()
    return packref.pack_rows(rec, nei, seq, 4)


def _worker(rank, world, port, fmd, n_ids, min_match, q, mode="device"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ids = fdist.shard_ids(n_ids, rank, world)
    prec, off, var = _oracle_rows(fmd, ids, min_match)
    cap = len(var) + 64     # the device buffer is larger than what is used: only off[-1] bytes travel
    var_t = torch.zeros(cap, dtype=torch.uint8); var_t[: len(var)] = torch.from_numpy(var)
    # the gather object of bench.py: buffers allocated in the first step, reused in the second.  mode: the direct batched form,
    # peer after peer through a staging buffer (what configs[4] needs on the root), or the padded all-gather a transport
    # without batch_isend_irecv falls back to (the probe is made to fail on every rank)
    g = fdist.PackedGather(torch, dist, n_ids, rank, world, force_path=None if mode == "fallback" else mode, timeout_s=120)
    if mode == "fallback":
        g._break_p2p = True
    fdist.describe_fabric(torch, dist, rank, world)
    args = (torch.from_numpy(prec.view(np.uint8).reshape(-1).copy()), torch.from_numpy(off.astype(np.int64)), var_t)
    first = g(*args)
    keep = None if first is None else [b[2].data_ptr() for b in first]
    got = g(*args)
    assert g.path == {"device": "device", "host-rounds": "host-rounds", "fallback": "all-gather"}[mode], g.path
    if rank == 0 and mode != "fallback":
        assert keep == [b[2].data_ptr() for b in got], "the receive buffers must be the first step's"
    if rank == 0:
        want_p, want_o, want_v = _oracle_rows(fmd, np.arange(n_ids, dtype=np.uint64), min_match)
        ok = len(got) == world
        for r in range(world):   # row i of the global table = arrays of rank i % world, index i // world
            p, o_, v = got[r]
            rows = np.arange(r, n_ids, world)
            ok = ok and p.numpy().tobytes() == want_p[rows].tobytes()
            lens = (o_[1:] - o_[:-1]).numpy()
            ok = ok and np.array_equal(lens, (want_o[rows + 1] - want_o[rows]).astype(np.int64)) and int(o_[-1]) == v.numel()
            for k in (0, len(rows) // 2, len(rows) - 1):
                i = rows[k]
                ok = ok and v[int(o_[k]):int(o_[k + 1])].numpy().tobytes() == want_v[int(want_o[i]):int(want_o[i + 1])].tobytes()
            # all variable parts of this rank, back to back, through the same index arithmetic bench.py's check uses
            pr, ln, vb = fdist.packed_rows(torch, got[r], torch.arange(len(rows), dtype=torch.int64))
            want_cat = b"".join(want_v[int(want_o[i]):int(want_o[i + 1])].tobytes() for i in rows)
            ok = ok and vb.numpy().tobytes() == want_cat
        q.put(bool(ok))
    else:
        assert got is None
    dist.barrier()
    dist.destroy_process_group()


def test_shard_ids_is_start_step_interleave():
    assert list(fdist.shard_ids(10, 1, 4)) == [1, 5, 9]
    allids = np.sort(np.concatenate([fdist.shard_ids(1001, r, 8) for r in range(8)]))
    assert np.array_equal(allids, np.arange(1001, dtype=np.uint64))
    assert [fdist.shard_size(1001, r, 8) for r in range(8)] == [len(fdist.shard_ids(1001, r, 8)) for r in range(8)]


def test_packed_rows_roundtrip(oracle_lib):
    import packref
    fmd = os.path.join(GOLD, "special.fmd")   # ragged lengths, Ns, palindromes: both packings occur
    import orcbind
    o = orcbind.OrcIndex(fmd)
    n = int(o.mcnt[1])
    rec, nei, seq = o.overlap_batch(np.arange(n, dtype=np.uint64), 20, 128, 4, 1, check_left=True)
    o.close()
    prec, off, var = packref.pack_rows(rec, nei, seq, 4)
    assert (prec["flags"] & packref.F_PACK4).any() and not (prec["flags"] & packref.F_PACK4).all()
    for i in range(n):
        ne, s = packref.unpack_row(prec[i], var[int(off[i]):int(off[i + 1])])
        if rec[i]["status"] != 0:
            assert len(s) == 0
            continue
        nb = rec[i]["len"] + rec[i]["ext_len"]
        assert np.array_equal(s, seq[i, :nb]) and ne.tobytes() == nei[i, :min(rec[i]["n_nei"], 4)].tobytes()


import pytest


@pytest.mark.parametrize("world,mode", [(2, "device"), (3, "host-rounds"), (2, "fallback")])
def test_gather_packed_records_gloo(oracle_lib, world, mode):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    n_ids = 1001  # odd: ranks hold different counts
    ps = [ctx.Process(target=_worker, args=(r, world, port, os.path.join(GOLD, "tiny.fmd"), n_ids, 50, q, mode)) for r in range(world)]
    for p in ps:
        p.start()
    ok = q.get(timeout=180)
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert ok


# ---- round 4: the gather in pieces and the key shard (fermi_amd/dist.py: the twin of fmd_ovlp_dist_step's exchange logic) ----------
def _fake_key(ids):
    """a 32-bit key per id standing in for the minimizer key: the smallest of 17 hashes, as a minimizer is -- crowded towards 0, so that equal
    key ranges would be very unequal shares (what the quantile splitters are there for)"""
    ids = np.asarray(ids, dtype=np.uint64)
    v = np.full(len(ids), 0xffffffff, dtype=np.uint64)
    for j in range(17):
        h = ((ids * np.uint64(17) + np.uint64(j)) * np.uint64(0x9E3779B97F4A7C15)) >> np.uint64(32)
        v = np.minimum(v, h)
    v[ids % np.uint64(97) == 5] = 0xffffffff     # strands that ended inside the head: stay at home
    v[ids % np.uint64(89) == 7] = 0xfffffffe     # no usable minimizer: stay at home
    return v.astype(np.uint32)


def _pieces_worker(rank, world, port, fmd, n_ids, min_match, pieces, key_shard, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    sys.path.insert(0, HERE)
    home = fdist.shard_ids(n_ids, rank, world)
    rows_of_rank = [fdist.shard_size(n_ids, r, world) for r in range(world)]
    own = home
    if key_shard:
        keys = _fake_key(home)
        order = np.argsort(keys, kind="stable")
        park = np.zeros((len(home), 64), dtype=np.uint8)
        park[:, 48:56] = home[:, None].view(np.uint8).reshape(len(home), 8)          # pad.x/y = the id, as k_ks_gather stamps it
        got, mat, split = fdist.key_shard_rows(torch, dist, rank, world, torch.from_numpy(park[order]), keys[order])
        own = got.numpy()[:, 48:56].copy().view(np.uint64).reshape(-1)
        # every row this rank owns has a key in its range, or is one of its own special rows; the shares are near n / world although
        # three quarters of the keys lie in the lowest eighth of the key space
        k_own = _fake_key(own).astype(np.uint64)
        sp = k_own >= 0xfffffffe
        assert ((k_own[~sp] >= split[rank]) & (k_own[~sp] < split[rank + 1])).all() and (own[sp] % world == rank).all()
        assert np.array_equal(fdist.key_dest(k_own[~sp], rank, split), np.full(int((~sp).sum()), rank))
        assert abs(len(own) - n_ids / world) < 0.15 * n_ids / world, (len(own), n_ids, world)
        cnt = torch.zeros(world, dtype=torch.int64); cnt[rank] = len(own)
        dist.all_reduce(cnt)
        rows_of_rank = [int(c) for c in cnt]
        assert sum(rows_of_rank) == n_ids and int(mat.sum()) == n_ids
    # this rank's rows in its computing order (a permutation: the sorted order of pass 2), cut into pieces
    rng = np.random.default_rng(100 + rank)
    comp = own[rng.permutation(len(own))]
    prec, off, var = _oracle_rows(fmd, comp, min_match)
    my = []
    for p in range(pieces):
        b, e = fdist.piece_begin(len(comp), p, pieces), fdist.piece_begin(len(comp), p + 1, pieces)
        o = off[b:e + 1].astype(np.int64) - int(off[b])
        my.append((torch.from_numpy(comp[b:e].astype(np.int32)), torch.from_numpy(prec[b:e].view(np.uint8).reshape(-1).copy()), torch.from_numpy(o),
                   torch.from_numpy(var[int(off[b]):int(off[e])].copy())))
    table = fdist.piecewise_exchange(torch, dist, rank, world, 0, pieces, rows_of_rank, my)
    if rank == 0:
        want_p, want_o, want_v = _oracle_rows(fmd, np.arange(n_ids, dtype=np.uint64), min_match)
        ok = len(table) == n_ids
        for i in range(n_ids):
            r, v = table[i]
            ok = ok and r == want_p[i].tobytes() and v == want_v[int(want_o[i]):int(want_o[i + 1])].tobytes()
        q.put(bool(ok))
    else:
        assert table is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,pieces,key_shard", [(2, 4, 0), (3, 5, 0), (2, 3, 1), (3, 4, 1), (2, 1, 1)])
def test_gather_in_pieces_and_key_shard_gloo(oracle_lib, world, pieces, key_shard):
    """pieces x {id shard, key shard} at world 2 / 3: the root ends up with every id's packed row, byte for byte the oracle's --
    i.e. key-shard rows == id-shard rows == the rows of one process."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    n_ids = 1001
    ps = [ctx.Process(target=_pieces_worker, args=(r, world, port, os.path.join(GOLD, "tiny.fmd"), n_ids, 50, pieces, key_shard, q)) for r in range(world)]
    for p in ps:
        p.start()
    ok = q.get(timeout=180)
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert ok


def _failing_worker(rank, world, port, fmd, n_ids, min_match, pieces, bad_rank, bad_piece, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    own = fdist.shard_ids(n_ids, rank, world)
    rows_of_rank = [fdist.shard_size(n_ids, r, world) for r in range(world)]
    prec, off, var = _oracle_rows(fmd, own, min_match)
    my = []
    for p in range(pieces):
        b, e = fdist.piece_begin(len(own), p, pieces), fdist.piece_begin(len(own), p + 1, pieces)
        if rank == bad_rank and p >= bad_piece:
            my.append(-5)                                # this rank ran out of memory packing piece p (FMD_E_NOMEM)
            continue
        o = off[b:e + 1].astype(np.int64) - int(off[b])
        my.append((torch.from_numpy(own[b:e].astype(np.int32)), torch.from_numpy(prec[b:e].view(np.uint8).reshape(-1).copy()), torch.from_numpy(o),
                   torch.from_numpy(var[int(off[b]):int(off[e])].copy())))
    try:
        fdist.piecewise_exchange(torch, dist, rank, world, 0, pieces, rows_of_rank, my)
        q.put((rank, None))
    except fdist.DistStepFailed as ex:
        q.put((rank, (ex.failed_rank, ex.code, ex.piece)))
    dist.barrier()                                       # nobody is left waiting in a send or a receive: all three get here
    dist.destroy_process_group()


@pytest.mark.parametrize("bad_rank,bad_piece", [(1, 0), (2, 2), (0, 1)])
def test_a_rank_that_fails_takes_every_rank_out_of_the_step_gloo(oracle_lib, bad_rank, bad_piece):
    """world 3, one rank fails at piece `bad_piece` (a peer or the root): the status word of the per-piece all-gather reaches everybody, all three leave the
    step with the same (rank, code, piece) and meet at the barrier behind it -- the protocol of fmd_ovlp_dist_step."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    world, pieces = 3, 4
    ps = [ctx.Process(target=_failing_worker, args=(r, world, port, os.path.join(GOLD, "tiny.fmd"), 601, 50, pieces, bad_rank, bad_piece, q)) for r in range(world)]
    for p in ps:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(world))
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got == {r: (bad_rank, -5, bad_piece) for r in range(world)}


def test_key_splitters_partition_the_key_space():
    for world in (1, 2, 3, 7, 8):
        split = np.array([0] + sorted(np.random.default_rng(world).integers(1, 0xfffffffe, world - 1).tolist()) + [0xfffffffe], dtype=np.uint64)
        ks = np.concatenate([np.arange(0, 1 << 12, dtype=np.uint64), np.random.default_rng(world + 9).integers(0, 0xfffffffe, 20000, dtype=np.uint64),
                             np.array([int(b) + d for b in split for d in (-1, 0, 1) if 0 <= int(b) + d < 0xfffffffe], dtype=np.uint64)])
        d = fdist.key_dest(ks, 0, split)
        assert d.min() >= 0 and d.max() < world
        for p in range(world):
            assert ((d == p) == ((ks >= split[p]) & (ks < split[p + 1]))).all()
        assert (fdist.key_dest(np.array([0xfffffffe, 0xffffffff], dtype=np.uint64), 5, split) == 5).all()
    assert list(fdist.local_quantiles(np.array([3, 5, 9, 11, 0xfffffffe, 0xffffffff], dtype=np.uint32), 2)) == [0, 9]


def _sink_worker(rank, world, port, fmd, min_match, pieces, break_sink, tmp, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    sys.path.insert(0, HERE)
    import orcbind
    import packref
    from fermi_amd import hostlib
    o = orcbind.OrcIndex(fmd)
    n_ids = int(o.mcnt[1])
    own = fdist.shard_ids(n_ids, rank, world)
    comp = own[np.random.default_rng(7 + rank).permutation(len(own))]          # the computing order of pass 2: not the id order
    rec, nei, seq = o.overlap_batch(comp, min_match, max_len=100, max_nei=8, n_threads=2)
    o.close()
    prec, off, var = packref.pack_rows(rec, nei, seq, 8)
    rows_of_rank = [fdist.shard_size(n_ids, r, world) for r in range(world)]
    my = []
    for p in range(pieces):
        b, e = fdist.piece_begin(len(comp), p, pieces), fdist.piece_begin(len(comp), p + 1, pieces)
        oo = off[b:e + 1].astype(np.int64) - int(off[b])
        my.append((torch.from_numpy(comp[b:e].astype(np.int32)), torch.from_numpy(prec[b:e].view(np.uint8).reshape(-1).copy()), torch.from_numpy(oo),
                   torch.from_numpy(var[int(off[b]):int(off[e])].copy())))
    root = hostlib.DistRoot(n_ids, 128) if rank == 0 else None
    fed = [0]

    def sink(ids, prec_u8, off_u64, var_u8):
        if break_sink and fed[0] >= 1:
            raise MemoryError("the consumer is out of room")
        fed[0] += 1
        root.feed(ids, prec_u8, off_u64, var_u8, 8)            # (records as the 64 bytes they are)

    try:
        fdist.piecewise_exchange(torch, dist, rank, world, 0, pieces, rows_of_rank, my, sink=sink if rank == 0 else True)
        if rank == 0:
            assert root.rows() == n_ids
            root.finish(min_match)
            out = os.path.join(tmp, "sink.mag")
            root.walk(min_match, out)
            q.put((rank, open(out, "rb").read()))
        else:
            q.put((rank, None))
    except fdist.DistStepFailed as ex:
        q.put((rank, (ex.failed_rank, ex.code)))
    if root:
        root.close()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,pieces,break_sink", [(2, 3, False), (3, 4, False), (3, 1, False), (3, 4, True)])
def test_root_folds_pieces_through_a_sink_gloo(oracle_lib, tmp_path, world, pieces, break_sink):
    """The N > 1 step with a root that keeps no table (fmd_ovlp_dist_cfg_t.host_table = 2; here the Python twin of its protocol over gloo, the C folding code of
    libfmdhost behind the sink): every peer's piece is folded into the rows `unitig` walks a piece behind the exchange, and the MAG the root prints is
    `fermi unitig -t1`'s.  A sink that fails takes every rank out of the step with FMD_E_IO, the root named as the one that failed."""
    import gzip
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_sink_worker, args=(r, world, port, os.path.join(GOLD, "tiny.fmd"), 50, pieces, break_sink, str(tmp_path), q)) for r in range(world)]
    for p in ps:
        p.start()
    got = dict(q.get(timeout=180) for _ in range(world))
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    if break_sink:
        assert got == {r: (0, -4) for r in range(world)}
    else:
        assert got[0] == gzip.open(os.path.join(GOLD, "tiny.mag.gz"), "rb").read()
        assert all(got[r] is None for r in range(1, world))
