"""GPU: the N > 1 step behind the C ABI (fmd_ovlp_dist_*, fermi_amd/csrc/fmd_ovlp_dist.hip).

One GPU is what the box has, so the ranks share it: world 2 / 3 processes on device 0, the transport handed to the C orchestrator
through the fmd_comm_t callbacks is torch.distributed/gloo (fermi_amd.dist.TorchComm) -- every kernel, every piece, the all-to-all of
the parked strands and the placement at the root are the product's; only the wire is not RCCL.  The RCCL transport itself
(fmd_comm_rccl_*: dlopen, ncclCommInitRank, ncclAllGather, grouped ncclSend / ncclRecv) is exercised with one rank talking to itself.
Expected rows: one process, fmd_ovlp_sorted_dev + the numpy statement of the packed format (tests/packref.py)."""
import ctypes as C
import os
import socket

import numpy as np
import pytest

import packref
from fermi_amd import synth

pytestmark = pytest.mark.gpu
U64 = np.uint64
HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _reads(N, err, ragged):
    reads = list(synth.reads(synth.DEFAULT_SEED + 77, N, 100, 30, err))
    if ragged:
        rng = np.random.default_rng(5)
        for i in range(0, N, 9):
            reads[i] = reads[i][: int(rng.integers(3, 60))]      # some end inside the head (their records are made there), some just behind it
        for i in range(4, N, 31):
            reads[i] = reads[i].copy(); reads[i][int(rng.integers(0, len(reads[i])))] = 5
    return reads


def _d2h(api, ptr, nbytes):
    a = np.empty(nbytes, dtype=np.uint8)
    if nbytes:
        api.check(api.lib().fmd_memcpy_d2h(a.ctypes.data, C.c_void_p(ptr), nbytes, None))
    return a


def _table_rows(api, t, n_ids, max_nei, stride, sample):
    """(records of all ids in id order, {id: variable part} for the ids of `sample`) read out of the root's table."""
    get = (lambda p, n: np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), (n,)).copy()) if t.on_host else (lambda p, n: _d2h(api, p, n))
    prec = get(t.prec, n_ids * 64).view(api.OVLP_DT)
    ids = get(t.ids, n_ids * 4).view(np.uint32)
    vaddr = get(t.vaddr, n_ids * 8).view(U64)
    row_of = get(t.row_of_id, n_ids * 4).view(np.uint32)
    assert np.array_equal(np.sort(ids), np.arange(n_ids, dtype=np.uint32)), "every id exactly once"
    assert np.array_equal(ids[row_of], np.arange(n_ids, dtype=np.uint32))
    rec_by_id = prec[row_of]
    var = {}
    for i in sample:
        r = rec_by_id[i]
        nb = 0
        if r["status"] == 0 and not (r["flags"] & packref.F_OVERFLOW):
            nbases = min(int(r["len"]) + int(r["ext_len"]), stride)
            sb = (nbases + 1) // 2 if (r["flags"] & packref.F_PACK4) else (nbases + 3) // 4
            nb = min(int(r["n_nei"]), max_nei) * 32 + (sb + 7) // 8 * 8
        var[int(i)] = get(int(vaddr[row_of[i]]), nb).tobytes()
    return rec_by_id, var


def _expected(api, d, n_ids, mm, L, max_nei):
    rec, nei, seq = d.overlap_sorted(np.arange(n_ids, dtype=U64), mm, L, max_nei, 0)
    return packref.pack_rows(rec, nei, seq, max_nei)


def _worker(rank, world, port, N, err, ragged, mm, pieces, key_shard, host_table, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    from fermi_amd import api, dist as fdist
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    reads = _reads(N, err, ragged)
    d = api.DevIndex.from_bwt(api.build_bwt(reads))        # every rank its own replica, as on N GPUs
    n_ids = int(d.mcnt[1])
    comm = fdist.TorchComm(api, dist, rank, world)
    job = fdist.DistJob(api, d, comm, n_ids, mm, 100, 4, pieces=pieces, key_shard=key_shard, root=0, host_table=host_table)
    ok = True
    for step in range(2):                                  # the second step reuses every buffer of the first
        st = job.step()
        assert st.pieces == pieces and (rank != 0 or st.on_host == (1 if host_table > 0 else 0)) and st.two_pass == (1 if mm >= 32 else 0)
        assert st.key_shard == (1 if key_shard and world > 1 and mm >= 32 else 0)
        if rank == 0:
            t = job.table()
            want_p, want_o, want_v = _expected(api, d, n_ids, mm, 100, 4)
            sample = np.unique(np.concatenate([np.arange(0, n_ids, 7), np.arange(n_ids - 50, n_ids)]))
            got_rec, got_var = _table_rows(api, t, n_ids, 4, 200, sample)
            if got_rec.tobytes() != want_p.tobytes():
                bad = np.nonzero(got_rec.view(np.uint8).reshape(n_ids, 64) != want_p.view(np.uint8).reshape(n_ids, 64))[0]
                print("step %d: %d records differ, first id %d:\n got  %r\n want %r" % (step, len(np.unique(bad)), bad[0], got_rec[bad[0]], want_p[bad[0]]), flush=True)
                ok = False
            for i in sample:
                if got_var[int(i)] != want_v[int(want_o[i]):int(want_o[i + 1])].tobytes():
                    print("step %d: variable part of id %d differs (%d bytes against %d)" % (step, i, len(got_var[int(i)]), int(want_o[i + 1] - want_o[i])), flush=True)
                    ok = False
                    break
            if not (st.bytes_received > 60 * (n_ids - st.rows_computed) and st.rows_computed > 0):
                print("step %d: stats %r" % (step, st.as_dict()), flush=True)
                ok = False
        else:
            assert st.rows_sent == st.rows_computed and st.bytes_sent > 64 * st.rows_sent
        if key_shard and world > 1 and mm >= 32:
            assert st.key_rows_sent > 0
    # the rows a rank computed are where fmd_ovlp_dist_local says, under the ids it says
    n_loc, p_ids, p_rec, _, _, stride = job.local()
    ids_loc = _d2h(api, p_ids, n_loc * 8).view(U64)
    rec_loc = _d2h(api, p_rec, n_loc * 64).view(api.OVLP_DT)
    cnt = torch.zeros(1, dtype=torch.int64); cnt[0] = n_loc
    dist.all_reduce(cnt)
    assert int(cnt[0]) == n_ids and stride == 200
    if not key_shard:
        assert np.array_equal(np.sort(ids_loc), np.arange(rank, n_ids, world, dtype=U64))     # (its id shard, in the order of pass 2)
    sub = np.arange(0, n_loc, 11)
    w_rec, _, _ = d.overlap(ids_loc[sub], mm, 100, 4, check_left=False)
    assert rec_loc[sub].tobytes() == w_rec.tobytes()
    if rank == 0:
        q.put(bool(ok))
    dist.barrier()
    job.free(); d.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,N,err,ragged,mm,pieces,key_shard,host_table", [
    (2, 20000, 0.0, False, 50, 3, 0, 0),      # id shard, three pieces, table in HBM
    (2, 20000, 0.01, True, 50, 4, 1, 0),      # key shard on reads with errors, ragged lengths and Ns
    (3, 15000, 0.005, True, 45, 2, 1, 1),     # three ranks, key shard, table in pinned host memory through the staging sets
    (3, 15000, 0.0, False, 50, 5, 0, 1),      # id shard, host table, more pieces than staging sets
    (2, 8000, 0.0, False, 25, 2, 1, 0),       # below the split: no two-pass form, the shard in id order, key shard switched off inside
])
def test_dist_step_ranks_sharing_one_gpu(gpu, world, N, err, ragged, mm, pieces, key_shard, host_table):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, world, port, N, err, ragged, mm, pieces, key_shard, host_table, q)) for r in range(world)]
    for p in ps:
        p.start()
    ok = q.get(timeout=240)
    for p in ps:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert ok


def _failing_worker(rank, world, port, inject, key_shard, host_table, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import ctypes as C
    import torch
    import torch.distributed as dist
    from fermi_amd import api, dist as fdist
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    d = api.DevIndex.from_bwt(api.build_bwt(_reads(12000, 0.0, False)))
    n_ids = int(d.mcnt[1])
    comm = fdist.TorchComm(api, dist, rank, world)
    job = fdist.DistJob(api, d, comm, n_ids, 50, 100, 4, pieces=3, key_shard=key_shard, root=0, host_table=host_table)
    job.step()                                                       # a good step first
    os.environ["FMD_DIST_INJECT"] = inject                           # "<rank>:<where>[:<piece>]": that rank fails there, on its own
    rc = api.lib().fmd_ovlp_dist_step(job.h, None, C.byref(job.stats))
    del os.environ["FMD_DIST_INJECT"]
    torch.cuda.synchronize()
    q.put((rank, int(rc)))
    dist.barrier()                                                   # every rank got out of the step: nobody hangs in a collective
    st = job.step()                                                  # and the job is good for another step
    ok = st.rows_computed > 0
    if rank == 0:
        t = job.table()
        want_p, _, _ = _expected(api, d, n_ids, 50, 100, 4)
        sample = np.arange(0, n_ids, 5)
        got_rec, _ = _table_rows(api, t, n_ids, 4, 200, sample)
        ok = ok and got_rec.tobytes() == want_p.tobytes()
    q.put((100 + rank, bool(ok)))
    dist.barrier()
    job.free(); d.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("inject,key_shard,host_table", [
    ("1:head", 0, 0),        # a peer's pass 1 fails; no collective until the first piece's sizes: they carry it
    ("2:head", 1, 0),        # ... with the key exchange ahead: the agreement in front of it
    ("1:keys", 1, 0),        # inside the key exchange: the status slot of the counts
    ("2:pack:1", 0, 1),      # a peer cannot pack its second piece (the first one has been gathered already)
    ("0:arena:2", 0, 0),     # the ROOT cannot make room for the last piece: the peers must not be left sending
    ("0:pack:0", 1, 1),      # the root's own first piece
])
def test_a_failing_rank_takes_all_ranks_out_of_the_step_with_one_code(gpu, inject, key_shard, host_table):
    """fmd_ovlp_dist_step used to return on a rank-local error while the peers sat in the matching collective.  Three ranks share
    the GPU; one is made to fail (FMD_DIST_INJECT); all three must come out of the step with the same code (FMD_E_NOMEM), meet at a barrier behind
    it, and run a correct step afterwards."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    world = 3
    ps = [ctx.Process(target=_failing_worker, args=(r, world, port, inject, key_shard, host_table, q)) for r in range(world)]
    for p in ps:
        p.start()
    got = dict(q.get(timeout=240) for _ in range(2 * world))
    for p in ps:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert [got[r] for r in range(world)] == [-5] * world, got
    assert all(got[100 + r] for r in range(world)), got


def test_head_and_tail_in_pieces_equal_the_sorted_job(gpu):
    """fmd_ovlp_head_dev + fmd_ovlp_tail_dev over slices of the order = fmd_ovlp_sorted_dev; fmd_ovlp_pack_rows_dev of a slice =
    the rows of that slice under their ids."""
    import torch
    api = gpu
    L = api.lib()
    N = 20000
    d = api.DevIndex.from_bwt(api.build_bwt(_reads(N, 0.01, True)))
    n = int(d.mcnt[1])
    dev = torch.device("cuda", 0)
    ids = torch.arange(n, dtype=torch.int64, device=dev)
    mm, ML, max_nei, stride = 50, 100, 4, 200
    assert L.fmd_ovlp_two_pass_ok(d.h, n, mm, ML) == 1 and L.fmd_ovlp_two_pass_ok(d.h, n, 20, ML) == 0
    rec = torch.zeros(n * 64, dtype=torch.uint8, device=dev); nei = torch.zeros(n * max_nei * 32, dtype=torch.uint8, device=dev); seq = torch.zeros(n * stride, dtype=torch.uint8, device=dev)
    park = torch.zeros(n * 64, dtype=torch.uint8, device=dev); keys = torch.zeros(n, dtype=torch.int32, device=dev); order = torch.zeros(n, dtype=torch.int32, device=dev)
    hb = L.fmd_ovlp_head_work_bytes(n)
    work = torch.empty(max(hb, L.fmd_ovlp_work_bytes(n, ML, mm)), dtype=torch.uint8, device=dev)
    api.check(L.fmd_ovlp_head_dev(d.h, None, n, ids.data_ptr(), mm, ML, rec.data_ptr(), park.data_ptr(), keys.data_ptr(), order.data_ptr(), work.data_ptr(), work.numel()))
    torch.cuda.synchronize()
    k = keys.cpu().numpy().view(np.uint32)
    assert (np.diff(k.astype(np.int64)) >= 0).all() and np.array_equal(np.sort(order.cpu().numpy()), np.arange(n))
    cuts = [0, n // 5, n // 5, n // 2 + 3, n]          # (an empty piece among them)
    for a, b in zip(cuts[:-1], cuts[1:]):
        api.check(L.fmd_ovlp_tail_dev(d.h, None, b - a, order.data_ptr() + 4 * a, park.data_ptr(), mm, ML, max_nei, rec.data_ptr(), nei.data_ptr(), seq.data_ptr(), stride, work.data_ptr(), work.numel()))
    torch.cuda.synchronize()
    want = d.overlap_sorted(np.arange(n, dtype=U64), mm, ML, max_nei, 0)
    g_rec = rec.cpu().numpy().view(api.OVLP_DT); g_nei = nei.cpu().numpy().view(api.INTV_DT).reshape(n, max_nei); g_seq = seq.cpu().numpy().reshape(n, stride)
    assert g_rec.tobytes() == want[0].tobytes()
    want_p, want_o, want_v = packref.pack_rows(g_rec, g_nei, g_seq, max_nei)
    # a piece packed through the row map, ids from a table and from first + step * row
    a, b = n // 5, n // 2 + 3
    m = b - a
    cap = L.fmd_ovlp_pack_max_bytes(m, max_nei, stride)
    pid = torch.zeros(m, dtype=torch.int32, device=dev); prec = torch.zeros(m * 64, dtype=torch.uint8, device=dev); off = torch.zeros(m + 1, dtype=torch.int64, device=dev)
    var = torch.zeros(cap, dtype=torch.uint8, device=dev); pw = torch.empty(L.fmd_ovlp_pack_work_bytes(m), dtype=torch.uint8, device=dev)
    row_ids = (ids * 3 + 1).contiguous()
    for tab, first, step in ((None, 5, 2), (row_ids.data_ptr(), 0, 0)):
        api.check(L.fmd_ovlp_pack_rows_dev(d.h, None, m, order.data_ptr() + 4 * a, tab, first, step, rec.data_ptr(), nei.data_ptr(), max_nei, seq.data_ptr(), stride,
                                           pid.data_ptr(), prec.data_ptr(), off.data_ptr(), var.data_ptr(), cap, pw.data_ptr(), pw.numel()))
        torch.cuda.synchronize()
        rows = order.cpu().numpy()[a:b].astype(np.int64)
        assert np.array_equal(pid.cpu().numpy().astype(np.int64), (5 + 2 * rows) if tab is None else (3 * rows + 1))
        assert prec.cpu().numpy().tobytes() == want_p[rows].tobytes()
        o = off.cpu().numpy(); v = var.cpu().numpy()
        assert np.array_equal(np.diff(o), (want_o[rows + 1] - want_o[rows]).astype(np.int64))
        for t in range(0, m, 17):
            i = rows[t]
            assert v[o[t]:o[t + 1]].tobytes() == want_v[int(want_o[i]):int(want_o[i + 1])].tobytes()
    d.close()


def test_rccl_transport_one_rank_talking_to_itself(gpu):
    """fmd_comm_rccl_*: librccl loaded on first use, a communicator of one rank, ncclAllGather and a grouped ncclSend + ncclRecv to
    itself through the fmd_comm_t function pointers; then a whole fmd_ovlp_dist_step over that communicator (world 1: every piece is
    packed straight into the table)."""
    import torch
    from fermi_amd import dist as fdist
    api = gpu
    L = api.lib()
    assert L.fmd_comm_rccl_version() > 0
    comm = fdist.RcclComm(api, None, 0, 1, 0)
    c = fdist.Comm.from_address(comm.ptr().value)
    assert c.rank == 0 and c.world == 1
    dev = torch.device("cuda", 0)
    a = torch.arange(1 << 20, dtype=torch.int32, device=dev); b = torch.zeros_like(a); g = torch.zeros_like(a)
    assert c.allgather(c.ctx, None, a.data_ptr(), g.data_ptr(), a.numel() * 4) == 0
    ops = (fdist.CommOp * 2)(fdist.CommOp(0, 0, a.data_ptr(), a.numel() * 4), fdist.CommOp(1, 0, b.data_ptr(), b.numel() * 4))
    assert c.exchange(c.ctx, None, 2, ops) == 0
    torch.cuda.synchronize()
    assert torch.equal(a, b) and torch.equal(a, g)
    N = 12000
    d = api.DevIndex.from_bwt(api.build_bwt(_reads(N, 0.003, False)))
    n_ids = int(d.mcnt[1])
    job = fdist.DistJob(api, d, comm, n_ids, 50, 100, 4, pieces=3, key_shard=1, root=0, host_table=0)
    st = job.step()
    assert st.key_shard == 0 and st.pieces == 3 and st.rows_computed == n_ids
    want_p, want_o, want_v = _expected(api, d, n_ids, 50, 100, 4)
    sample = np.arange(0, n_ids, 13)
    got_rec, got_var = _table_rows(api, job.table(), n_ids, 4, 200, sample)
    assert got_rec.tobytes() == want_p.tobytes()
    for i in sample:
        assert got_var[int(i)] == want_v[int(want_o[i]):int(want_o[i + 1])].tobytes()
    job.free(); comm.free(); d.close()


def _sink_worker(rank, world, port, N, err, ragged, key_shard, pieces, tmp, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import ctypes as C
    import hashlib
    import torch
    import torch.distributed as dist
    from fermi_amd import api, dist as fdist, hostlib
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    bwt = api.build_bwt(_reads(N, err, ragged))
    d = api.DevIndex.from_bwt(bwt)
    n_ids = int(d.mcnt[1])
    comm = fdist.TorchComm(api, dist, rank, world)
    root = hostlib.DistRoot(n_ids, 128) if rank == 0 else None
    job = fdist.DistJob(api, d, comm, n_ids, 50, 100, 4, pieces=pieces, key_shard=key_shard, root=0, host_table=2,
                        row_sink=root.sink if root else None, sink_ctx=root.ctx if root else None)
    st = job.step()
    assert st.pieces == pieces and (rank != 0 or st.on_host == 2)
    if rank == 0:
        assert root.rows() == n_ids
        assert api.lib().fmd_ovlp_dist_table(job.h, C.byref(fdist.DistTable())) == -2       # no table was kept
        nbytes = root.finish(50, d.h)               # rows beyond a capacity again and the open check_left edges: on this rank's GPU
        out = os.path.join(tmp, "dist.mag")
        root.walk(50, out)
        fmd = os.path.join(tmp, "r.fmd")
        hostlib.write_rld_from_bwt(bwt, fmd)
        one = os.path.join(tmp, "one.mag")
        hostlib.unitig(fmd, 50, one, devices=(0,))   # what `fermi-amd unitig -l50` prints from one process
        a, b = open(out, "rb").read(), open(one, "rb").read()
        q.put((0, hashlib.md5(a).hexdigest() == hashlib.md5(b).hexdigest() and len(a) > 1000, nbytes / n_ids))
        root.close()
    # a sink that turns rows down fails the step on EVERY rank with one code
    bad = hostlib.DistRoot(n_ids, 128) if rank == 0 else None
    job2 = fdist.DistJob(api, d, comm, n_ids, 50, 100, 4, pieces=pieces, key_shard=key_shard, root=0, host_table=2,
                         row_sink=bad.sink if bad else None, sink_ctx=None)              # (no context: fmdh_dist_root_sink returns -EINVAL on the first piece)
    rc = api.lib().fmd_ovlp_dist_step(job2.h, None, C.byref(job2.stats))
    torch.cuda.synchronize()
    q.put((10 + rank, int(rc), 0.0))
    dist.barrier()
    if bad:
        bad.close()
    job2.free(); job.free(); d.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,N,err,ragged,key_shard,pieces", [
    (3, 15000, 0.0, False, 0, 3),       # id shard, error-free reads
    (3, 15000, 0.01, True, 1, 4),       # key shard, reads with errors, ragged lengths, Ns: rows beyond max_nei = 4 are computed again on the root's GPU
    (3, 9000, 0.005, False, 0, 1),      # one piece: everything reaches the sink after the loop
    (1, 9000, 0.01, True, 0, 2),        # a world of one: the root's own pieces take the same way
])
def test_root_folds_arriving_pieces_into_the_rows_unitig_walks(gpu, tmp_path, world, N, err, ragged, key_shard, pieces):
    """the root of the N-process step kept the packed rows (125 bytes per id) where the CLI holds 44.5.  With host_table = 2 it keeps no
    table: every piece of every peer goes from its pinned landing buffers through cfg.row_sink (fmdh_dist_root_sink, libfmdhost) into the slim rows while
    the next piece is computed; fmdh_dist_root_finish + the walk then print the MAG -- the bytes `fermi-amd unitig -l50` prints from one process.  Three
    ranks share the GPU (gloo transport).  A sink that fails takes every rank out of the step with FMD_E_IO."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_sink_worker, args=(r, world, port, N, err, ragged, key_shard, pieces, str(tmp_path), q)) for r in range(world)]
    for p in ps:
        p.start()
    got = {}
    for _ in range(1 + world):
        k, v, x = q.get(timeout=300)
        got[k] = (v, x)
    for p in ps:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert got[0][0] is True, got
    assert got[0][1] < 70.0, got             # bytes per id at the root (packed rows: 125 and more)
    assert [got[10 + r][0] for r in range(world)] == [-4] * world, got
