"""CPU, world_size 2, gloo: the N>1 path -- id sharding (start/step interleave) and the single
gather of per-id records on rank 0 -- with the record dtype the GPU path produces."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from fermi_amd import api
from fermi_amd import dist as fdist


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _fake_records(ids):
    rec = np.zeros(len(ids), dtype=api.OVLP_DT)
    rec["rank"] = ids * 7 + 1
    rec["k"][:, 0] = ids; rec["k"][:, 1] = ids ^ 1; rec["k"][:, 2] = 1
    rec["len"] = 100; rec["n_nei"] = (ids % 3).astype(np.int32)
    nei = np.zeros((len(ids), 4), dtype=api.INTV_DT)
    nei["x"][:, 0, 0] = ids + 5; nei["info"][:, 0] = 80 + ids % 17
    return rec, nei


def _worker(rank, world, port, n_ids, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ids = fdist.shard_ids(n_ids, rank, world)
    rec, nei = _fake_records(ids)
    g_rec = fdist.gather_rows(rec, n_ids, rank, world, dist)
    g_nei = fdist.gather_rows(nei, n_ids, rank, world, dist)
    if rank == 0:
        want_rec, want_nei = _fake_records(np.arange(n_ids, dtype=np.uint64))
        q.put(bool(g_rec.tobytes() == want_rec.tobytes() and g_nei.tobytes() == want_nei.tobytes()))
    else:
        assert g_rec is None and g_nei is None
    dist.barrier()
    dist.destroy_process_group()


def test_shard_ids_is_start_step_interleave():
    assert list(fdist.shard_ids(10, 1, 4)) == [1, 5, 9]
    allids = np.sort(np.concatenate([fdist.shard_ids(1001, r, 8) for r in range(8)]))
    assert np.array_equal(allids, np.arange(1001, dtype=np.uint64))


def test_gather_records_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    n_ids = 1001  # odd: ranks hold different counts
    ps = [ctx.Process(target=_worker, args=(r, 2, port, n_ids, q)) for r in range(2)]
    for p in ps:
        p.start()
    ok = q.get(timeout=120)
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert ok
