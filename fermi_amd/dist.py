"""Multi-GPU plumbing: one process per GPU, full index replicated in every GPU's HBM, sequence ids
sharded with the reference's own start/step interleave (unitig.c:333, 398-399: worker j takes
i = j, j+step, ...), and ONE exchange: the final gather of the per-id records on rank 0
(torch.distributed; backend "nccl" is RCCL over xGMI on the GPU box, "gloo" in the CPU tests).

The gather is direct, not a ring: an all-gather of per-rank counts, then every peer sends its shard
straight to the root with one batched isend/irecv -- on MI355X each peer has its own xGMI link to
the root, so the transfer is per-link bound."""
import numpy as np


def shard_ids(n_ids, rank, world):
    """ids handled by `rank`: rank, rank+world, ... (start/step interleave)."""
    return np.arange(rank, n_ids, world, dtype=np.uint64)


def gather_rows(local, n_total, rank, world, dist, device=None, dst=0):
    """Gather row-sharded records (numpy structured or plain array, first axis = this rank's ids in
    shard_ids order) on `dst`, returned in global id order; other ranks get None."""
    import torch
    if world == 1:
        return local
    row_bytes = local.dtype.itemsize * int(np.prod(local.shape[1:], dtype=np.int64))
    flat = np.ascontiguousarray(local).view(np.uint8).reshape(-1)
    t = torch.from_numpy(flat.copy())
    if device is not None:
        t = t.to(device)
    cnt = torch.tensor([local.shape[0]], dtype=torch.int64, device=t.device)
    counts = [torch.zeros(1, dtype=torch.int64, device=t.device) for _ in range(world)]
    dist.all_gather(counts, cnt)
    counts = [int(c.item()) for c in counts]
    expect = [len(range(r, n_total, world)) for r in range(world)]
    assert counts == expect, (counts, expect)
    if rank == dst:
        bufs = [t if r == dst else torch.empty(counts[r] * row_bytes, dtype=torch.uint8, device=t.device) for r in range(world)]
        ops = [dist.P2POp(dist.irecv, bufs[r], r) for r in range(world) if r != dst and counts[r]]
        for w in (dist.batch_isend_irecv(ops) if ops else []):
            w.wait()
        out = np.empty((n_total,) + local.shape[1:], dtype=local.dtype)
        for r in range(world):
            part = bufs[r].cpu().numpy().view(local.dtype).reshape((counts[r],) + local.shape[1:])
            out[r::world] = part
        return out
    if counts[rank]:
        for w in dist.batch_isend_irecv([dist.P2POp(dist.isend, t, dst)]):
            w.wait()
    return None
