"""Multi-GPU plumbing: one process per GPU, full index replicated in every GPU's HBM, sequence ids
sharded with the reference's own start/step interleave (unitig.c:333, 398-399: worker j takes
i = j, j+step, ...), and ONE exchange: the final gather of the packed per-id rows on rank 0
(torch.distributed; backend "nccl" is RCCL over xGMI on the GPU box, "gloo" in the CPU tests).

What travels is the packed form of include/fmd_hip.h (fmd_ovlp_pack_dev): per rank a record array
(64 B per id), an offset array and the variable part (neighbours + 2-bit bases).  The gather is direct, not
a ring: an all-gather of the per-rank byte counts, then every peer sends its three arrays straight to the
root in one batched isend/irecv -- on MI355X each peer has its own xGMI link to the root, so the transfer is
per-link bound.  Buffers stay where they are: device tensors under nccl (no host bounce), host
tensors under gloo.  Row i of the global table = arrays of rank i % world, index i // world: rank 0 never
re-interleaves (fermi_amd/host/unitig_walk.c addresses shards the same way)."""
import numpy as np


def shard_ids(n_ids, rank, world):
    """ids handled by `rank`: rank, rank+world, ... (start/step interleave)."""
    return np.arange(rank, n_ids, world, dtype=np.uint64)


def shard_size(n_ids, rank, world):
    return len(range(rank, n_ids, world))


def gather_packed(prec, off, var, n_ids, rank, world, dist, dst=0):
    """prec: uint8 [64 * n_r], off: int64 [n_r + 1], var: uint8 [>= off[-1]] of this rank's shard.
    -> on `dst`: list over ranks of (prec, off, var) with var trimmed to its used bytes; None elsewhere."""
    import torch
    if world == 1:
        return [(prec, off, var[: int(off[-1].item())])]
    home = prec.device
    if dist.get_backend() != "nccl" and prec.is_cuda:
        # gloo moves host memory only: the one-GPU test form of the N > 1 path (bench.py FMD_BENCH_BACKEND=gloo) bounces
        # through the host here; under nccl (RCCL) the device tensors below go peer to peer over xGMI as they are
        prec, off, var = prec.cpu(), off.cpu(), var[: int(off[-1].item())].cpu()
    tot = off[-1:].clone()
    sizes = [torch.zeros(1, dtype=torch.int64, device=off.device) for _ in range(world)]
    dist.all_gather(sizes, tot)
    if rank == dst:
        sizes = [int(s.item()) for s in sizes]
        bufs, ops = [], []
        for r in range(world):
            if r == dst:
                bufs.append((prec, off, var[: sizes[r]]))
                continue
            n_r = shard_size(n_ids, r, world)
            b = (torch.empty(n_r * 64, dtype=torch.uint8, device=prec.device), torch.empty(n_r + 1, dtype=torch.int64, device=prec.device),
                 torch.empty(sizes[r], dtype=torch.uint8, device=prec.device))
            bufs.append(b)
            if n_r:
                ops += [dist.P2POp(dist.irecv, b[0], r), dist.P2POp(dist.irecv, b[1], r)]
            if sizes[r]:
                ops.append(dist.P2POp(dist.irecv, b[2], r))
        for w in (dist.batch_isend_irecv(ops) if ops else []):
            w.wait()
        if bufs[0][0].device != home:
            bufs = [tuple(t.to(home) for t in b) for b in bufs]
        return bufs
    mine = int(tot.item())
    ops = []
    if prec.numel():
        ops += [dist.P2POp(dist.isend, prec, dst), dist.P2POp(dist.isend, off, dst)]
    if mine:
        ops.append(dist.P2POp(dist.isend, var[:mine], dst))
    for w in (dist.batch_isend_irecv(ops) if ops else []):
        w.wait()
    return None


def packed_rows(torch, bufs, rows):
    """(prec rows [m, 64], lengths [m], bytes of the variable parts back to back) of the rows `rows` (int64 tensor,
    indices into one rank's arrays) -- what a pack of exactly those rows, in that order, would produce."""
    prec, off, var = bufs
    p = prec.view(-1, 64)[rows]
    beg, end = off[rows], off[rows + 1]
    lens = end - beg
    total = int(lens.sum().item())
    if total == 0:
        return p, lens, var[:0]
    starts_out = torch.cumsum(lens, 0) - lens
    idx = torch.repeat_interleave(beg - starts_out, lens) + torch.arange(total, dtype=torch.int64, device=var.device)
    return p, lens, var[idx]


def check_gathered(torch, api, job, gathered, n_ids, world, sample=200_000):
    """Rank 0, outside the timed region: recompute a spread sample of the ids that OTHER ranks computed and
    compare with the rows that arrived (records byte for byte, variable parts byte for byte)."""
    import ctypes as C
    lib = api.lib()
    n_checked = 0
    per_rank = max(1, sample // max(1, world - 1))
    for r in range(1, world):
        n_r = shard_size(n_ids, r, world)
        m = min(per_rank, n_r)
        if m == 0:
            continue
        rows = torch.arange(m, dtype=torch.int64, device=job.dev) * (n_r // m)
        ids = rows * world + r
        rec = torch.zeros(m * 64, dtype=torch.uint8, device=job.dev)
        nei = torch.zeros(m * job.max_nei * 32, dtype=torch.uint8, device=job.dev)
        seq = torch.zeros(m * job.stride, dtype=torch.uint8, device=job.dev)
        api.check(lib.fmd_ovlp_dev(job.index.h, job.sh, m, ids.data_ptr(), job.min_match, job.L, job.max_nei, rec.data_ptr(), nei.data_ptr(),
                                   seq.data_ptr(), job.stride, job.work.data_ptr(), job.wb))
        cap = lib.fmd_ovlp_pack_max_bytes(m, job.max_nei, job.stride)
        prec = torch.empty(m * 64, dtype=torch.uint8, device=job.dev)
        off = torch.zeros(m + 1, dtype=torch.int64, device=job.dev)
        var = torch.empty(cap, dtype=torch.uint8, device=job.dev)
        api.check(lib.fmd_ovlp_pack_dev(job.index.h, job.sh, m, rec.data_ptr(), nei.data_ptr(), job.max_nei, seq.data_ptr(), job.stride,
                                        prec.data_ptr(), off.data_ptr(), var.data_ptr(), cap, job.work.data_ptr(), job.wb))
        torch.cuda.synchronize()
        p, lens, vb = packed_rows(torch, gathered[r], rows)
        want_lens = off[1:] - off[:-1]
        if not (torch.equal(p.reshape(-1), prec) and torch.equal(lens, want_lens) and torch.equal(vb, var[: int(off[-1].item())])):
            return "MISMATCH (rows of rank %d)" % r
        n_checked += m
    return "ok: %d rows computed by ranks 1..%d recomputed on rank 0, packed bytes identical" % (n_checked, world - 1)
