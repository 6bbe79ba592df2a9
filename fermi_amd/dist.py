"""Multi-GPU plumbing: one process per GPU, full index replicated in every GPU's HBM, sequence ids
sharded with the reference's own start/step interleave (unitig.c:333, 398-399: worker j takes
i = j, j+step, ...), and ONE exchange: the final gather of the packed per-id rows on rank 0
(torch.distributed; backend "nccl" is RCCL over xGMI on the GPU box, "gloo" in the CPU tests).

What travels is the packed form of include/fmd_hip.h (fmd_ovlp_pack_dev): per rank a record array
(64 B per id), an offset array and the variable part (neighbours + 2-bit bases).  The gather is direct, not
a ring: an all-gather of the per-rank byte counts, then every peer sends its three arrays straight to the
root in one batched isend/irecv -- on MI355X each peer has its own xGMI link to the root, so the transfer is
per-link bound.  Buffers stay where they are: device tensors under nccl (no host bounce), host
tensors under gloo.  The receive buffers are allocated once (PackedGather) and live in pinned host memory, filled peer
by peer, when the root's HBM cannot hold them beside the index (configs[4]).  Row i of the global table = arrays of rank i % world, index i // world: rank 0 never
re-interleaves (fermi_amd/host/unitig_walk.c addresses shards the same way)."""
import numpy as np


def shard_ids(n_ids, rank, world):
    """ids handled by `rank`: rank, rank+world, ... (start/step interleave)."""
    return np.arange(rank, n_ids, world, dtype=np.uint64)


def shard_size(n_ids, rank, world):
    return len(range(rank, n_ids, world))


class Watchdog:
    """`with Watchdog(120, "what"):` -- if the block is still running after `seconds`, say so on stderr and end the process (a hung
    collective never returns to Python; the driver should see a dead rank with a reason, not a silent stall)."""

    def __init__(self, seconds, what):
        self.seconds, self.what, self.t = seconds, what, None

    def __enter__(self):
        import os, sys, threading

        def fire():
            print("[fermi_amd.dist] %s did not finish within %d s -- giving up (rank %s)" % (self.what, self.seconds, os.environ.get("RANK", "?")), file=sys.stderr, flush=True)
            os._exit(3)
        if self.seconds and self.seconds > 0:
            self.t = threading.Timer(self.seconds, fire)
            self.t.daemon = True
            self.t.start()
        return self

    def __exit__(self, *exc):
        if self.t is not None:
            self.t.cancel()
        return False


def describe_fabric(torch, dist, rank, world):
    """One stderr line per job about what carries the exchange (rank 0 only): backend, RCCL version, peer access between the GPUs."""
    import sys
    if rank != 0:
        return
    try:
        be = dist.get_backend()
        ver = ".".join(str(v) for v in torch.cuda.nccl.version()) if be == "nccl" and torch.cuda.is_available() else "-"
        n_dev = torch.cuda.device_count() if torch.cuda.is_available() else 0
        peers = sum(1 for a in range(n_dev) for b in range(n_dev) if a != b and torch.cuda.can_device_access_peer(a, b)) if n_dev > 1 else 0
        print("[fermi_amd.dist] world %d, backend %s (RCCL %s), %d visible GPUs, %d of %d ordered GPU pairs with peer access"
              % (world, be, ver, n_dev, peers, n_dev * (n_dev - 1)), file=sys.stderr, flush=True)
    except Exception as ex:   # diagnostics must never take the job down
        print("[fermi_amd.dist] fabric description unavailable: %r" % (ex,), file=sys.stderr, flush=True)


class PackedGather:
    """The one exchange of the N > 1 path, with its receive buffers kept between steps.

    Root side, per peer r: record array (64 B per row), offsets (8 B per row + 8) and the variable part, whose size is learnt from
    the all-gather of byte counts in the first step (the same data gives the same sizes in every step; a larger size re-allocates).
    Where the peers' arrays fit in the root's free HBM beside what is resident they are received device to device in ONE batched
    isend/irecv (`path` = "device"); where they do not -- BASELINE configs[4]: 7 x 1.75*10^8 rows x ~133 B = 163 GB next to a
    94.5 GB index and the work area -- peer after peer through one device staging buffer into pinned host memory (`path` =
    "host-rounds"), which is also where `fermi-amd unitig -g` keeps its table.  If the batched point-to-point call itself raises
    (a transport that lacks it), the arrays are padded to the largest rank's and all-gathered (`path` = "all-gather")."""

    def __init__(self, torch, dist, n_ids, rank, world, dst=0, force_path=None, timeout_s=120):
        self.torch, self.dist, self.n_ids, self.rank, self.world, self.dst = torch, dist, n_ids, rank, world, dst
        self.force_path, self.timeout_s = force_path, timeout_s
        self.path = None
        self.bufs = None          # root: list over ranks of [prec, off, var] receive buffers (None for the root's own slot)
        self.stage = None         # root, host-rounds: device staging buffers of one peer
        self.sizes = None

    # -- helpers
    def _alloc_root(self, sizes, like, pinned_host):
        torch = self.torch
        bufs = []
        for r in range(self.world):
            if r == self.dst:
                bufs.append(None)
                continue
            n_r = shard_size(self.n_ids, r, self.world)
            if pinned_host:
                pin = like.is_cuda
                mk = lambda n, dt: torch.empty(n, dtype=dt, pin_memory=pin)
            else:
                mk = lambda n, dt: torch.empty(n, dtype=dt, device=like.device)
            bufs.append([mk(n_r * 64, torch.uint8), mk(n_r + 1, torch.int64), mk(sizes[r], torch.uint8)])
        return bufs

    def _choose(self, sizes, like):
        if self.force_path:
            return self.force_path
        if not like.is_cuda:
            return "device"       # host tensors (gloo): "device" = where the arrays live
        need = sum(shard_size(self.n_ids, r, self.world) * 72 + 8 + sizes[r] for r in range(self.world) if r != self.dst)
        free_b, _ = self.torch.cuda.mem_get_info(like.device)
        return "device" if need + (4 << 30) <= free_b else "host-rounds"

    def __call__(self, prec, off, var):
        """prec: uint8 [64 * n_r], off: int64 [n_r + 1], var: uint8 [>= off[-1]] of this rank's shard.
        -> on the root: list over ranks of (prec, off, var) with var trimmed to its used bytes; None elsewhere."""
        torch, dist, world, rank, dst = self.torch, self.dist, self.world, self.rank, self.dst
        if world == 1:
            self.path = "none"
            return [(prec, off, var[: int(off[-1].item())])]
        home = prec.device
        if dist.get_backend() != "nccl" and prec.is_cuda:
            # gloo moves host memory only: the one-GPU test form of the N > 1 path (bench.py FMD_BENCH_BACKEND=gloo) bounces
            # through the host here; under nccl (RCCL) the device tensors below go peer to peer over xGMI as they are
            prec, off, var = prec.cpu(), off.cpu(), var[: int(off[-1].item())].cpu()
        with Watchdog(self.timeout_s, "the gather of the packed overlap records"):
            tot = off[-1:].clone()
            sizes_t = [torch.zeros(1, dtype=torch.int64, device=off.device) for _ in range(world)]
            dist.all_gather(sizes_t, tot)
            sizes = [int(s.item()) for s in sizes_t]
            mine = sizes[rank]
            if self.path is None:
                # every rank takes the same decision: the root's choice (its free memory) is broadcast, and a transport whose
                # batched point-to-point call does not work on ANY rank (probed once with 8 bytes) sends everybody to the all-gather
                flag = torch.zeros(1, dtype=torch.int64, device=off.device)
                if rank == dst:
                    flag[0] = {"device": 0, "host-rounds": 1, "all-gather": 2}[self._choose(sizes, prec)]
                dist.broadcast(flag, dst)
                self.path = ("device", "host-rounds", "all-gather")[int(flag.item())]
                if self.path != "all-gather":
                    ok = torch.ones(1, dtype=torch.int64, device=off.device)
                    try:
                        self._probe(off.device)
                    except Exception as ex:
                        import sys
                        print("[fermi_amd.dist] batched isend/irecv failed on rank %d (%r)" % (rank, ex), file=sys.stderr, flush=True)
                        ok[0] = 0
                    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
                    if int(ok.item()) == 0:
                        if self.force_path in ("device", "host-rounds"):
                            raise RuntimeError("point-to-point transport unavailable")
                        self.path = "all-gather"
            if self.path != "all-gather":
                return self._p2p(prec, off, var, sizes, mine, home)
            return self._all_gather(prec, off, var, sizes, mine, home)

    def _probe(self, device):
        """8 bytes from every peer to the root through the call the gather uses."""
        torch, dist = self.torch, self.dist
        if self.rank == self.dst:
            t = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(self.world)]
            ops = [dist.P2POp(dist.irecv, t[r], r) for r in range(self.world) if r != self.dst]
        else:
            ops = [dist.P2POp(dist.isend, torch.full((1,), self.rank, dtype=torch.int64, device=device), self.dst)]
        if getattr(self, "_break_p2p", False):      # (tests: a transport without the batched call)
            raise RuntimeError("batch_isend_irecv disabled")
        for w in dist.batch_isend_irecv(ops):
            w.wait()

    def _p2p(self, prec, off, var, sizes, mine, home):
        torch, dist, world, rank, dst = self.torch, self.dist, self.world, self.rank, self.dst
        if rank != dst:
            ops = []
            if prec.numel():
                ops += [dist.P2POp(dist.isend, prec, dst), dist.P2POp(dist.isend, off, dst)]
            if mine:
                ops.append(dist.P2POp(dist.isend, var[:mine], dst))
            for w in (dist.batch_isend_irecv(ops) if ops else []):
                w.wait()
            return None
        host_rounds = self.path == "host-rounds"
        if self.bufs is None or any(self.bufs[r] is not None and self.bufs[r][2].numel() < sizes[r] for r in range(world)):
            self.bufs = self._alloc_root(sizes, prec, pinned_host=host_rounds)   # first step (or the sizes grew): allocated once, outside later steps
            if host_rounds:
                n_max = max(shard_size(self.n_ids, r, world) for r in range(world) if r != dst)
                s_max = max(sizes[r] for r in range(world) if r != dst)
                self.stage = [torch.empty(n_max * 64, dtype=torch.uint8, device=prec.device), torch.empty(n_max + 1, dtype=torch.int64, device=prec.device),
                              torch.empty(max(s_max, 1), dtype=torch.uint8, device=prec.device)]
        out = [None] * world
        out[dst] = (prec, off, var[: sizes[dst]])
        if not host_rounds:
            ops = []
            for r in range(world):
                if r == dst:
                    continue
                n_r = shard_size(self.n_ids, r, world)
                b = self.bufs[r]
                if n_r:
                    ops += [dist.P2POp(dist.irecv, b[0], r), dist.P2POp(dist.irecv, b[1], r)]
                if sizes[r]:
                    ops.append(dist.P2POp(dist.irecv, b[2][: sizes[r]], r))
                out[r] = (b[0], b[1], b[2][: sizes[r]])
            for w in (dist.batch_isend_irecv(ops) if ops else []):
                w.wait()
            if out[dst][0].device != home:
                out = [tuple(t.to(home) for t in b) for b in out]
            return out
        for r in range(world):     # one peer at a time through the staging buffers into pinned host memory
            if r == dst:
                continue
            n_r = shard_size(self.n_ids, r, world)
            ops = []
            if n_r:
                ops += [dist.P2POp(dist.irecv, self.stage[0][: n_r * 64], r), dist.P2POp(dist.irecv, self.stage[1][: n_r + 1], r)]
            if sizes[r]:
                ops.append(dist.P2POp(dist.irecv, self.stage[2][: sizes[r]], r))
            for w in (dist.batch_isend_irecv(ops) if ops else []):
                w.wait()
            b = self.bufs[r]
            b[0].copy_(self.stage[0][: n_r * 64]); b[1].copy_(self.stage[1][: n_r + 1]); b[2][: sizes[r]].copy_(self.stage[2][: sizes[r]])
            out[r] = (b[0], b[1], b[2][: sizes[r]])
        if prec.is_cuda:
            torch.cuda.synchronize()
        return out

    def _all_gather(self, prec, off, var, sizes, mine, home):
        torch, dist, world, rank, dst = self.torch, self.dist, self.world, self.rank, self.dst
        n_max = max(shard_size(self.n_ids, r, world) for r in range(world))
        s_max = max(max(sizes), 1)
        pad = lambda t, n: t if t.numel() == n else torch.cat([t, torch.zeros(n - t.numel(), dtype=t.dtype, device=t.device)])
        mine_t = [pad(prec, n_max * 64), pad(off, n_max + 1), pad(var[:mine], s_max)]
        got = []
        for t in mine_t:
            lst = [torch.empty_like(t) for _ in range(world)]
            dist.all_gather(lst, t)
            got.append(lst)
        if rank != dst:
            return None
        out = []
        for r in range(world):
            n_r = shard_size(self.n_ids, r, world)
            out.append((got[0][r][: n_r * 64], got[1][r][: n_r + 1], got[2][r][: sizes[r]]))
        if out[0][0].device != home:
            out = [tuple(t.to(home) for t in b) for b in out]
        return out


def gather_packed(prec, off, var, n_ids, rank, world, dist, dst=0, force_path=None):
    """One-shot form of PackedGather (tests; bench.py keeps a PackedGather across its steps)."""
    import torch
    return PackedGather(torch, dist, n_ids, rank, world, dst, force_path)(prec, off, var)


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
        p, lens, vb = packed_rows(torch, gathered[r], rows.to(gathered[r][0].device))   # (the arrays may live in pinned host memory)
        p, lens, vb = p.to(job.dev), lens.to(job.dev), vb.to(job.dev)
        want_lens = off[1:] - off[:-1]
        if not (torch.equal(p.reshape(-1), prec) and torch.equal(lens, want_lens) and torch.equal(vb, var[: int(off[-1].item())])):
            return "MISMATCH (rows of rank %d)" % r
        n_checked += m
    return "ok: %d rows computed by ranks 1..%d recomputed on rank 0, packed bytes identical" % (n_checked, world - 1)
