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

    def restart(self, seconds=None):
        """Start the clock again (after a step that is slow but not a stall: pinning the root's receive buffers)."""
        if self.t is not None:
            self.t.cancel()
            self.t = None
        if seconds is not None:
            self.seconds = seconds
        self.__enter__()


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
        with Watchdog(self.timeout_s, "the gather of the packed overlap records") as wd:
            self._wd = wd
            tot = off[-1:].clone()
            sizes_t = [torch.zeros(1, dtype=torch.int64, device=off.device) for _ in range(world)]
            dist.all_gather(sizes_t, tot)
            sizes = [int(s.item()) for s in sizes_t]
            mine = sizes[rank]
            if self.timeout_s:
                wd.restart(self.timeout_s + int((sum(sizes) + 72 * self.n_ids) / 1e9) * (10 if self.bufs is None else 1))   # the first step also allocates (and may pin) the root's buffers
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
            # (pinning ~163 GB of host memory takes minutes and is not a stall: the clock starts again here, with room for the bytes to come --
            # the peers' clocks run on while they wait in their sends, so theirs is the same budget: timeout_s + 1 s per GB expected)
            getattr(self, "_wd", Watchdog(0, "")).restart(self.timeout_s + int(sum(sizes) / 1e9))
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


def gather_packed(prec, off, var, n_ids, rank, world, dist, dst=0, force_path=None, timeout_s=120):
    """One-shot form of PackedGather (tests; bench.py keeps a PackedGather across its steps).  timeout_s = 0: no watchdog."""
    import torch
    return PackedGather(torch, dist, n_ids, rank, world, dst, force_path, timeout_s)(prec, off, var)


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


# ======================================================================================================================
# Round 4: the N > 1 step behind the C ABI (fmd_ovlp_dist_*, fmd_comm_t; fermi_amd/csrc/fmd_ovlp_dist.hip) -- pass 2 in pieces
# whose rows leave under the compute of the next piece, and the optional key shard (one all-to-all of the parked strands).
# Here: the ctypes view of those entry points, the two transports a Python host can hand them (RCCL created from a unique id that
# travels through torch.distributed's store; torch.distributed itself through the fmd_comm_t callbacks -- gloo in the tests), and
# `piecewise_exchange` / `key_shard_rows`: the same exchange logic over plain tensors, the twin the CPU tests run on rows the
# oracle computed.
import ctypes as C


class CommOp(C.Structure):
    _fields_ = [("is_recv", C.c_int), ("peer", C.c_int), ("d_ptr", C.c_void_p), ("bytes", C.c_size_t)]


ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t)
EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(CommOp))
DESTROY_FN = C.CFUNCTYPE(None, C.c_void_p)


class Comm(C.Structure):       # fmd_comm_t
    _fields_ = [("rank", C.c_int), ("world", C.c_int), ("ctx", C.c_void_p), ("allgather", ALLGATHER_FN), ("exchange", EXCHANGE_FN), ("destroy", DESTROY_FN)]


class DistCfg(C.Structure):    # fmd_ovlp_dist_cfg_t
    _fields_ = [("n_ids", C.c_uint64), ("min_match", C.c_int), ("max_len", C.c_uint32), ("max_nei", C.c_uint32), ("pieces", C.c_uint32),
                ("key_shard", C.c_int), ("root", C.c_int), ("host_table", C.c_int), ("batch", C.c_size_t),
                ("row_sink", C.c_void_p), ("sink_ctx", C.c_void_p)]      # host_table = 2: the root hands every piece to row_sink(sink_ctx, ...) and keeps no table


class DistStats(C.Structure):  # fmd_ovlp_dist_stats_t
    _fields_ = [("head_ms", C.c_double), ("key_exchange_ms", C.c_double), ("tail_ms", C.c_double), ("last_piece_pack_send_ms", C.c_double),
                ("gather_exposed_ms", C.c_double), ("step_ms", C.c_double), ("rows_computed", C.c_uint64), ("rows_sent", C.c_uint64),
                ("bytes_sent", C.c_uint64), ("bytes_received", C.c_uint64), ("key_rows_sent", C.c_uint64),
                ("pieces", C.c_int), ("on_host", C.c_int), ("key_shard", C.c_int), ("two_pass", C.c_int)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class DistTable(C.Structure):  # fmd_ovlp_dist_table_t
    _fields_ = [("on_host", C.c_int), ("n_rows", C.c_uint64), ("prec", C.c_void_p), ("ids", C.c_void_p), ("vaddr", C.c_void_p), ("row_of_id", C.c_void_p)]


class TorchComm:
    """fmd_comm_t over torch.distributed through the two callbacks: the transport of the tests (gloo: device buffers bounce through
    host arrays; the call synchronises the stream).  The product's transport is RcclComm."""

    def __init__(self, api, dist, rank, world):
        import numpy as np
        import torch
        self.api, self.dist, self.np, self.torch = api, dist, np, torch
        self.lib = api.lib()
        self._ag = ALLGATHER_FN(self._allgather)
        self._ex = EXCHANGE_FN(self._exchange)
        self._de = DESTROY_FN(lambda ctx: None)
        self.c = Comm(rank, world, None, self._ag, self._ex, self._de)

    def ptr(self):
        return C.addressof(self.c)

    def _d2h(self, d_ptr, n):
        a = self.np.empty(n, dtype=self.np.uint8)
        if n:
            self.api.check(self.lib.fmd_memcpy_d2h(a.ctypes.data, d_ptr, n, None))
        return a

    def _h2d(self, d_ptr, a):
        if a.size:
            self.api.check(self.lib.fmd_memcpy_h2d(d_ptr, a.ctypes.data, a.size, None))

    def _sync(self, stream):
        self.torch.cuda.synchronize()

    def _allgather(self, ctx, stream, d_send, d_recv, nbytes):
        try:
            self._sync(stream)
            mine = self.torch.from_numpy(self._d2h(d_send, nbytes))
            got = [self.torch.empty(nbytes, dtype=self.torch.uint8) for _ in range(self.c.world)]
            self.dist.all_gather(got, mine)
            self._h2d(d_recv, self.torch.cat(got).numpy())
            self._sync(stream)
            return 0
        except Exception as ex:   # an exception must not cross the C frame
            import sys
            print("[fermi_amd.dist] TorchComm.allgather: %r" % (ex,), file=sys.stderr, flush=True)
            return -6

    def _exchange(self, ctx, stream, n_ops, ops):
        try:
            self._sync(stream)
            reqs, recvs = [], []
            for i in range(n_ops):
                o = ops[i]
                if o.bytes == 0:
                    continue
                if o.is_recv:
                    t = self.torch.empty(o.bytes, dtype=self.torch.uint8)
                    recvs.append((o.d_ptr, t))
                    reqs.append(self.dist.irecv(t, o.peer))
                else:
                    reqs.append(self.dist.isend(self.torch.from_numpy(self._d2h(o.d_ptr, o.bytes)), o.peer))
            for r in reqs:
                r.wait()
            for d_ptr, t in recvs:
                self._h2d(d_ptr, t.numpy())
            self._sync(stream)
            return 0
        except Exception as ex:
            import sys
            print("[fermi_amd.dist] TorchComm.exchange: %r" % (ex,), file=sys.stderr, flush=True)
            return -6

    def free(self):
        pass


class DryComm:
    """A stand-in fmd_comm_t of `world` ranks that carries nothing (both calls fail): what fmd_ovlp_dist_new needs to size and allocate one rank's buffers of an
    N-rank job on a box with one GPU (FMD_DIST_DRY=1 makes it say what it allocates, the root's table and arena included).  Never step() such a job."""

    def __init__(self, api, rank, world):
        self._ag = ALLGATHER_FN(lambda ctx, stream, d_send, d_recv, nbytes: -6)
        self._ex = EXCHANGE_FN(lambda ctx, stream, n_ops, ops: -6)
        self._de = DESTROY_FN(lambda ctx: None)
        self.c = Comm(rank, world, None, self._ag, self._ex, self._de)

    def ptr(self):
        return C.addressof(self.c)

    def free(self):
        pass


class RcclComm:
    """fmd_comm_rccl_init on this rank's device: rank 0 makes the unique id (ncclGetUniqueId), torch.distributed carries it."""

    def __init__(self, api, dist, rank, world, device):
        import torch
        self.lib = api.lib()
        ident = (C.c_uint8 * 128)()
        if rank == 0:
            api.check(self.lib.fmd_comm_rccl_unique_id(ident))
        box = [bytes(ident)]
        if world > 1:
            dist.broadcast_object_list(box, src=0)
        ident = (C.c_uint8 * 128).from_buffer_copy(box[0])
        h = C.c_void_p()
        api.check(self.lib.fmd_comm_rccl_init(device, rank, world, ident, C.byref(h)))
        self.h = h
        if world > 1:
            torch.cuda.synchronize()

    def ptr(self):
        return self.h

    def free(self):
        if self.h:
            self.lib.fmd_comm_free(self.h)
            self.h = None


class DistJob:
    """fmd_ovlp_dist_t: one pass of overlap discovery over ids 0 .. n_ids-1 on `world` GPUs per step()."""

    def __init__(self, api, index, comm, n_ids, min_match, max_len, max_nei=4, pieces=0, key_shard=0, root=0, host_table=-1, batch=0, row_sink=None, sink_ctx=None):
        """row_sink / sink_ctx: with host_table = 2, the address of a C function (fmd_ovlp_dist_cfg_t.row_sink: e.g. fmdh_dist_root_sink of libfmdhost) and its context"""
        self.api, self.lib, self.comm = api, api.lib(), comm
        self.cfg = DistCfg(n_ids, min_match, max_len, max_nei, pieces, key_shard, root, host_table, batch, row_sink, sink_ctx)
        h = C.c_void_p()
        api.check(self.lib.fmd_ovlp_dist_new(index.h, comm.ptr(), C.byref(self.cfg), C.byref(h)))
        self.h = h
        self.stats = DistStats()

    def step(self, stream=None):
        self.api.check(self.lib.fmd_ovlp_dist_step(self.h, stream, C.byref(self.stats)))
        return self.stats

    def table(self):
        t = DistTable()
        self.api.check(self.lib.fmd_ovlp_dist_table(self.h, C.byref(t)))
        return t

    def local(self):
        n, ids, rec, nei, seq, stride = C.c_uint64(), C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_uint32()
        self.api.check(self.lib.fmd_ovlp_dist_local(self.h, C.byref(n), C.byref(ids), C.byref(rec), C.byref(nei), C.byref(seq), C.byref(stride)))
        return n.value, ids.value, rec.value, nei.value, seq.value, stride.value

    def free(self):
        if self.h:
            self.lib.fmd_ovlp_dist_free(self.h)
            self.h = None


# ---- the same exchange logic over plain tensors (the twin the CPU tests run, gloo, on rows the oracle computed) ------------------
def piece_begin(rows, p, pieces):
    """first row of piece p of `pieces`: the pieces shrink P : P-1 : ... : 1 (fmd_ovlp_dist.hip), the last one -- whose transfer nothing hides -- is the smallest."""
    return rows * (p * (2 * pieces - p + 1)) // (pieces * (pieces + 1))


def local_quantiles(keys_sorted, world):
    """the p / W quantiles (p = 0 .. W-1; 0 for p = 0) of this rank's regular keys (k_ks_quantiles)."""
    k = np.asarray(keys_sorted, dtype=np.uint64)
    n_reg = int(np.searchsorted(k, np.uint64(0xfffffffe), side="left"))
    return np.array([0 if p == 0 or n_reg == 0 else int(k[n_reg * p // world]) for p in range(world)], dtype=np.int64)


def key_splitters(torch, dist, keys_sorted, world):
    """The boundaries of the W key ranges, the same on every rank: split[0] = 0, split[W] = 0xfffffffe, split[p] = the median over the ranks of
    their p / W quantiles (a minimizer is the smallest of 17 hashes: the keys crowd towards 0, equal ranges would not be equal shares)."""
    mine = torch.from_numpy(local_quantiles(keys_sorted, world))
    allq = [torch.zeros(world, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(allq, mine)
    allq = torch.stack(allq).numpy()
    split = [0]
    for p in range(1, world):
        m = int(np.sort(allq[:, p])[(world - 1) // 2])
        split.append(min(max(m, split[-1]), 0xfffffffe))
    split.append(0xfffffffe)
    return np.array(split, dtype=np.uint64)


def key_dest(keys, rank, split):
    """destination rank of every key (numpy array): rank p takes [split[p], split[p + 1]); the two special keys stay on `rank`."""
    k = np.asarray(keys, dtype=np.uint64)
    d = (np.searchsorted(split, k, side="right") - 1).astype(np.int64)
    d[k >= np.uint64(0xfffffffe)] = rank
    return d


def key_shard_rows(torch, dist, rank, world, park, keys):
    """The all-to-all of the parked strands: park [n, 64] uint8 rows in ascending key order, keys [n] (numpy uint32, ascending).
    -> (rows this rank owns afterwards [m, 64] in arrival order: rank 0's, rank 1's, ..., then its own special rows; counts matrix; the splitters)."""
    split = key_splitters(torch, dist, keys, world)
    dest = key_dest(keys, rank, split)
    special = np.asarray(keys, dtype=np.uint64) >= np.uint64(0xfffffffe)
    counts = np.array([int(((dest == q) & ~special).sum()) for q in range(world)] + [int(special.sum())], dtype=np.int64)
    mat = [torch.zeros(world + 1, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(mat, torch.from_numpy(counts))
    mat = torch.stack(mat).numpy()
    soff = np.concatenate([[0], np.cumsum(counts)])
    reqs, got = [], [None] * world
    for q in range(world):
        seg = park[int(soff[q]):int(soff[q + 1])]
        if q == rank:
            got[q] = seg.clone()
            continue
        if len(seg):
            reqs.append(dist.isend(seg.contiguous(), q))
        rc = int(mat[q, rank])
        got[q] = torch.empty((rc, park.shape[1]), dtype=park.dtype)
        if rc:
            reqs.append(dist.irecv(got[q], q))
    for r in reqs:
        r.wait()
    got.append(park[int(soff[world]):int(soff[world + 1])].clone())
    return torch.cat(got), mat, split


class DistStepFailed(RuntimeError):
    """A rank failed inside the step; EVERY rank raises this, with the same (rank, code, piece), before anything further is posted."""

    def __init__(self, rank, code, piece):
        super().__init__("rank %d reports code %d in piece %d: every rank leaves the step" % (rank, code, piece))
        self.failed_rank, self.code, self.piece = rank, code, piece


def piecewise_exchange(torch, dist, rank, world, root, pieces, rows_of_rank, my_pieces, sink=None):
    """The gather in pieces: my_pieces = list over p of (pid int32 [np], prec uint8 [np * 64], off int64 [np + 1], var uint8 [bytes]) of
    this rank's rows piece_begin(rows, p) .. piece_begin(rows, p + 1) in its computing order -- or an int (a negative status code): this
    rank FAILED on that piece (and on every later one).  Per piece: an all-gather of (rows, variable bytes, status), then every peer's four
    arrays straight to the root.  The status word is how a failure reaches everybody (fmd_ovlp_dist_step, fmd_ovlp_dist.hip: the same three
    words): all ranks read the same gathered words and raise DistStepFailed for the first failing rank BEFORE any send or receive of the piece.
    -> root: dict id -> (record bytes, variable-part bytes) of every row of the job; others: None.
    sink (the same on every rank: a callable at the root, anything true elsewhere): the root keeps NO table -- every peer's piece goes to
    sink(ids uint32 [n], prec uint8 [n * 64], off uint64 [n + 1], var uint8) as it has arrived (fmd_ovlp_dist_cfg_t.host_table = 2 / row_sink), a piece behind
    the exchange as the C step does it; a sink that raises is a failure of the root (FMD_E_IO = -4) and reaches everybody with the next status word -- one
    more is exchanged after the last piece."""
    table = {} if rank == root and not sink else None
    failed = 0
    waiting = []          # root with a sink: the piece that has arrived and not been handed on yet

    def hand_on():
        nonlocal failed
        while waiting and not failed:
            qpid, qprec, qoff, qvar = waiting.pop(0)
            try:
                sink(qpid.numpy().view(np.uint32), qprec.numpy(), qoff.numpy().astype(np.uint64), qvar.numpy())
            except Exception:
                failed = -4
    for p in range(pieces):
        if rank == root and sink:
            hand_on()                                    # piece p - 1, before this rank says how it fares
        if not isinstance(my_pieces[p], tuple):
            failed = failed or int(my_pieces[p])
        if failed:
            pid = prec = off = var = None
            mine = torch.tensor([0, 0, failed], dtype=torch.int64)
        else:
            pid, prec, off, var = my_pieces[p]
            mine = torch.tensor([len(pid), int(off[-1]), 0], dtype=torch.int64)
        sizes = [torch.zeros(3, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(sizes, mine)
        for q in range(world):
            if int(sizes[q][2]):
                raise DistStepFailed(q, int(sizes[q][2]), p)
        for q in range(world):   # every rank checks every count: the same verdict everywhere
            nq = int(sizes[q][0])
            assert nq == piece_begin(rows_of_rank[q], p + 1, pieces) - piece_begin(rows_of_rank[q], p, pieces), "rank %d, piece %d: %d rows" % (q, p, nq)
        if rank != root:
            ops = []
            if len(pid):
                ops = [dist.P2POp(dist.isend, pid, root), dist.P2POp(dist.isend, prec, root), dist.P2POp(dist.isend, off, root)]
                if int(off[-1]):
                    ops.append(dist.P2POp(dist.isend, var[: int(off[-1])], root))
            for w in (dist.batch_isend_irecv(ops) if ops else []):
                w.wait()
            continue
        got, ops = {}, []
        for q in range(world):
            nq, vb = int(sizes[q][0]), int(sizes[q][1])
            if q == root:
                got[q] = (pid, prec, off, var[:vb])
                continue
            b = (torch.empty(nq, dtype=torch.int32), torch.empty(nq * 64, dtype=torch.uint8), torch.empty(nq + 1, dtype=torch.int64), torch.empty(vb, dtype=torch.uint8))
            got[q] = b
            if nq:
                ops += [dist.P2POp(dist.irecv, b[0], q), dist.P2POp(dist.irecv, b[1], q), dist.P2POp(dist.irecv, b[2], q)]
                if vb:
                    ops.append(dist.P2POp(dist.irecv, b[3], q))
        for w in (dist.batch_isend_irecv(ops) if ops else []):
            w.wait()
        for q in range(world):   # placement: what k_place does at the root
            qpid, qprec, qoff, qvar = got[q]
            if sink:
                if len(qpid):
                    waiting.append((qpid, qprec, qoff, qvar))
                continue
            for t in range(len(qpid)):
                table[int(qpid[t])] = (qprec[t * 64:(t + 1) * 64].numpy().tobytes(), qvar[int(qoff[t]):int(qoff[t + 1])].numpy().tobytes())
    if sink:   # the last piece, and how the sink fared with it: everybody's code
        if rank == root:
            hand_on()
        words = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(words, torch.tensor([failed], dtype=torch.int64))
        for q in range(world):
            if int(words[q][0]):
                raise DistStepFailed(q, int(words[q][0]), pieces)
    return table


class _DevView:
    """A device pointer as something torch.as_tensor() accepts (the CUDA array interface, which torch's ROCm build reads too)."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False), "version": 2}


def table_tensors(torch, t, n_ids, device):
    """The root's table (fmd_ovlp_dist_table) as tensors: prec [n, 64] uint8, ids int32 [n], vaddr int64 [n], row_of_id int32 [n] --
    views of the job's own memory (device or pinned host), valid until the next step."""
    if t.on_host:
        mk = lambda p, nb: torch.from_numpy(np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), (nb,)))
    else:
        mk = lambda p, nb: torch.as_tensor(_DevView(p, nb), device=device)
    return (mk(t.prec, n_ids * 64).view(n_ids, 64), mk(t.ids, n_ids * 4).view(torch.int32), mk(t.vaddr, n_ids * 8).view(torch.int64),
            mk(t.row_of_id, n_ids * 4).view(torch.int32))


def check_table(torch, api, index, djob, n_ids, min_match, max_len, max_nei, device, sample=200_000, var_sample=2_000):
    """Rank 0, outside the timed region: every id present exactly once; a spread sample of ids recomputed here (fmd_ovlp_dev + fmd_ovlp_pack_dev)
    and compared with the rows that arrived -- records of `sample` ids byte for byte, variable parts of `var_sample` of them byte for byte."""
    lib = api.lib()
    t = djob.table()
    prec, ids, vaddr, row_of = table_tensors(torch, t, n_ids, device)
    ro = row_of.to(device).to(torch.int64)
    if int((ro < 0).sum().item()) or int((ro >= n_ids).sum().item()):
        return "MISMATCH (ids without a row)"
    if not torch.equal(ids.to(device)[ro].to(torch.int64), torch.arange(n_ids, dtype=torch.int64, device=device)):
        return "MISMATCH (row_of_id does not invert ids)"
    m = min(sample, n_ids)
    sel = (torch.arange(m, dtype=torch.int64, device=device) * (n_ids // m)).contiguous()
    stride = 2 * ((max_len + 3) // 4 * 4)
    rec = torch.zeros(m * 64, dtype=torch.uint8, device=device); nei = torch.zeros(m * max_nei * 32, dtype=torch.uint8, device=device)
    seq = torch.zeros(m * stride, dtype=torch.uint8, device=device)
    wb = max(lib.fmd_ovlp_work_bytes(m, max_len, min_match), lib.fmd_ovlp_pack_work_bytes(m))
    work = torch.empty(wb, dtype=torch.uint8, device=device)
    api.check(lib.fmd_ovlp_dev(index.h, None, m, sel.data_ptr(), min_match, max_len, max_nei, rec.data_ptr(), nei.data_ptr(), seq.data_ptr(), stride, work.data_ptr(), wb))
    cap = lib.fmd_ovlp_pack_max_bytes(m, max_nei, stride)
    wp = torch.empty(m * 64, dtype=torch.uint8, device=device); wo = torch.zeros(m + 1, dtype=torch.int64, device=device); wv = torch.empty(cap, dtype=torch.uint8, device=device)
    api.check(lib.fmd_ovlp_pack_dev(index.h, None, m, rec.data_ptr(), nei.data_ptr(), max_nei, seq.data_ptr(), stride, wp.data_ptr(), wo.data_ptr(), wv.data_ptr(), cap, work.data_ptr(), wb))
    torch.cuda.synchronize()
    rows = ro[sel]
    got = prec[rows.to(prec.device)].to(device)
    if not torch.equal(got.reshape(-1), wp):
        return "MISMATCH (records)"
    vs = min(var_sample, m)
    pick = (torch.arange(vs, dtype=torch.int64) * (m // vs)).tolist()
    wo_h, va_h = wo.cpu().numpy(), vaddr.to("cpu").numpy()
    rows_h = rows.cpu().numpy()
    for j in pick:
        nb = int(wo_h[j + 1] - wo_h[j])
        if nb == 0:
            continue
        addr = int(va_h[rows_h[j]])
        if t.on_host:
            g = bytes((C.c_uint8 * nb).from_address(addr))
        else:
            a = np.empty(nb, dtype=np.uint8)
            api.check(lib.fmd_memcpy_d2h(a.ctypes.data, C.c_void_p(addr), nb, None))
            g = a.tobytes()
        if g != wv[int(wo_h[j]):int(wo_h[j + 1])].cpu().numpy().tobytes():
            return "MISMATCH (variable part of id %d)" % int(sel[j].item())
    return "ok: all %d ids present once; %d ids spread over every rank's rows recomputed on rank 0: records identical, variable parts of %d of them identical" % (n_ids, m, vs)
