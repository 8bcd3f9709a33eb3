"""Row-strip partition of one level across processes (one process per GPU).

The reference partitions a level's boxes over MPI ranks and couples them only through
LevelData::exchange (src/VCAMRNonLinearPoissonOp.cpp:47,124,304,405,692,751).  Here every
rank owns a strip of rows; the library calls the exchange hook wherever the reference
exchanges a field, the hook packs the strip's edge rows on the device
(suhmo_level_pack_rows), hands them to a transport, and unpacks the neighbour's rows into
the ghost rows.  Transports:
  * native RCCL (attach_rccl, the default on the "nccl" backend) -- pack, ncclSend/ncclRecv and unpack are
    enqueued by the library itself on the kernels' stream (suhmo_amd/csrc/suhmo_rccl.hip); Python only
    distributes the communicator id once.  No host round trip per exchange.
  * TorchDistTransport -- torch.distributed P2P (backend "nccl" = RCCL over xGMI on the
    GPU box, "gloo" in CPU tests); tensors are plain device buffers, plumbing only.
  * ThreadTransport    -- N "ranks" as threads of one process (tests on a 1-GPU box).
No collective is on the data path except the 8-byte MAX all-reduce of the residual norm.
"""
import ctypes as C
import os
import threading

from . import capi
from .capi import check


class StripExchanger:
    """Owns the hook closures for one level (keep a reference alive as long as the level)."""

    def __init__(self, level, transport, rank, world, periodic_y, peers=None):
        self.level, self.tr, self.rank, self.world = level, transport, rank, world
        self.lo = rank - 1 if rank > 0 else (world - 1 if periodic_y else None)
        self.hi = rank + 1 if rank < world - 1 else (0 if periodic_y else None)
        if peers is not None:          # rank / world index a sub-group (AMR patch strips): the transport addresses peers[k]
            self.lo = peers[self.lo] if self.lo is not None else None
            self.hi = peers[self.hi] if self.hi is not None else None
        self._bufs = {}
        self._ex = capi.EXCHANGE_FN(self._exchange)
        self._ar = capi.ALLREDUCE_FN(self._allreduce)
        check(capi.lib().suhmo_level_set_hooks(level.h, self._ex, self._ar, None))
        self._red = capi.REDUCE_FN(self._reduce)
        check(capi.lib().suhmo_level_set_reduce_hook(level.h, self._red))
        self.calls = self.gathers = 0
        # all-gather over the ranks of the level: with it the coarse multigrid depths are agglomerated (suhmo_agg.hip); the strips of
        # an AMR patch do not agglomerate (the library declines: a patch is not a whole level)
        self._ag = capi.ALLGATHER_FN(self._allgather)
        whole = getattr(level, "ny", 0) * world == getattr(level, "ny_global", -1)     # the `world` ranks of this exchanger hold the whole level in equal strips
        if whole and hasattr(transport, "allgather"):
            check(capi.lib().suhmo_level_set_allgather(level.h, self._ag, None))

    def _geom(self, depth):
        g = [C.c_int() for _ in range(5)]
        check(capi.lib().suhmo_level_halo_info(self.level.h, depth, *[C.byref(x) for x in g]))
        rows, _, _, nx, ny = [x.value for x in g]
        return min(rows, ny), nx

    def _exchange(self, user, L, depth, pfields, nfields, stream):
        try:
            fields = tuple(pfields[k] for k in range(nfields))
            rows, nx = self._geom(depth)
            n = rows * (nx + 1)                      # doubles per field and side
            key = (depth, fields)
            if key not in self._bufs:
                self._bufs[key] = [self.tr.alloc(n * nfields) for _ in range(4)]   # send lo/hi, recv lo/hi
            slo, shi, rlo, rhi = self._bufs[key]
            # L: the handle whose rows travel (the level itself, or its implicit gap-height solver: same strip, same hooks)
            lib, h, st = capi.lib(), (C.c_void_p(L) if L else self.level.h), C.c_void_p(stream)
            for q, f in enumerate(fields):
                off = q * n * 8
                if self.lo is not None:
                    check(lib.suhmo_level_pack_rows(h, depth, f, 0, rows, C.c_void_p(self.tr.ptr(slo) + off), st))
                if self.hi is not None:
                    check(lib.suhmo_level_pack_rows(h, depth, f, 1, rows, C.c_void_p(self.tr.ptr(shi) + off), st))
            self.tr.sendrecv(self.rank, self.lo, self.hi, slo, shi, rlo, rhi, key)
            for q, f in enumerate(fields):
                off = q * n * 8
                if self.lo is not None:
                    check(lib.suhmo_level_unpack_rows(h, depth, f, 0, rows, C.c_void_p(self.tr.ptr(rlo) + off), st))
                if self.hi is not None:
                    check(lib.suhmo_level_unpack_rows(h, depth, f, 1, rows, C.c_void_p(self.tr.ptr(rhi) + off), st))
            self.calls += 1
            return 0
        except Exception:  # never let an exception cross the C boundary
            import traceback
            traceback.print_exc()
            return -9

    def _allreduce(self, user, pval):
        try:
            pval[0] = self.tr.allreduce_max(self.rank, float(pval[0]))
            return 0
        except Exception:
            import traceback
            traceback.print_exc()
            return -9

    def _allgather(self, user, send, count, recv, stream):
        try:
            self.tr.allgather(self.rank, send, count, recv)
            self.gathers += 1
            return 0
        except Exception:
            import traceback
            traceback.print_exc()
            return -9

    def _reduce(self, user, pval, n, op):
        try:
            out = self.tr.allreduce(self.rank, [float(pval[k]) for k in range(n)], op)
            for k in range(n):
                pval[k] = out[k]
            return 0
        except Exception:
            import traceback
            traceback.print_exc()
            return -9

    def exchange_static(self):
        """halo rows of the caller-provided coefficient fields (depth 0)"""
        from . import level as lv
        fields = (lv.F_RHS, lv.F_ACOEF, lv.F_B, lv.F_PI, lv.F_ZB, lv.F_MASK, lv.F_BX, lv.F_BY)
        arr = (C.c_int * len(fields))(*fields)
        st = self.level.stream.value or 0
        rc = self._exchange(None, None, 0, arr, len(fields), st)
        if rc:
            raise capi.SuhmoError("exchange of the coefficient halos failed")


class HierGather:
    """The all-gather hook of a hierarchy whose level 0 is cut into rank strips (suhmo_hier_set_allgather) over a host
    transport; keep a reference alive as long as the hierarchy.  The native path is suhmo_hier_attach_rccl."""

    def __init__(self, hier, transport, rank):
        self.tr, self.rank, self.calls = transport, rank, 0
        self._fn = capi.ALLGATHER_FN(self._gather)
        check(capi.lib().suhmo_hier_set_allgather(hier.h, self._fn, None))

    def _gather(self, user, send, count, recv, stream):
        try:
            self.tr.allgather(self.rank, send, count, recv)
            self.calls += 1
            return 0
        except Exception:
            import traceback
            traceback.print_exc()
            return -9


class TorchDistTransport:
    """torch.distributed point-to-point; buffers are torch tensors on `device`."""

    def __init__(self, dist, device, group=None):
        import torch
        self.torch, self.dist, self.device, self.group = torch, dist, device, group
        self._ops = {}

    def alloc(self, n):
        return self.torch.empty(n, dtype=self.torch.float64, device=self.device)

    def ptr(self, t):
        return t.data_ptr()

    def sendrecv(self, rank, lo, hi, slo, shi, rlo, rhi, tag):
        ops = self._ops.get(tag) if tag is not None else None
        if ops is None:
            d, ops = self.dist, []
            # order matters when lo == hi (2 ranks, periodic): to-hi before to-lo, from-lo before from-hi
            if hi is not None:
                ops.append(d.P2POp(d.isend, shi, hi))
            if lo is not None:
                ops.append(d.P2POp(d.isend, slo, lo))
            if lo is not None:
                ops.append(d.P2POp(d.irecv, rlo, lo))
            if hi is not None:
                ops.append(d.P2POp(d.irecv, rhi, hi))
            if tag is not None:
                self._ops[tag] = ops      # buffers are fixed per (depth, fields): build the op list once
        if ops:
            # torch.distributed orders its transfers against torch's current stream only (nccl) or moves device tensors through the
            # host on streams of its own (gloo); the library's pack / unpack kernels run on the stream the hook was called with:
            # fence the device on both sides.  (The stream-ordered, overlapping transport is the native one, suhmo_rccl.hip.)
            fence = self.device.type == "cuda"
            if fence:
                self.torch.cuda.synchronize()
            for w in self.dist.batch_isend_irecv(ops):
                w.wait()
            if fence:
                self.torch.cuda.synchronize()

    def allreduce_max(self, rank, v):
        t = self.torch.tensor([v], dtype=self.torch.float64, device=self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX, group=self.group)
        return float(t.item())

    def allreduce(self, rank, vals, op):
        """n values, op 0 MAX / 1 SUM"""
        t = self.torch.tensor(vals, dtype=self.torch.float64, device=self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM if op else self.dist.ReduceOp.MAX, group=self.group)
        return [float(x) for x in t.tolist()]

    def allgather(self, rank, send, count, recv):
        """raw device pointers (the hierarchy's shadow refresh): staged through tensors the backend can move"""
        torch = self.torch
        key = ("ag", count)
        if key not in self._ops:
            world = self.dist.get_world_size(self.group)
            self._ops[key] = (torch.empty(count, dtype=torch.float64, device=self.device),
                              [torch.empty(count, dtype=torch.float64, device=self.device) for _ in range(world)])
            import ctypes
            self._hip = ctypes.CDLL("libamdhip64.so")
            self._hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
        ts, tr = self._ops[key]
        torch.cuda.synchronize()
        assert self._hip.hipMemcpy(ts.data_ptr(), send, count * 8, 3) == 0
        self.dist.all_gather(tr, ts, group=self.group)
        torch.cuda.synchronize()
        for r, t in enumerate(tr):
            assert self._hip.hipMemcpy(recv + r * count * 8, t.data_ptr(), count * 8, 3) == 0


class ThreadTransport:
    """N ranks = N threads of one process sharing one GPU (test harness).  Buffers are raw
    device allocations made through torch-free HIP calls of the library's own canvases: we
    borrow pinned staging through ctypes hipMalloc."""

    def __init__(self, world):
        self.world = world
        self.barrier = threading.Barrier(world, timeout=120)      # a rank that skips a collective breaks the barrier instead of hanging
        self.box = {}
        self.vals = [0.0] * world
        self._hip = C.CDLL("libamdhip64.so")
        self._hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        self._hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        self._hip.hipDeviceSynchronize.argtypes = []

    def alloc(self, n):
        p = C.c_void_p()
        assert self._hip.hipMalloc(C.byref(p), n * 8) == 0
        return (p.value, n)

    def ptr(self, b):
        return b[0]

    def sendrecv(self, rank, lo, hi, slo, shi, rlo, rhi, tag):
        self._hip.hipDeviceSynchronize()
        self.box[(rank, "lo")] = slo
        self.box[(rank, "hi")] = shi
        self.barrier.wait()
        if lo is not None:   # my lo ghosts <- lo neighbour's top rows
            src = self.box[(lo, "hi")]
            assert self._hip.hipMemcpy(C.c_void_p(rlo[0]), C.c_void_p(src[0]), rlo[1] * 8, 3) == 0
        if hi is not None:
            src = self.box[(hi, "lo")]
            assert self._hip.hipMemcpy(C.c_void_p(rhi[0]), C.c_void_p(src[0]), rhi[1] * 8, 3) == 0
        self._hip.hipDeviceSynchronize()
        self.barrier.wait()

    def allreduce_max(self, rank, v):
        self.vals[rank] = v
        self.barrier.wait()
        m = max(self.vals)
        self.barrier.wait()
        return m

    def allreduce(self, rank, vals, op):
        """n values, op 0 MAX / 1 SUM (in rank order: every rank gets the same bits)"""
        self.box[(rank, "red")] = list(vals)
        self.barrier.wait()
        cols = list(zip(*[self.box[(r, "red")] for r in range(self.world)]))
        out = [sum(c) if op else max(c) for c in cols]
        self.barrier.wait()
        return out

    def allgather(self, rank, send, count, recv):
        self._hip.hipDeviceSynchronize()
        self.box[(rank, "ag")] = send
        self.barrier.wait()
        for r in range(self.world):
            assert self._hip.hipMemcpy(C.c_void_p(recv + r * count * 8), C.c_void_p(self.box[(r, "ag")]), count * 8, 3) == 0
        self._hip.hipDeviceSynchronize()
        self.barrier.wait()


STATIC_FIELDS = (1, 2, 3, 4, 5, 6, 7, 8)    # RHS, ACOEF, B, PI, ZB, MASK, BX, BY: caller-provided at depth 0


def attach_rccl(level, rank, world, periodic_y=False, dist=None, unique_id=None):
    """Native transport.  The 128-byte communicator id is made on rank 0 and broadcast with
    torch.distributed (any backend) unless the caller passes it."""
    lib = capi.lib()
    # librccl: the one next to the HIP runtime this library is bound to (see suhmo_rccl_load); SUHMO_LIBRCCL overrides
    path = os.environ.get("SUHMO_LIBRCCL")
    check(lib.suhmo_rccl_load(path.encode() if path else None))
    if unique_id is None:
        idbuf = (C.c_char * 128)()
        if rank == 0:
            check(lib.suhmo_rccl_unique_id(C.cast(idbuf, C.c_void_p)))
        unique_id = bytes(idbuf)
        if world > 1:
            import torch                      # already loaded by the caller that made `dist`
            t = torch.tensor(list(unique_id), dtype=torch.uint8)
            dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
            t = t.to(dev)
            dist.broadcast(t, 0)
            unique_id = bytes(t.cpu().tolist())
    idarr = (C.c_char * 128).from_buffer_copy(unique_id)
    check(lib.suhmo_level_attach_rccl(level.h, C.cast(idarr, C.c_void_p), rank, world, int(periodic_y), level.stream))
    for f in STATIC_FIELDS:
        check(lib.suhmo_level_exchange(level.h, 0, f, level.stream))
    level._exchanger = "rccl"
    return level


def ipc_export(level):
    """step 1 of the peer-direct transport: this rank's arena; returns the 128-byte blob its neighbours need"""
    blob = (C.c_char * 128)()
    check(capi.lib().suhmo_level_ipc_export(level.h, C.cast(blob, C.c_void_p)))
    return bytes(blob)


def ipc_attach(level, rank, world, periodic_y, blobs):
    """step 2: blobs[r] = the blob of rank r (all ranks, or at least this rank's neighbours); halo exchanges from here on are peer-direct
    stores + flag words (suhmo_amd/csrc/suhmo_ipc.hip); reductions and all-gathers keep the hooks the level already has"""
    lo = rank - 1 if rank > 0 else (world - 1 if periodic_y else None)
    hi = rank + 1 if rank < world - 1 else (0 if periodic_y else None)
    keep = [(C.c_char * 128).from_buffer_copy(blobs[q]) if q is not None else None for q in (lo, hi)]
    check(capi.lib().suhmo_level_attach_ipc(level.h, rank, world, int(periodic_y),
                                            C.cast(keep[0], C.c_void_p) if keep[0] is not None else None,
                                            C.cast(keep[1], C.c_void_p) if keep[1] is not None else None))
    level._transport = "ipc (peer-direct stores into the neighbour's halo slots, hipIpcOpenMemHandle; reductions: %s)" % (
        "rccl" if getattr(level, "_exchanger", None) == "rccl" else "host hooks")


def _all_ok(dist, ok, dev):
    import torch
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    return int(flag.item()) == 1


def attach_ipc(level, dist, rank, world, periodic_y=False, probe=False):
    """COLLECTIVE: every rank exports its arena, the blobs travel by torch.distributed all_gather, every rank maps its neighbours'.
    probe: do not raise when a step fails on some rank or the first messages do not arrive intact -- every rank then goes back to the
    transport it had (suhmo_level_detach_ipc) and False is returned; three single-field messages of rank-coded values are checked row by row."""
    import torch
    import numpy as np
    from .level import F_CORR
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    err = None
    try:
        if probe and os.environ.get("SUHMO_IPC_PROBE_FAIL_RANK") == str(rank):      # fault injection (tests of the fallback)
            raise RuntimeError("failure injected on rank %d" % rank)
        mine = ipc_export(level)
    except Exception as e:
        if not probe:
            raise
        mine, err = bytes(128), e
    t = torch.tensor(list(mine), dtype=torch.uint8, device=dev)
    out = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(out, t)
    blobs = [bytes(o.cpu().tolist()) for o in out]
    if err is None:
        try:
            ipc_attach(level, rank, world, periodic_y, blobs)
        except Exception as e:
            if not probe:
                raise
            err = e
    if probe and not _all_ok(dist, err is None, dev):
        check(capi.lib().suhmo_level_detach_ipc(level.h))
        level._transport = None
        level._ipc_probe = "peer-direct transport not available: %s" % (err if err is not None else "another rank could not map its neighbours")
        dist.barrier()
        return False
    dist.barrier()                        # every arena is mapped before anybody stores into one
    if probe:
        lo = rank - 1 if rank > 0 else (world - 1 if periodic_y else None)
        hi = rank + 1 if rank < world - 1 else (0 if periodic_y else None)
        good = True
        try:
            for rnd in range(3):          # (three messages: both slots of the channel and the first acknowledgement are exercised)
                level.set_value(F_CORR, 1000.0 * (rnd + 1) + rank)
                check(capi.lib().suhmo_level_exchange(level.h, 0, F_CORR, level.stream))
                a = level.get(F_CORR, ghosted=True)
                if lo is not None:
                    good = good and bool(np.all(a[0, 1:-1] == 1000.0 * (rnd + 1) + lo))
                if hi is not None:
                    good = good and bool(np.all(a[-1, 1:-1] == 1000.0 * (rnd + 1) + hi))
        except Exception as e:
            good, err = False, e
        if not _all_ok(dist, good, dev):
            dist.barrier()                # nobody unmaps while a neighbour may still store
            check(capi.lib().suhmo_level_detach_ipc(level.h))
            level._transport = None
            level._ipc_probe = "peer-direct messages did not arrive intact on some rank (%s)" % (err if err is not None else "values differ")
            dist.barrier()
            return False
        level.set_value(F_CORR, 0.0)
        level._ipc_probe = "three probe messages arrived intact on every rank"
    return True


def attach(level, dist, rank, world, periodic_y=False):
    """bench.py / production entry: couple this rank's strip to its neighbours.  Backend "nccl":
    the native RCCL transport (SUHMO_TRANSPORT=torch forces the torch.distributed P2P one);
    other backends (gloo in CPU-side tests): torch.distributed P2P.  Returns the exchanger."""
    import torch
    if dist.get_backend() == "nccl" and os.environ.get("SUHMO_TRANSPORT", "auto") != "torch":
        # every rank must end up on the same transport: agree that librccl could be loaded everywhere BEFORE the
        # collective part (id broadcast, ncclCommInitRank) starts
        ok, err = 1, None
        try:
            path = os.environ.get("SUHMO_LIBRCCL")
            check(capi.lib().suhmo_rccl_load(path.encode() if path else None))
        except Exception as e:
            ok, err = 0, e
        flag = torch.tensor([ok], dtype=torch.int32, device=torch.device("cuda", torch.cuda.current_device()))
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 1:
            attach_rccl(level, rank, world, periodic_y, dist)
            # halo rows peer-direct, RCCL stays for reductions and all-gathers.  "ipc": insist; default ("auto"): probe -- map the neighbours,
            # send three checked messages -- and stay on RCCL's send / recv when any rank reports a failure; "rccl": do not try
            how = os.environ.get("SUHMO_TRANSPORT", "auto")
            if how == "ipc":
                attach_ipc(level, dist, rank, world, periodic_y)
            elif how == "auto":
                attach_ipc(level, dist, rank, world, periodic_y, probe=True)
            return level
        import sys
        print("suhmo_amd.multigpu: native RCCL transport unavailable (%s); falling back to torch.distributed P2P"
              % (err if err is not None else "another rank failed"), file=sys.stderr, flush=True)
    tr = TorchDistTransport(dist, torch.device("cuda", torch.cuda.current_device()))
    ex = StripExchanger(level, tr, rank, world, periodic_y)
    level._exchanger = ex
    ex.exchange_static()
    how = os.environ.get("SUHMO_TRANSPORT")                        # (gloo rehearsal: the halo rows peer-direct, reductions through the host hooks;
    if how in ("ipc", "ipc-probe"):                                #  ipc-probe: as the default on the nccl backend, with the fallback)
        attach_ipc(level, dist, rank, world, periodic_y, probe=how == "ipc-probe")
    return ex


def attach_hier(hier, dist, rank, world):
    """A hierarchy of box unions whose level 0 is this rank's strip: halo rows of level 0 as attach() does, and the all-gather of
    the coarse cells level 1 reads on the same transport (native RCCL: ncclAllGather on the strip's communicator).  COLLECTIVE."""
    base = hier.level[0][0]
    ex = attach(base, dist, rank, world, False)
    if ex == base or getattr(base, "_exchanger", None) == "rccl":
        check(capi.lib().suhmo_hier_attach_rccl(hier.h))
        hier._gather = "rccl"
    else:
        hier._gather = HierGather(hier, ex.tr, rank)
    return hier


def attach_amr(levels, ranges, dist, rank, world):
    """AMR hierarchy cut into rank strips: levels[l] is this rank's strip of level l (None where the patch does not reach
    this rank's slab), ranges[l] = ranks that hold a strip of level l (ascending rows, the same list on every rank).
    Every level gets its own communicator over exactly those ranks (native RCCL on the "nccl" backend, torch.distributed
    P2P otherwise).  COLLECTIVE over all ranks."""
    import torch
    native = dist.get_backend() == "nccl" and os.environ.get("SUHMO_TRANSPORT", "rccl") != "torch"
    lib = capi.lib()
    if native:
        path = os.environ.get("SUHMO_LIBRCCL")
        check(lib.suhmo_rccl_load(path.encode() if path else None))
    dev = torch.device("cuda", torch.cuda.current_device())       # halo staging buffers are device memory on every backend
    keep = []
    for l, part in enumerate(ranges):
        group = dist.new_group(part) if not native else None          # collective over the world, also for non-members
        if native:
            idbuf = (C.c_char * 128)()
            if rank == part[0]:
                check(lib.suhmo_rccl_unique_id(C.cast(idbuf, C.c_void_p)))
            t = torch.tensor(list(bytes(idbuf)), dtype=torch.uint8).to(dev)
            dist.broadcast(t, part[0])
            if levels[l] is not None:
                attach_rccl(levels[l], part.index(rank), len(part), False, dist, unique_id=bytes(t.cpu().tolist()))
        elif levels[l] is not None:
            tr = TorchDistTransport(dist, dev, group)
            ex = StripExchanger(levels[l], tr, part.index(rank), len(part), False, peers=part)
            ex.exchange_static()
            levels[l]._exchanger = ex
            keep.append(ex)
    return keep
