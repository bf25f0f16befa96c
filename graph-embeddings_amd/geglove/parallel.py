"""Multi-GPU row sharding of the trainer: one rank per GPU.

GPU g owns focus rows / fBias / their gradSq for a contiguous row block and every nonzero whose
i falls in it (BookmarkColoring output is already grouped by i).  The context factors (context,
cBias, gradSqContext, gradSqCBias) are replicated, updated locally Hogwild, and reconciled by a
periodic all-reduce of the per-rank deltas over RCCL/xGMI.

The exchange itself -- the merge rule, the device kernels, RCCL -- lives behind the C ABI (ge_sync_*, include/geglove.h,
csrc/sync.hip), where a Java or C++ host reaches it too; ContextSync below is the thin Python caller.  Transports:
  "rccl"   the library opens RCCL itself; the 128-byte id is created on rank 0 and broadcast through torch.distributed
           (any backend) or passed in;
  "torch"  the library calls back into torch.distributed for every buffer (tests: two ranks on ONE GPU over gloo, which RCCL
           refuses).
The reference has no multi-device semantics (single JVM); parity of this path is statistical
(cost history vs. a 1-GPU run), stated in DESIGN.md.
"""
import ctypes as C

import numpy as np

from . import capi


def shard_rows(vocab_size, world, rank):
    """Contiguous row block [begin, end) of `rank`; the first V % world ranks get one extra row."""
    base, rem = divmod(int(vocab_size), int(world))
    begin = rank * base + min(rank, rem)
    end = begin + base + (1 if rank < rem else 0)
    return begin, end


def shard_nonzeros(I, J, X, row_range):
    """Entries with row_begin <= i < row_end, original (matrix) order kept."""
    b, e = row_range
    I = np.asarray(I)
    keep = (I >= b) & (I < e)
    return (np.ascontiguousarray(I[keep]), np.ascontiguousarray(np.asarray(J)[keep]),
            np.ascontiguousarray(np.asarray(X)[keep]))


class DeviceArray:
    """Zero-copy view of library-owned device memory for torch.as_tensor (CUDA array interface)."""

    def __init__(self, ptr, count, typestr="<f4"):
        self.__cuda_array_interface__ = {"shape": (int(count),), "typestr": typestr,
                                         "data": (int(ptr), False), "version": 2, "strides": None}


class _TorchTransport:
    """ge_transport over torch.distributed: the library hands device buffers, torch sums them (async all_reduce)."""

    def __init__(self, device, group=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.group, self.device = torch, dist, group, device
        self.work = {}
        self.next = 1
        self._start = capi.TRANSPORT_START(self.start)
        self._wait = capi.TRANSPORT_WAIT(self.wait)
        self._bcast = capi.TRANSPORT_BCAST(self.broadcast)
        self.struct = capi.Transport(None, self._start, self._wait, self._bcast)

    def _tensor(self, buf, count, dtype):
        t = self.torch.as_tensor(DeviceArray(buf, count, "<i2" if dtype == capi.GE_DTYPE_BF16 else "<f4"), device=self.device)
        return t.view(self.torch.bfloat16) if dtype == capi.GE_DTYPE_BF16 else t

    def start(self, user, buf, count, dtype, ticket):
        try:
            t = self._tensor(buf, count, dtype)
            k = self.next; self.next += 1
            self.work[k] = (self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group, async_op=True), t)
            ticket[0] = k
            return capi.GE_OK
        except Exception as e:                        # no exception may cross the C boundary
            print("geglove transport.start:", e)
            return capi.GE_ERR_STATE

    def wait(self, user, ticket):
        try:
            w, _ = self.work.pop(int(ticket or 0))
            w.wait()
            if self.torch.cuda.is_available():
                self.torch.cuda.synchronize(self.device)
            return capi.GE_OK
        except Exception as e:
            print("geglove transport.wait:", e)
            return capi.GE_ERR_STATE

    def broadcast(self, user, buf, count, dtype, src):
        try:
            t = self._tensor(buf, count, dtype)
            self.dist.broadcast(t, src=src, group=self.group)
            if self.torch.cuda.is_available():
                self.torch.cuda.synchronize(self.device)
            return capi.GE_OK
        except Exception as e:
            print("geglove transport.broadcast:", e)
            return capi.GE_ERR_STATE


class LocalGroup:
    """ge_local_group: the ranks are THREADS of this process (one ge_glove + one ContextSync each) and meet in host memory.
    What the C++ CLI uses for `device: {gpus: N}` on fewer than N devices, and what lets one GPU run an N-rank exchange through
    the library's own take / land kernels (the callback transport needs a process per rank; a box allows few of those)."""

    def __init__(self, world):
        self.world = int(world)
        self._g = C.c_void_p()
        capi.check(capi.lib().ge_local_group_create(self.world, C.byref(self._g)))

    def abort(self):
        capi.lib().ge_local_group_abort(self._g)

    def close(self):
        if getattr(self, "_g", None) and self._g.value:
            capi.lib().ge_local_group_destroy(self._g)
            self._g = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ContextSync:
    """ge_sync of one rank's trainer handle.  sync() = exact replicas after every step; turn() = the all-reduce of step k
    runs under the epoch of step k + 1 (deltas land one step late); replicate() ends a run with identical fp32 tables."""

    def __init__(self, optimizer, world, rank, wire="bf16", accum_every=4, transport="rccl", device=None, group=None, rccl_id=None, local_group=None):
        self.world, self.rank = int(world), int(rank)
        self._h = C.c_void_p()
        self._keep = None
        cfg = capi.SyncCfg()
        cfg.world, cfg.rank = self.world, self.rank
        cfg.wire = capi.GE_DTYPE_BF16 if wire == "bf16" else capi.GE_DTYPE_F32
        cfg.accum_every = int(accum_every)
        if self.world > 1 and local_group is not None:
            self._keep = local_group                      # the group outlives every ge_sync made for it
            cfg.local_group = local_group._g
        elif self.world > 1 and transport == "torch":
            self._keep = _TorchTransport(device, group)
            cfg.transport = C.pointer(self._keep.struct)
        elif self.world > 1:
            if rccl_id is None:
                # rank 0 makes the id and EVERY rank takes part in the broadcast whatever happened on rank 0: a failure there
                # (librccl not loadable) travels as an error marker, so that all ranks raise together and reach the caller's
                # vote (context_sync_for) instead of one rank leaving the others inside the broadcast
                import torch.distributed as dist
                box = [None]
                if dist.get_rank(group) == 0:
                    buf = (C.c_char * 128)()
                    st = capi.lib().ge_rccl_unique_id(buf)
                    box[0] = bytes(buf.raw) if st == capi.GE_OK else (int(st), capi.lib().ge_last_error().decode(errors="replace"))
                dist.broadcast_object_list(box, src=0, group=group)
                if not isinstance(box[0], (bytes, bytearray)):
                    st, msg = box[0] if box[0] else (capi.GE_ERR_STATE, "no RCCL id arrived from rank 0")
                    raise capi.GeError(st, "rank 0 could not make an RCCL unique id: " + msg)
                rccl_id = box[0]
            self._keep = C.create_string_buffer(bytes(rccl_id), 128)
            cfg.rccl_id = C.cast(self._keep, C.c_void_p)
        capi.check(capi.lib().ge_sync_create(optimizer._h, C.byref(cfg), C.byref(self._h)))

    def epoch(self, iteration=0, segments=0):
        """One epoch of this rank's handle with the hub rows reconciled `segments` times on the way (ge_sync_epoch; every rank calls
        it).  Returns the rank's cost sum, like Adagrad.epoch."""
        c = C.c_double(0.0)
        capi.check(capi.lib().ge_sync_epoch(self._h, int(iteration), int(segments), C.byref(c)))
        return c.value

    def hub_rows(self):
        """The context rows that ge_sync_epoch reconciles between the segments of an epoch (the union of the ranks' hub columns)."""
        n = C.c_int32(0)
        capi.check(capi.lib().ge_sync_hub_rows(self._h, None, 0, C.byref(n)))
        out = np.empty(n.value, np.int32)
        if n.value:
            capi.check(capi.lib().ge_sync_hub_rows(self._h, out.ctypes.data_as(C.POINTER(C.c_int32)), n.value, C.byref(n)))
        return out

    def hub_exchange(self): capi.check(capi.lib().ge_sync_hub_exchange(self._h))
    def hub_exchange_live(self): capi.check(capi.lib().ge_sync_hub_exchange_live(self._h))

    def live_rows(self):
        """The context rows exchanged beside the running epoch kernel: hubs of the kernel on every rank."""
        n = C.c_int32(0)
        capi.check(capi.lib().ge_sync_live_rows(self._h, None, 0, C.byref(n)))
        out = np.empty(n.value, np.int32)
        if n.value:
            capi.check(capi.lib().ge_sync_live_rows(self._h, out.ctypes.data_as(C.POINTER(C.c_int32)), n.value, C.byref(n)))
        return out

    def hub_plan(self, segments=0):
        """What epoch(., segments) does with the hub rows: {"live": bool, "exchanges": per epoch, "live_rows": rows exchanged beside the kernel}."""
        live, n, rows = C.c_int32(0), C.c_int32(0), C.c_int32(0)
        capi.check(capi.lib().ge_sync_hub_plan(self._h, int(segments), C.byref(live), C.byref(n), C.byref(rows)))
        return {"live": bool(live.value), "exchanges": n.value, "live_rows": rows.value}

    def begin(self, everything=False): capi.check(capi.lib().ge_sync_begin(self._h, int(bool(everything))))
    def finish(self): capi.check(capi.lib().ge_sync_finish(self._h))
    def turn(self): capi.check(capi.lib().ge_sync_turn(self._h))
    def sync(self): capi.check(capi.lib().ge_sync_sync(self._h))
    def replicate(self, src=0): capi.check(capi.lib().ge_sync_replicate(self._h, int(src)))

    def allreduce(self, values, op="sum"):
        """Host doubles over the ranks through the library's RCCL communicator (what a Java host would use for the cost)."""
        a = np.ascontiguousarray(values, np.float64).reshape(-1)
        capi.check(capi.lib().ge_sync_allreduce_f64(self._h, a.ctypes.data_as(C.POINTER(C.c_double)), a.shape[0], 0 if op == "sum" else 1))
        return a

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            capi.lib().ge_sync_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def context_sync_for(optimizer, device, lazy_every=4, wire="bf16", group=None, transport=None):
    """The ContextSync of one rank's handle inside an initialised torch.distributed job: RCCL when the backend is nccl (one
    GPU per rank), callbacks into torch.distributed otherwise (gloo: ranks that share a GPU).  Should the library's own RCCL
    communicator fail to come up on ANY rank (the ranks agree through an all-reduce), every rank falls back to the callback
    transport over the job's own process group -- the same exchange, torch's communicator instead of the library's."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    if transport is None:
        transport = "rccl" if dist.get_backend(group) == "nccl" else "torch"
    if transport != "rccl" or world == 1:
        return ContextSync(optimizer, world, rank, wire=wire, accum_every=lazy_every, transport=transport, device=device, group=group)
    sync, err = None, None
    try:
        sync = ContextSync(optimizer, world, rank, wire=wire, accum_every=lazy_every, transport="rccl", device=device, group=group)
    except capi.GeError as e:
        err = e
    ok = torch.tensor([0.0 if sync is None else 1.0], device=device)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
    if float(ok.item()) == 1.0:
        return sync
    if sync is not None:
        sync.close()
    if rank == 0:
        print("geglove: the library's RCCL communicator did not come up (%s); exchanging through torch.distributed instead" % (err,), flush=True)
    return ContextSync(optimizer, world, rank, wire=wire, accum_every=lazy_every, transport="torch", device=device, group=group)
