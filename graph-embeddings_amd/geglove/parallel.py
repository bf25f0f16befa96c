"""Multi-GPU row sharding of the trainer: one process per GPU.

GPU g owns focus rows / fBias / their gradSq for a contiguous row block and every nonzero whose
i falls in it (BookmarkColoring output is already grouped by i).  The context factors (context,
cBias, gradSqContext, gradSqCBias) are replicated, updated locally Hogwild, and reconciled by a
periodic all-reduce of the per-rank deltas over RCCL/xGMI (ContextSync below: deltas of the context rows and
of the AdaGrad accumulators add up, the context biases take the mean over the ranks that changed them).

The reference has no multi-device semantics (single JVM); parity of this path is statistical
(cost history vs. a 1-GPU run), stated in DESIGN.md.

This module only needs torch + torch.distributed; it works on CPU tensors with gloo (tests)
and on device memory with nccl (= RCCL).  It never computes updates itself.
"""
import numpy as np


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


class Bf16Context:
    """Zero-copy torch views of a GE_DTYPE_BF16 handle's context rows (ge_glove_context_layout): the bf16 table and the
    fp32 master rows of the columns that are hubs on this rank."""

    def __init__(self, optimizer, device):
        import ctypes as C
        import torch
        from . import capi
        lay = capi.ContextLayout()
        capi.check(capi.lib().ge_glove_context_layout(optimizer._h, C.byref(lay)))
        if lay.dtype != capi.GE_DTYPE_BF16:
            raise ValueError("the handle stores its context rows as fp32: pass them as `sums`")
        self.vocab_size, self.dim, self.n_hub = lay.vocab_size, lay.dim, lay.n_hub
        self.hub_index_ptr = lay.hub_index
        self.table = torch.as_tensor(DeviceArray(lay.table, lay.vocab_size * lay.dim, "<i2"), device=device).view(torch.bfloat16)
        self.hub_rows = (torch.as_tensor(DeviceArray(lay.hub_rows, lay.n_hub * lay.dim), device=device) if lay.n_hub
                         else torch.empty(0, dtype=torch.float32, device=device))
        self.hub_index = torch.as_tensor(DeviceArray(lay.hub_index, lay.vocab_size, "<i4"), device=device)

    def values_f32(self):
        """The current row values as one fp32 tensor [vocab_size * dim]: the bf16 table widened, hub rows from their masters."""
        import torch
        full = self.table.to(torch.float32).view(self.vocab_size, self.dim)
        if self.n_hub:
            cols = torch.nonzero(self.hub_index >= 0).view(-1)
            full[cols] = self.hub_rows.view(self.n_hub, self.dim)[self.hub_index[cols].long()]
        return full.view(-1).contiguous()


def context_sync_for(optimizer, device, lazy_every=4, wire="bf16", group=None):
    """The ContextSync of one rank's trainer handle, whatever its storage: zero-copy torch views of the library's device
    tables (ge_glove_context_layout), each under the merge rule DESIGN.md section 7 gives it.
      fp32 rows (fat: [V x (dim+4)], bias at [dim]): context rows summed (ge_exchange_turn_rows), cBias column averaged over
        the ranks that moved it (a strided view), the whole accumulator table -- gradSqContext with gradSqCBias in its bias
        column -- summed every `lazy_every`-th turn;
      bf16 rows: Bf16Context + the separate fp32 bias / accumulator tables."""
    import ctypes as C
    import torch
    from . import capi
    lay = capi.ContextLayout()
    capi.check(capi.lib().ge_glove_context_layout(optimizer._h, C.byref(lay)))
    V, D, DS = lay.vocab_size, lay.dim, lay.row_stride

    def wrap(name):
        ptr, cnt = optimizer.device_ptr(name)
        return torch.as_tensor(DeviceArray(ptr, cnt), device=device)

    if lay.dtype == capi.GE_DTYPE_BF16:
        return ContextSync(sums=[], means=[wrap("cbias")], bf16_tables=[Bf16Context(optimizer, device)],
                           lazy_sums=[wrap("gsq_context"), wrap("gsq_cbias")], lazy_every=lazy_every, wire=wire, group=group)
    table = torch.as_tensor(DeviceArray(lay.table, V * DS), device=device)
    accum = torch.as_tensor(DeviceArray(lay.accum, V * DS), device=device)
    if DS == D:                                     # plain rows: biases are tables of their own
        return ContextSync(sums=[table], means=[wrap("cbias")], lazy_sums=[accum, wrap("gsq_cbias")], lazy_every=lazy_every, wire=wire, group=group)
    return ContextSync(sums=[], means=[table.view(V, DS)[:, D]], row_tables=[(table, V, DS, D)], lazy_sums=[accum],
                       lazy_every=lazy_every, wire=wire, group=group)


class ContextSync:
    """Reconciles the replicated context-side tables after every rank has run its local pass.

    sums:   [tensor] tables whose per-rank deltas ADD:  new = old + sum_g (local_g - old).  The context rows
            (steps are scaled by the 0.05 learning rate, so concurrent moves compose like sequential ones) and the
            AdaGrad accumulators (squared gradients add up no matter which rank saw them).
    means:  [tensor] per-element tables merged by the MEAN over the ranks that changed the element:
            new = old + sum_g delta_g / #{g : delta_g != 0}.  The context biases: the reference updates biases
            WITHOUT a learning rate (Adagrad.java:88-89), one rank alone already moves a hub bias most of the
            way, and adding eight such moves diverges (x129 / x2148 cost after 6 epochs with 4 / 8 ranks in the
            oracle simulation); an element only one rank touched still gets its full update.
    With this rule 8 simulated ranks, one sync per epoch, stay within 6 % of the single-process oracle for the
    first epochs and within 0.2 % from epoch 7 on.
    Works on CPU tensors with gloo (tests) and on device memory with nccl = RCCL over xGMI (bench.py).
    """

    def __init__(self, sums, means, lazy_sums=(), lazy_every=4, wire="bf16", group=None, bf16_tables=(), row_tables=()):
        """lazy_sums: `sums` tables that are reconciled only every `lazy_every`-th call -- the AdaGrad accumulators:
        between syncs each rank keeps adding its own squared gradients (its steps are then at most sqrt(world)
        larger than with the global sum); the oracle simulation shows no difference in the cost trajectory
        (within 1 %) between syncing them every step, every 4th step or never, and it halves the bytes.
        wire: "bf16" sends the deltas of the large tables as bf16 (half the bytes over xGMI); they are small
        increments on top of an fp32 table that never leaves the GPU, and the oracle simulation shows cost
        trajectories identical to three decimals with a bf16 ring sum.  "f32" sends them as they are.
        bf16_tables: Bf16Context objects (context rows stored as bf16 + fp32 hub master rows, GE_DTYPE_BF16); they take
        the begin / finish / turn form only (sync() then is begin() + finish())."""
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.group = torch, dist, group
        self.world = dist.get_world_size(group)
        self.wire = wire
        self.sums = list(sums)
        self.lazy = list(lazy_sums)
        self.lazy_every = max(1, int(lazy_every))
        self.means = list(means)
        self.old_s = [t.clone() for t in self.sums]
        self.old_l = [t.clone() for t in self.lazy]
        self.old_m = [t.clone() for t in self.means]
        self.calls = 0
        self._wbuf = {}
        # bf16_tables: Bf16Context objects -- context tables stored as bf16 with fp32 master rows for the hub columns
        # (GE_DTYPE_BF16).  They only go through the take / land form below (ge_exchange_turn_bf16).
        self.bf16 = list(bf16_tables)
        # row_tables: (tensor, rows, row_stride, cols) -- the row part of a table of fat rows (fp32 Hogwild layout,
        # ge_context_layout.row_stride): summed like `sums`, through ge_exchange_turn_rows; begin / finish / turn form only.
        self.rows = list(row_tables)
        if (self.bf16 or self.rows) and self.world > 1:
            self._entries()                             # their base is the row values NOW, before any local pass

    def sync(self):
        if self.world == 1:
            return
        if self.bf16 or self.rows:                      # these layouts have no torch-op form: take, all-reduce, land at once
            self.begin()
            self.finish()
            return
        torch, dist = self.torch, self.dist
        self.calls += 1
        pairs = list(zip(self.sums, self.old_s))
        if self.calls % self.lazy_every == 0:
            pairs += list(zip(self.lazy, self.old_l))
        work, counts, wired = [], [], []
        for t, o in pairs:
            if self.wire == "bf16" and t.numel() >= (1 << 20):
                w = self._wbuf.get(id(t))
                if w is None:
                    w = self._wbuf[id(t)] = torch.empty(t.shape, dtype=torch.bfloat16, device=t.device)
                torch.sub(t, o, out=w)                                  # delta, narrowed in the same pass
            else:
                t.sub_(o)                                               # t now holds this rank's delta
                w = t
            wired.append(w)
            work.append(dist.all_reduce(w, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        for t, o in zip(self.means, self.old_m):
            t.sub_(o)
            cnt = t.ne(0).to(torch.float32)
            counts.append(cnt)
            work.append(dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            work.append(dist.all_reduce(cnt, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        for w in work:
            w.wait()
        for (t, o), w in zip(pairs, wired):
            if w is not t:
                torch.add(o, w, out=t)                                  # old + summed delta, widened in the same pass
            else:
                t.add_(o)
            o.copy_(t)
        for t, o, cnt in zip(self.means, self.old_m, counts):
            t.div_(cnt.clamp_(min=1.0))
            t.add_(o)
            o.copy_(t)

    # ---- overlapped exchange: the all-reduce of step k's deltas runs under step k+1 -------------------------
    #
    #   take   after the local pass of step k: delta_k = table - base is narrowed into the wire buffer, a copy of it
    #          is kept ("own"), base += own (what bf16 dropped stays in table - base and leaves with the next delta), and
    #          the all-reduces start on the backend's own stream;
    #   land   after the local pass of step k+1: waits for them and adds what the OTHER ranks contributed,
    #          merged_k - own_k, to the table (and to the base, so it is not taken for this rank's next delta).
    #
    # turn() = land what is in flight, then take; begin() = take only; finish() = land only.  The table itself is
    # never on the wire, so the local pass may keep updating it while the exchange is in flight; a rank sees the
    # other ranks' moves one step late (tests/tools/multirank_sim.py: the merged model trails the synchronous exchange
    # by about one epoch, stable).  Between turns the replicas differ by what is in flight and by the bf16
    # rounding of their own deltas; replicate() makes them identical again.
    #
    # Large fp32 tables in device memory go through ONE fused pass of the HIP library per turn
    # (ge_exchange_turn, csrc/exchange.hip: 24 B per element instead of 44 B over six torch passes); host tensors
    # (the gloo tests) and the small bias tables use the torch ops written out below.

    def _entries(self):
        if not hasattr(self, "_ent"):
            self._ent = ([dict(t=t, o=o, mean=False, lazy=False, work=None) for t, o in zip(self.sums, self.old_s)] +
                         [dict(t=t, o=o, mean=False, lazy=True, work=None) for t, o in zip(self.lazy, self.old_l)] +
                         [dict(t=t, o=o, mean=True, lazy=False, work=None) for t, o in zip(self.means, self.old_m)])
            for k, (t, nrows, stride, cols) in enumerate(self.rows):      # first: the largest exchange gets the whole next epoch to hide under
                self._ent.insert(k, dict(t=t, o=t.clone(), mean=False, lazy=False, work=None, rows=(int(nrows), int(stride), int(cols)), fused=True, cnt=None,
                                      w=self.torch.empty(t.shape, dtype=self.torch.bfloat16, device=t.device),
                                      own=self.torch.empty(t.shape, dtype=self.torch.bfloat16, device=t.device)))
            for b in self.bf16:
                t = b.table
                e = dict(t=t, o=b.values_f32(), mean=False, lazy=False, work=None, bf16=b, fused=True, cnt=None,
                         w=self.torch.empty_like(t), own=self.torch.empty_like(t))
                self._ent.append(e)
            for e in self._ent:
                if "bf16" in e or "rows" in e:
                    continue
                t = e["t"]
                narrow = self.wire == "bf16" and not e["mean"] and t.numel() >= (1 << 20)
                e["w"] = self.torch.empty(t.shape, dtype=self.torch.bfloat16 if narrow else t.dtype, device=t.device)
                e["own"] = self.torch.empty_like(e["w"])
                e["cnt"] = None
                e["fused"] = bool(narrow and t.is_cuda and t.dtype == self.torch.float32 and t.is_contiguous())
        return self._ent

    def _fused_turn(self, e, land, take):
        from . import capi                      # the HIP library; fails loudly when it has not been built
        t = e["t"]
        if "rows" in e:
            nrows, stride, cols = e["rows"]
            with self.torch.cuda.device(t.device):
                capi.check(capi.lib().ge_exchange_turn_rows(t.data_ptr(), e["o"].data_ptr(), e["w"].data_ptr(), e["own"].data_ptr(),
                                                            nrows, stride, cols, int(land), int(take),
                                                            self.torch.cuda.current_stream(t.device).cuda_stream))
            return
        if "bf16" in e:
            b = e["bf16"]
            self._seed = (getattr(self, "_seed", 0x5EED) * 1664525 + 1013904223) & 0xFFFFFFFF      # same on every call site, new per turn
            with self.torch.cuda.device(t.device):
                capi.check(capi.lib().ge_exchange_turn_bf16(
                    t.data_ptr(), b.hub_rows.data_ptr() if b.n_hub else None, b.hub_index_ptr, b.vocab_size, b.dim,
                    e["o"].data_ptr(), e["w"].data_ptr(), e["own"].data_ptr(), int(land), int(take),
                    self._seed ^ (self.dist.get_rank(self.group) * 0x9E3779B1 & 0xFFFFFFFF),
                    self.torch.cuda.current_stream(t.device).cuda_stream))
            return
        with self.torch.cuda.device(t.device):
            capi.check(capi.lib().ge_exchange_turn(t.data_ptr(), e["o"].data_ptr(), e["w"].data_ptr(), e["own"].data_ptr(),
                                                   t.numel(), int(land), int(take),
                                                   self.torch.cuda.current_stream(t.device).cuda_stream))

    def _turn(self, land, take, everything=False):
        if self.world == 1:
            return
        torch, dist = self.torch, self.dist
        due = False
        if take:
            self.calls += 1
            due = everything or self.calls % self.lazy_every == 0
        for e in self._entries():
            do_land = land and e["work"] is not None
            do_take = take and (due or not e["lazy"])
            if take and e["work"] is not None and not land:
                raise RuntimeError("finish() the previous exchange first")
            if not (do_land or do_take):
                continue
            t, o, w, own = e["t"], e["o"], e["w"], e["own"]
            if do_land:
                for x in e["work"]:
                    x.wait()
                e["work"] = None
                if e["mean"]:
                    w.div_(e["cnt"].clamp_(min=1.0))
            if e["fused"]:
                self._fused_turn(e, do_land, do_take)
            else:
                if do_land:
                    w.sub_(own)                                         # what the other ranks contributed
                    t.add_(w)
                    o.add_(w)
                if do_take:
                    torch.sub(t, o, out=w)
                    own.copy_(w)
                    o.add_(own)                                         # the base advances by what is sent (error feedback)
            if do_take:
                work = []
                if e["mean"]:
                    e["cnt"] = own.ne(0).to(torch.float32)
                    work.append(dist.all_reduce(e["cnt"], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
                work.append(dist.all_reduce(w, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
                e["work"] = work

    def begin(self, everything=False):
        self._turn(False, True, everything)

    def finish(self):
        self._turn(True, False)

    def turn(self, everything=False):
        self._turn(True, True, everything)

    def replicate(self, src=0):
        """Ends an overlapped run: lands what is in flight, exchanges what has not been sent yet (accumulators
        included), then every rank takes rank `src`'s replica of all tables (their remaining differences are the
        bf16 rounding of own deltas, see above)."""
        if self.world == 1:
            return
        self.turn(everything=True)
        self.finish()
        for e in self._entries():
            if "bf16" in e:
                continue        # rows live partly in per-rank fp32 master rows (hub sets differ per rank): left as landed, equal up to bf16 rounding
            t = e["t"]
            if t.is_contiguous():
                self.dist.broadcast(t, src=src, group=self.group)
            else:                                   # a bias column of fat rows
                tmp = t.contiguous()
                self.dist.broadcast(tmp, src=src, group=self.group)
                t.copy_(tmp)
            e["o"].copy_(t)
