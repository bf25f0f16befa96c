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

    def __init__(self, sums, means, lazy_sums=(), lazy_every=4, wire="bf16", group=None):
        """lazy_sums: `sums` tables that are reconciled only every `lazy_every`-th call -- the AdaGrad accumulators:
        between syncs each rank keeps adding its own squared gradients (its steps are then at most sqrt(world)
        larger than with the global sum); the oracle simulation shows no difference in the cost trajectory
        (within 1 %) between syncing them every step, every 4th step or never, and it halves the bytes."""
        """wire: "bf16" sends the deltas of the large tables as bf16 (half the bytes over xGMI); they are small
        increments on top of an fp32 table that never leaves the GPU, and the oracle simulation shows cost
        trajectories identical to three decimals with a bf16 ring sum.  "f32" sends them as they are."""
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

    def sync(self):
        if self.world == 1:
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
