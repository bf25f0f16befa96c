"""Multi-GPU row sharding of the trainer: one process per GPU.

GPU g owns focus rows / fBias / their gradSq for a contiguous row block and every nonzero whose
i falls in it (BookmarkColoring output is already grouped by i).  The context factors (context,
cBias, gradSqContext, gradSqCBias) are replicated, updated locally Hogwild, and reconciled by a
periodic all-reduce of the per-rank deltas over RCCL/xGMI (ContextSync below: mean over the ranks
that touched a row for the parameters, sum for the AdaGrad accumulators).

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

    params:  [(tensor, row_len)] tables that hold PARAMETERS (context [V*D] with row_len=D, cBias [V] with
             row_len=1).  Merge rule per row:  new = old + (sum over ranks of delta) / (number of ranks whose
             delta for that row is non-zero).  A row that only one rank touched gets its full update; a hub
             row that every rank moved gets the mean of the moves.  (A plain sum of whole-pass deltas
             diverges: each rank alone already moves a hub row most of the way -- measured in DESIGN.md.)
    accums:  [tensor] AdaGrad accumulators (gradSqContext, gradSqCBias): plain sum of deltas -- squared
             gradients add up no matter which rank saw them.
    Works on CPU tensors with gloo (tests) and on device memory with nccl = RCCL over xGMI (bench.py).
    """

    def __init__(self, params, accums, group=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.group = torch, dist, group
        self.world = dist.get_world_size(group)
        self.params = [(t, int(n)) for t, n in params]
        self.accums = list(accums)
        self.old_p = [t.clone() for t, _ in self.params]
        self.old_a = [t.clone() for t in self.accums]

    def sync(self):
        if self.world == 1:
            return
        torch, dist = self.torch, self.dist
        work = []
        counts = []
        for (t, n), o in zip(self.params, self.old_p):
            t.sub_(o)                                                   # t now holds this rank's delta
            cnt = t.view(-1, n).ne(0).any(dim=1).to(torch.float32)
            counts.append(cnt)
            work.append(dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            work.append(dist.all_reduce(cnt, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        for t, o in zip(self.accums, self.old_a):
            t.sub_(o)
            work.append(dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        for w in work:
            w.wait()
        for (t, n), o, cnt in zip(self.params, self.old_p, counts):
            t.view(-1, n).div_(cnt.clamp_(min=1.0).unsqueeze(1))
            t.add_(o)
            o.copy_(t)
        for t, o in zip(self.accums, self.old_a):
            t.add_(o)
            o.copy_(t)
