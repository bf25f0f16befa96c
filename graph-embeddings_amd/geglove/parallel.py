"""Multi-GPU row sharding of the trainer: one process per GPU.

GPU g owns focus rows / fBias / their gradSq for a contiguous row block and every nonzero whose
i falls in it (BookmarkColoring output is already grouped by i).  The context factors (context,
cBias, gradSqContext, gradSqCBias) are replicated, updated locally Hogwild, and reconciled by a
periodic SUM-OF-DELTAS all-reduce over RCCL/xGMI:   new = old + sum_g (local_g - old).

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
    """Sum-of-deltas all-reduce of the replicated context-side tables.

    tensors: list of torch tensors (one per replicated table) that the local trainer updates in
    place.  After every rank has finished its local pass call sync(): each table becomes
    old + sum over ranks of (local - old), computed as  allreduce_sum(local) - (world-1)*old
    so that only ONE collective buffer per table is in flight and no delta temp is needed.
    """

    def __init__(self, tensors, group=None):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.world = dist.get_world_size(group)
        self.tensors = list(tensors)
        self.old = [t.clone() for t in self.tensors]

    def sync(self):
        if self.world == 1:
            return
        handles = [self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group, async_op=True)
                   for t in self.tensors]
        for h in handles:
            h.wait()
        for t, o in zip(self.tensors, self.old):
            t.sub_(o, alpha=float(self.world - 1))
            o.copy_(t)
