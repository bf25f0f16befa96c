"""Time of the elementwise passes of ContextSync around the all-reduce at the 8-GPU table size (5 M x 200 floats),
on one GPU and without communication.   python tools/exchange_passes.py [elements]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "graph-embeddings_amd"))
from geglove import capi                # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000_000
dev = torch.device("cuda", 0)
t = torch.randn(n, device=dev); o = t.clone(); t.add_(0.01)
w = torch.empty(n, dtype=torch.bfloat16, device=dev); own = torch.empty_like(w)


def timed(f, reps=3):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def begin():
    torch.sub(t, o, out=w); own.copy_(w); o.copy_(t)


def finish():
    w.sub_(own); t.add_(w); o.add_(w)


def sync_form():
    torch.sub(t, o, out=w); torch.add(o, w, out=t); o.copy_(t)


def fused():
    capi.check(capi.lib().ge_exchange_turn(t.data_ptr(), o.data_ptr(), w.data_ptr(), own.data_ptr(), n, 1, 1, None))


print("fused land+take (ge_exchange_turn): %.2f ms = %.2f TB/s" % (timed(fused), 24e-9 * n / timed(fused)))
print("overlap: begin %.2f ms, finish %.2f ms;  synchronous form %.2f ms  (%d elements)" % (timed(begin), timed(finish), timed(sync_form), n))
