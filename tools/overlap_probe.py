"""Does a small kernel on a side stream run BESIDE the persistent epoch kernel?  The stand-in for a collective
(tools/micro/probe.hip: `blocks` workgroups streaming 1 GB) is enqueued on a non-blocking side stream right before
ge_glove_epoch, as ContextSync.turn() enqueues the all-reduce; events tell when it started and ended relative to the epoch.
    hipcc -O3 -fPIC -shared --offload-arch=gfx950 tools/micro/probe.hip -o tools/micro/libprobe.so
    python tools/overlap_probe.py [nnz=125000000]
"""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "graph-embeddings_amd"))
import geglove                          # noqa: E402
from geglove import synth               # noqa: E402

nnz = int(sys.argv[1]) if len(sys.argv) > 1 else 125_000_000
V, D = 625_000, 200
I, J, X, xmax = synth.synthetic_coo_shard(V, (0, V), nnz, seed=0xC0FFEE)
probe = C.CDLL(os.path.join(ROOT, "tools", "micro", "libprobe.so"))
probe.probe_copy.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p]
dev = torch.device("cuda", 0)
n = 256 * 1024 * 1024                                   # 1 GB of floats each way
src = torch.ones(n, device=dev); dst = torch.empty(n, device=dev)
scratch = torch.ones(1 << 30, device=dev)              # a 4 GB pass on the main stream takes a couple of ms
side = torch.cuda.Stream(device=dev)                    # pool streams are non-blocking w.r.t. the null stream


def probe_alone(blocks):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    probe.probe_copy(dst.data_ptr(), src.data_ptr(), n, blocks, side.cuda_stream)
    side.synchronize()
    return (time.perf_counter() - t0) * 1e3


for blocks in (32, 64):
    print("probe alone, %d workgroups: %.1f ms for 2 GB of traffic" % (blocks, probe_alone(blocks)), flush=True)

for reserve in (0, -256, 0, -256):
    cfg = geglove.Configuration({"graph": "synthetic", "method": "glove", "dim": D, "threads": 1,
                                 "bca": {"alpha": 0.1, "epsilon": 1e-3, "directed": True},
                                 "opt": {"method": "adagrad", "tolerance": 0, "maxiter": 4}, "output": {"uri": []},
                                 "device": {"mode": "hogwild", "shuffle": "device", "seed": 42, "workers": reserve}})
    opt = geglove.createOptimizer(cfg, geglove.CooMatrix(V, I, J, X, xmax))
    opt.epoch(0)
    opt.epoch(1); base_ms = opt.last_kernel_ms()[0]
    for blocks in (32, 64):
        torch.cuda.synchronize()
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        t0 = time.perf_counter()
        for racy in (False, True):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            if racy:
                # as in ContextSync.turn(): a pass on the main stream (the fused exchange kernel), the side stream waits for
                # it, and the epoch is queued behind it on the main stream -> probe and epoch become ready TOGETHER
                scratch.mul_(1.0001)
                gate = torch.cuda.Event(); gate.record()
                side.wait_event(gate)
            with torch.cuda.stream(side):
                e0.record()
                probe.probe_copy(dst.data_ptr(), src.data_ptr(), n, blocks, side.cuda_stream)
                e1.record()
                probe.probe_copy(dst.data_ptr(), src.data_ptr(), n, blocks, side.cuda_stream)   # a SECOND collective: ready only mid-epoch
                e2.record()
            opt.epoch(2)                                    # blocks the host until the epoch kernel is done
            t_epoch = (time.perf_counter() - t0) * 1e3
            side.synchronize()
            t_all = (time.perf_counter() - t0) * 1e3
            print("workers %5d (%d in flight), probe %2d workgroups, %s: epoch alone %.1f ms | with probes: kernel %.1f ms | 1st probe done %.1f ms, 2nd %.1f ms after the 1st started | all done after %.1f ms"
                  % (reserve, opt.info()["groups_in_flight"], blocks, "ready together" if racy else "probe queued first", base_ms, opt.last_kernel_ms()[0], e0.elapsed_time(e1), e0.elapsed_time(e2), t_all), flush=True)
    opt.close()
