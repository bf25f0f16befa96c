"""One rank's share of the 8-GPU problem (BASELINE C4: 5 M vertices, dim 200) on ONE GPU: this rank's 625 k focus rows and ~103 M
nonzeros, the whole replicated context side, and ge_sync's take / land passes over it with a transport that moves nothing
(so the all-reduce itself is NOT in these numbers: they are what the exchange costs on the compute stream).
    python3 tools/r02/shard_rehearsal.py [world]"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "graph-embeddings_amd"))
import geglove                          # noqa: E402
from geglove import capi, parallel, synth   # noqa: E402

world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
D, rows_per = 200, 625_000
V = rows_per * world
rows = parallel.shard_rows(V, world, 0)
I, J, X, xmax = synth.synthetic_coo_shard(V, rows, 125_000_000, seed=0xC0FFEE)
cfg = geglove.Configuration({"graph": "synthetic", "method": "glove", "dim": D, "threads": 1,
                             "bca": {"alpha": 0.1, "epsilon": 1e-3, "directed": True},
                             "opt": {"method": "adagrad", "tolerance": 0, "maxiter": 8}, "output": {"uri": []},
                             "device": {"mode": "hogwild", "shuffle": "device", "seed": 42, "row_range": rows, "workers": -256}})
t0 = time.perf_counter()
opt = geglove.createOptimizer(cfg, geglove.CooMatrix(V, I, J, X, xmax))
t_create = time.perf_counter() - t0

# a transport that moves nothing
start = capi.TRANSPORT_START(lambda user, buf, count, dtype, ticket: 0)
wait = capi.TRANSPORT_WAIT(lambda user, ticket: 0)
bcast = capi.TRANSPORT_BCAST(lambda user, buf, count, dtype, src: 0)
tr = capi.Transport(None, start, wait, bcast)
sc = capi.SyncCfg(); sc.world, sc.rank, sc.wire, sc.accum_every = world, 0, capi.GE_DTYPE_BF16, 4
sc.transport = C.pointer(tr)
h = C.c_void_p()
t0 = time.perf_counter()
capi.check(capi.lib().ge_sync_create(opt._h, C.byref(sc), C.byref(h)))
t_sync_create = time.perf_counter() - t0
out = {"world": world, "V": V, "nnz": len(I), "create_s": round(t_create, 2), "sync_create_s": round(t_sync_create, 2), "epoch_ms": [], "turn_ms": []}
hip = C.CDLL(None)                                # the HIP runtime capi.lib() has mapped (RTLD_GLOBAL)
for it in range(8):
    opt.epoch(it)
    out["epoch_ms"].append(round(opt.last_kernel_ms()[0], 2))
    t0 = time.perf_counter()
    capi.check(capi.lib().ge_sync_turn(h))          # the null transport makes the library drain its stream: wall time = the passes
    hip.hipDeviceSynchronize()                      # waits for the land / take kernels
    out["turn_ms"].append(round((time.perf_counter() - t0) * 1e3, 2))
print(json.dumps(out), flush=True)
capi.lib().ge_sync_destroy(h)
opt.close()
