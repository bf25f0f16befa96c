#!/usr/bin/env python3
"""ge_sync_turn's passes at the 8-GPU table size on one GPU: V = 5 M context rows, dim 200, bf16 wire, a transport
that moves nothing -- the time on the compute stream of take / land+take / with the accumulator table, by wall clock around a device
synchronize.  Under `rocprofv3 --kernel-trace --stats` / `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` for the kernel's own figures.
    python3 tools/r03/turn_bench.py [dim] [dtype]"""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "graph-embeddings_amd"))
import geglove                              # noqa: E402
from geglove import capi, parallel, synth   # noqa: E402

D = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dtype = sys.argv[2] if len(sys.argv) > 2 else "f32"
world, V = 8, 5_000_000
rows = parallel.shard_rows(V, world, 0)
I, J, X, xmax = synth.synthetic_coo_shard(V, rows, 4_000_000, seed=0xC0FFEE)
cfg = geglove.Configuration({"graph": "synthetic", "method": "glove", "dim": D, "threads": 1, "bca": {"alpha": 0.1, "epsilon": 1e-3, "directed": True},
                             "opt": {"method": "adagrad", "tolerance": 0, "maxiter": 8}, "output": {"uri": []},
                             "device": {"mode": "hogwild", "shuffle": "device", "seed": 42, "row_range": rows, "workers": -256, "dtype": dtype,
                                        "layout": ["first_placement"]}})
opt = geglove.createOptimizer(cfg, geglove.CooMatrix(V, I, J, X, xmax))
start = capi.TRANSPORT_START(lambda user, buf, count, dt, ticket: 0)
wait = capi.TRANSPORT_WAIT(lambda user, ticket: 0)
bcast = capi.TRANSPORT_BCAST(lambda user, buf, count, dt, src: 0)
tr = capi.Transport(None, start, wait, bcast)
sc = capi.SyncCfg(); sc.world, sc.rank, sc.wire, sc.accum_every = world, 0, capi.GE_DTYPE_BF16, 4
sc.transport = C.pointer(tr)
h = C.c_void_p()
capi.check(capi.lib().ge_sync_create(opt._h, C.byref(sc), C.byref(h)))
hip = C.CDLL(None)
out = {"V": V, "dim": D, "dtype": dtype, "row_stride": opt.info()["row_stride"], "turn_ms": []}
for it in range(12):
    opt.epoch(it)
    hip.hipDeviceSynchronize()
    t0 = time.perf_counter()
    capi.check(capi.lib().ge_sync_turn(h))
    hip.hipDeviceSynchronize()
    out["turn_ms"].append(round((time.perf_counter() - t0) * 1e3, 3))
elems = V * D
out["bytes_land_take_rows"] = elems * 22 if dtype == "f32" else None
print(json.dumps(out), flush=True)
capi.lib().ge_sync_destroy(h)
opt.close()
