"""Epoch time against the number of wavefront slots left free (cfg.workers = -k): what the overlapped multi-GPU
exchange pays for keeping room for the all-reduce kernels.   python tools/worker_sweep.py [nnz] [dim] [ranks]
The variants are measured in alternation (A B A B ...) because boxes drift by several percent over a minute."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "graph-embeddings_amd"))
import geglove                          # noqa: E402
from geglove import parallel, synth     # noqa: E402

nnz = int(sys.argv[1]) if len(sys.argv) > 1 else 32_000_000
D = int(sys.argv[2]) if len(sys.argv) > 2 else 200
world = int(sys.argv[3]) if len(sys.argv) > 3 else 1
V = 625_000 * world
rows = parallel.shard_rows(V, world, 0)
I, J, X, xmax = synth.synthetic_coo_shard(V, rows, nnz, seed=0xC0FFEE)
variants = tuple(int(x) for x in os.environ.get("GE_SWEEP", "0,-128,-256,-512,-1024").split(","))
opts = {}
variants = list(enumerate(variants))                 # the same setting may appear twice
for key in variants:
    k = key[1]
    cfg = geglove.Configuration({"graph": "synthetic", "method": "glove", "dim": D, "threads": 1,
                                 "bca": {"alpha": 0.1, "epsilon": 1e-3, "directed": True},
                                 "opt": {"method": "adagrad", "tolerance": 0, "maxiter": 4}, "output": {"uri": []},
                                 "device": {"mode": "hogwild", "shuffle": "device", "seed": 42, "workers": k,
                                            "row_range": rows if world > 1 else (0, 0)}})
    opts[key] = geglove.createOptimizer(cfg, geglove.CooMatrix(V, I, J, X, xmax))
    opts[key].epoch(0)
ms = {key: [] for key in variants}
for it in range(1, 6):
    for key in variants:
        opts[key].epoch(it)
        ms[key].append(opts[key].last_kernel_ms()[0])
for key in variants:
    print("workers %6d -> %5d in flight: %.2f ms/epoch (min %.2f, max %.2f)  %.3g updates/s"
          % (key[1], opts[key].info()["groups_in_flight"], np.mean(ms[key]), np.min(ms[key]), np.max(ms[key]), len(I) / np.mean(ms[key]) * 1e3), flush=True)
