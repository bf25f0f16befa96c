"""Epoch time against the number of wavefront slots left free (cfg.workers = -k): what the overlapped multi-GPU
exchange pays for keeping room for the all-reduce kernels.   python tools/worker_sweep.py [nnz] [dim]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "graph-embeddings_amd"))
import geglove                          # noqa: E402
from geglove import synth               # noqa: E402

nnz = int(sys.argv[1]) if len(sys.argv) > 1 else 32_000_000
D = int(sys.argv[2]) if len(sys.argv) > 2 else 200
V = 625_000
I, J, X, xmax = synth.synthetic_coo_shard(V, (0, V), nnz, seed=0xC0FFEE)
for k in (0, -128, -256, -512, -1024):
    cfg = geglove.Configuration({"graph": "synthetic", "method": "glove", "dim": D, "threads": 1,
                                 "bca": {"alpha": 0.1, "epsilon": 1e-3, "directed": True},
                                 "opt": {"method": "adagrad", "tolerance": 0, "maxiter": 4}, "output": {"uri": []},
                                 "device": {"mode": "hogwild", "shuffle": "device", "seed": 42, "workers": k}})
    opt = geglove.createOptimizer(cfg, geglove.CooMatrix(V, I, J, X, xmax))
    opt.epoch(0)
    ms = []
    for it in range(1, 4):
        opt.epoch(it)
        ms.append(opt.last_kernel_ms()[0])
    print("workers %6d -> %5d in flight: %.2f ms/epoch  (%.3g updates/s)" % (k, opt.info()["groups_in_flight"], np.mean(ms), len(I) / np.mean(ms) * 1e3), flush=True)
    opt.close()
