"""Does the epoch time depend on WHERE the tables land?  The same handle is created again and again in one process, each
time after a dummy allocation of another size has shifted what the allocator hands out.   python tools/placement_probe.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "graph-embeddings_amd"))
import geglove                          # noqa: E402
from geglove import synth               # noqa: E402

nnz = int(sys.argv[1]) if len(sys.argv) > 1 else 125_000_000
V, D = 625_000, 200
I, J, X, xmax = synth.synthetic_coo_shard(V, (0, V), nnz, seed=0xC0FFEE)
cfg = geglove.Configuration({"graph": "synthetic", "method": "glove", "dim": D, "threads": 1,
                             "bca": {"alpha": 0.1, "epsilon": 1e-3, "directed": True},
                             "opt": {"method": "adagrad", "tolerance": 0, "maxiter": 4}, "output": {"uri": []},
                             "device": {"mode": "hogwild", "shuffle": "device", "seed": 42}})
dev = torch.device("cuda", 0)
for mb in (0, 0, 300, 0, 1300, 2700, 0, 64, 7000, 0):
    pad = torch.empty(mb * 1024 * 1024, dtype=torch.uint8, device=dev) if mb else None
    opt = geglove.createOptimizer(cfg, geglove.CooMatrix(V, I, J, X, xmax))
    opt.epoch(0)
    ms = []
    for it in range(1, 4):
        opt.epoch(it); ms.append(opt.last_kernel_ms()[0])
    ptr = opt.device_ptr("context")[0]
    print("dummy %5d MB before create: epoch %.2f ms   (context table at 0x%x)" % (mb, np.mean(ms), ptr), flush=True)
    opt.close()
    del pad
    torch.cuda.empty_cache()
