"""Is the slow / fast epoch mode a property of the hardware queue?  One process, the same handle created on the null stream
and on several created streams (each maps to its own HSA queue), epochs timed on each.   python tools/queue_probe.py"""
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
dev = torch.device("cuda", 0)
streams = [None] + [torch.cuda.Stream(device=dev) for _ in range(3)] + [torch.cuda.Stream(device=dev, priority=-1), None]
for k, st in enumerate(streams):
    device = {"mode": "hogwild", "shuffle": "device", "seed": 42}
    if st is not None:
        device["stream"] = st.cuda_stream
    cfg = geglove.Configuration({"graph": "synthetic", "method": "glove", "dim": D, "threads": 1,
                                 "bca": {"alpha": 0.1, "epsilon": 1e-3, "directed": True},
                                 "opt": {"method": "adagrad", "tolerance": 0, "maxiter": 4}, "output": {"uri": []}, "device": device})
    opt = geglove.createOptimizer(cfg, geglove.CooMatrix(V, I, J, X, xmax))
    opt.epoch(0)
    ms = []
    for it in range(1, 4):
        opt.epoch(it); ms.append(opt.last_kernel_ms()[0])
    print("stream %-28s epoch %.2f ms" % ("null" if st is None else "created #%d (0x%x)%s" % (k, st.cuda_stream, " high priority" if k == 4 else ""), np.mean(ms)), flush=True)
    opt.close()
