#!/usr/bin/env python3
"""Hub threshold 0.25 against 0.05 at the bench size on BOTH kinds of placement: handles created one after another in one process (each
while the previous one is alive: other physical pages), no placement search, alternating thresholds."""
import json, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "graph-embeddings_amd"))
import geglove
from geglove import synth
V, D = 625_000, 200
I, J, X, xmax = synth.synthetic_coo_shard(V, (0, V), 125_000_000, seed=0xC0FFEE)
co = geglove.CooMatrix(V, I, J, X, xmax)
prev = None
for k in range(int(sys.argv[1]) if len(sys.argv) > 1 else 12):
    theta = (0.25, 0.05)[k % 2]
    cfg = geglove.Configuration({"graph": "s", "method": "glove", "dim": D, "threads": 1, "bca": {"alpha": 0.1, "epsilon": 1e-3},
                                 "opt": {"method": "adagrad", "tolerance": 0, "maxiter": 4}, "output": {"uri": []},
                                 "device": {"mode": "hogwild", "shuffle": "device", "seed": 42, "layout": ["first_placement"], "hot_theta": theta}})
    opt = geglove.createOptimizer(cfg, co)
    ms = []
    for it in range(3):
        opt.epoch(it); ms.append(round(opt.last_kernel_ms()[0], 2))
    print(json.dumps({"handle": k, "theta": theta, "epoch_ms": ms, "hub_columns": opt.info()["hot_columns"]}), flush=True)
    if prev is not None: prev.close()
    prev = opt
prev.close()
