#!/usr/bin/env python3
"""Is the mode of a handle (two epoch times, fixed when the tables are placed) drawn per PROCESS or per ALLOCATION?
One process creates the bench-size trainer several times, each new handle while the previous one is still alive (so it cannot get the
same memory back), and times three epochs of each.   python3 tools/r02/realloc_probe.py [handles]"""
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "graph-embeddings_amd"))
import geglove                      # noqa: E402
from geglove import synth           # noqa: E402

n_handles = int(sys.argv[1]) if len(sys.argv) > 1 else 6
layout = [x for x in (sys.argv[2] if len(sys.argv) > 2 else "").split(",") if x]        # e.g. first_placement
V, D = 625_000, 200
I, J, X, xmax = synth.synthetic_coo_shard(V, (0, V), 125_000_000, seed=0xC0FFEE)      # the bench matrix
cfg = geglove.Configuration({"graph": "synthetic", "method": "glove", "dim": D, "threads": 1,
                             "bca": {"alpha": 0.1, "epsilon": 1e-3, "directed": True},
                             "opt": {"method": "adagrad", "tolerance": 0, "maxiter": 4}, "output": {"uri": []},
                             "device": {"mode": "hogwild", "shuffle": "device", "seed": 42, "layout": layout}})
co = geglove.CooMatrix(V, I, J, X, xmax)
prev, out = None, []
for k in range(n_handles):
    opt = geglove.createOptimizer(cfg, co)
    ms = []
    for it in range(3):
        opt.epoch(it)
        ms.append(round(opt.last_kernel_ms()[0], 2))
    inf = opt.info()
    out.append({"handle": k, "epoch_ms": ms, "context": "0x%x" % opt.device_ptr("context")[0], "placements": inf["placements"],
                "probe_ms_kept_worst": [round(inf["placement_best_ms"], 3), round(inf["placement_worst_ms"], 3)]})
    print(json.dumps(out[-1]), flush=True)
    if prev is not None:
        prev.close()
    prev = opt
prev.close()
