#!/usr/bin/env python3
"""Where in the device's memory do the fast placements lie?  Bench-size handles are created one after another and ALL kept alive (so every
one sits in other physical pages: ~4.5 GB each, 40 of them walk through 180 GB), no placement search; the second epoch's time of each.
   python3 tools/r03/range_probe.py [handles]"""
import json, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "graph-embeddings_amd"))
import geglove
from geglove import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
V, D = 625_000, 200
I, J, X, xmax = synth.synthetic_coo_shard(V, (0, V), 125_000_000, seed=0xC0FFEE)
co = geglove.CooMatrix(V, I, J, X, xmax)
cfg = geglove.Configuration({"graph": "s", "method": "glove", "dim": D, "threads": 1, "bca": {"alpha": 0.1, "epsilon": 1e-3},
                             "opt": {"method": "adagrad", "tolerance": 0, "maxiter": 4}, "output": {"uri": []},
                             "device": {"mode": "hogwild", "shuffle": "device", "seed": 42, "layout": ["first_placement"]}})
keep = []
for k in range(n):
    opt = geglove.createOptimizer(cfg, co)
    opt.epoch(0); opt.epoch(1)
    print(json.dumps({"handle": k, "epoch_ms": round(opt.last_kernel_ms()[0], 2), "focus": "0x%x" % opt.device_ptr("focus")[0], "context": "0x%x" % opt.device_ptr("context")[0]}), flush=True)
    keep.append(opt)
for o in keep: o.close()
