#!/usr/bin/env python3
"""Several bench-size trainer handles in ONE process, each created while the previous one is alive (so the driver hands out other
physical pages: DESIGN.md 6, the per-allocation draw), two epochs each.  Run under `rocprofv3 --pmc <per-instance TCC counters>`
by tools/r03/channel_probe.sh: fast and slow placements of the same kernel on the same data then sit side by side in one trace.
   python3 tools/r03/place_modes.py [handles] [layout-flags] [dim]"""
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "graph-embeddings_amd"))
import geglove                      # noqa: E402
from geglove import synth           # noqa: E402

n_handles = int(sys.argv[1]) if len(sys.argv) > 1 else 5
layout = [x for x in (sys.argv[2] if len(sys.argv) > 2 else "first_placement").split(",") if x and x != "none"]
D = int(sys.argv[3]) if len(sys.argv) > 3 else 200
V = 625_000
I, J, X, xmax = synth.synthetic_coo_shard(V, (0, V), 125_000_000, seed=0xC0FFEE)      # the bench matrix
cfg = geglove.Configuration({"graph": "synthetic", "method": "glove", "dim": D, "threads": 1,
                             "bca": {"alpha": 0.1, "epsilon": 1e-3, "directed": True},
                             "opt": {"method": "adagrad", "tolerance": 0, "maxiter": 4}, "output": {"uri": []},
                             "device": {"mode": "hogwild", "shuffle": "device", "seed": 42, "layout": layout}})
co = geglove.CooMatrix(V, I, J, X, xmax)
prev = None
for k in range(n_handles):
    opt = geglove.createOptimizer(cfg, co)
    ms = []
    for it in range(2):
        opt.epoch(it)
        ms.append(round(opt.last_kernel_ms()[0], 3))
    inf = opt.info()
    print(json.dumps({"handle": k, "epoch_ms": ms, "focus": "0x%x" % opt.device_ptr("focus")[0], "context": "0x%x" % opt.device_ptr("context")[0],
                      "placements": inf["placements"], "row_stride": inf["row_stride"]}), flush=True)
    if prev is not None:
        prev.close()
    prev = opt
prev.close()
