#!/usr/bin/env python3
"""Per-epoch cost and table health of an N-rank run through a ge_local_group on one GPU, at the shape of tests/test_parallel_gpu.py's
eight-rank test.   python3 tools/r03/ranks_debug.py world exchange wire lazy_every [V N D epochs]"""
import os, sys, threading
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "graph-embeddings_amd"))
import geglove
from geglove import parallel, synth

world, exchange, wire, lazy = int(sys.argv[1]), sys.argv[2], sys.argv[3], int(sys.argv[4])
V, N, D, epochs = (int(x) for x in (sys.argv[5:9] if len(sys.argv) >= 9 else (100000, 3000000, 200, 8)))
segments = int(sys.argv[9]) if len(sys.argv) > 9 else 0          # -1: plain ge_glove_epoch (one exchange per epoch for every row)
CTX = ("context", "cbias", "gsq_context", "gsq_cbias")
I, J, X, xmax = synth.synthetic_coo(V, N, seed=13)
col = np.bincount(J, minlength=V)
print("segments %d; world %d %s wire %s lazy %d: V=%d N=%d D=%d; busiest columns %s" % (segments, world, exchange, wire, lazy, V, len(I), D, np.sort(col)[-4:].tolist()), flush=True)
grp = parallel.LocalGroup(world) if world > 1 else None
bar = threading.Barrier(world)
cost = np.zeros((epochs, world)); health = [[None] * world for _ in range(epochs)]

def body(r):
    rows = parallel.shard_rows(V, world, r)
    si, sj, sx = parallel.shard_nonzeros(I, J, X, rows)
    cfg = geglove.Configuration({"graph": "s", "method": "glove", "dim": D, "threads": 1, "bca": {"alpha": .1, "epsilon": 1e-3},
                                 "opt": {"method": "adagrad", "maxiter": epochs, "tolerance": 0}, "output": {"uri": []},
                                 "device": {"mode": "hogwild", "shuffle": "device", "seed": 42, "row_range": rows if world > 1 else (0, 0),
                                            "workers": -256 if exchange == "overlap" else 0}})
    opt = geglove.Adagrad(geglove.CooMatrix(V, si, sj, sx, xmax), cfg, cfg.costFunction())
    bar.wait()
    sync = parallel.ContextSync(opt, world, r, wire=wire, accum_every=lazy, local_group=grp) if world > 1 else None
    for it in range(epochs):
        cost[it, r] = sync.epoch(it, segments) if (sync and segments >= 0) else opt.epoch(it)
        pre = opt.get_state("context")
        if sync: sync.turn() if exchange == "overlap" else sync.sync()
        post = opt.get_state("context"); g = opt.get_state("gsq_context")
        health[it][r] = (float(np.abs(pre).max()), float(np.abs(post).max()), int(np.count_nonzero(~np.isfinite(post))), float(g.max()), float(g.min()))
        bar.wait()
    if sync: sync.close()
    opt.close()

th = [threading.Thread(target=body, args=(r,)) for r in range(world)]
[t.start() for t in th]; [t.join() for t in th]
for it in range(epochs):
    print("epoch %d cost %.6f  rank0: max|ctx| before/after exchange %.3g / %.3g, non-finite %d, gsq max %.3g min %.3g" % ((it + 1, cost[it].sum() / len(I)) + health[it][0]), flush=True)
