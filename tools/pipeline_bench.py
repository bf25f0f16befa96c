#!/usr/bin/env python3
"""End-to-end on the device, shaped like the shipped dblp.config.yml run: DBLP-like graph -> BookmarkColoring
(alpha 0.1, eps 1e-3, directed) -> pGloVe AdaGrad, dim 200.  Prints builder time, per-epoch cost and rate.
This is BASELINE config C3's stand-in at scale (the DBLP dump itself is not available offline)."""
import argparse, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "graph-embeddings_amd")]
import numpy as np
import geglove
from geglove import synth

ap = argparse.ArgumentParser()
ap.add_argument("--authors", type=int, default=60000)
ap.add_argument("--papers", type=int, default=90000)
ap.add_argument("--dim", type=int, default=200)
ap.add_argument("--epochs", type=int, default=8)
ap.add_argument("--method", default="pglove")
ap.add_argument("--layout", default="", help="comma list of device.layout flags (ablation)")
a = ap.parse_args()
g = synth.dblp_like_graph(a.authors, a.papers, 50)
cfg = geglove.Configuration({"graph": "dblp-like", "method": a.method, "dim": a.dim, "threads": 1,
                             "bca": {"alpha": 0.1, "epsilon": 1e-3, "directed": True},
                             "opt": {"method": "adagrad", "tolerance": 1e-4, "maxiter": a.epochs}, "output": {"uri": []},
                             "device": {"mode": "hogwild", "shuffle": "device", "seed": 42, "layout": [x for x in a.layout.split(",") if x]}})
t = time.perf_counter(); bca = geglove.BookmarkColoring(g, cfg); tb = time.perf_counter() - t
n = bca.coOccurrenceCount()
cnt = np.bincount(bca.J, minlength=g["V"])
print("graph V=%d; BCA %.2f s -> %d nonzeros (%.0f per row), max %.6g; busiest column holds %.2f %% of them" % (
    g["V"], tb, n, n / g["V"], bca.max(), 100.0 * cnt.max() / n), flush=True)
t = time.perf_counter(); opt = geglove.Adagrad(bca, cfg, cfg.costFunction()); tc = time.perf_counter() - t
print("trainer create %.2f s; %s" % (tc, opt.info()), flush=True)
for it in range(a.epochs):
    c = opt.epoch(it) / n
    ms, _ = opt.last_kernel_ms()
    print("epoch %d  cost %.6f  kernel %.2f ms  %.3g pair-updates/s" % (it, c, ms, n / ms * 1e3), flush=True)
