#!/usr/bin/env python3
"""Per-epoch mean cost of the device trainer (Hogwild) next to the sequential CPU oracle on the
same matrix and seed.  Diagnostic for the statistical parity of the racy mode."""
import argparse, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(REPO, "graph-embeddings_amd"), os.path.join(REPO, "oracle")]
import numpy as np
import geglove
from geglove import synth
import oracle as O

ap = argparse.ArgumentParser()
ap.add_argument("--V", type=int, default=100_000)
ap.add_argument("--N", type=int, default=10_000_000)
ap.add_argument("--D", type=int, default=100)
ap.add_argument("--epochs", type=int, default=3)
ap.add_argument("--method", default="glove")
ap.add_argument("--oracle-threads", type=int, default=1)
ap.add_argument("--no-oracle", action="store_true")
ap.add_argument("--hot", default="auto")
a = ap.parse_args()

I, J, X, xmax = synth.synthetic_coo_shard(a.V, (0, a.V), a.N, seed=0xC0FFEE)
n = len(I)
cfg = geglove.Configuration({"graph": "s", "method": a.method, "dim": a.D, "threads": 1,
    "bca": {"alpha": .1, "epsilon": 1e-3}, "opt": {"method": "adagrad", "maxiter": a.epochs, "tolerance": 0},
    "output": {"uri": []}, "device": {"mode": "hogwild", "shuffle": "device", "seed": 42, "hot": a.hot}})
opt = geglove.Adagrad(geglove.CooMatrix(a.V, I, J, X, xmax), cfg, cfg.costFunction())
dev = []
print("info", opt.info(), flush=True)
for it in range(a.epochs):
    c = opt.epoch(it) / n
    dev.append(c)
    print("device epoch %d cost %.6f  (%.2f ms kernel)" % (it, c, opt.last_kernel_ms()[0]), flush=True)
if not a.no_oracle:
    kind = O.COST_GLOVE if a.method == "glove" else O.COST_PGLOVE
    ora = O.Glove(a.V, a.D, I, J, X, xmax, kind, seed=42, threads=a.oracle_threads)
    for it in range(a.epochs):
        t = time.time(); c = ora.epoch(race=a.oracle_threads > 1)
        print("oracle epoch %d cost %.6f  (%.1f s, T=%d)  device/oracle = %.3f" % (it, c, time.time() - t, a.oracle_threads, dev[it] / c), flush=True)
