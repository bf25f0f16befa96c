import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "graph-embeddings_amd"), os.path.join(REPO, "oracle"), os.path.join(REPO, "tests")]
import numpy as np
import geglove
from geglove import synth
import oracle as O
from helpers import make_config, cost_kind
V, N, D = 20000, 600000, 50
I, J, X, xmax = synth.synthetic_coo(V, N, seed=13)
n = len(I)
for method in ("glove", "pglove"):
    ora = O.Glove(V, D, I, J, X, xmax, cost_kind(method), seed=42, threads=1)
    ref = np.array([ora.epoch() for _ in range(5)])
    print(method, "oracle", np.round(ref, 5))
    for hot in ("auto", "all"):
        cfg = make_config(D, method, mode="hogwild", shuffle="device", seed=42, hot=hot)
        opt = geglove.Adagrad(geglove.CooMatrix(V, I, J, X, xmax), cfg, cfg.costFunction())
        dev = np.array([opt.epoch(e) / n for e in range(5)])
        print(os.environ.get("GE_GLOVE_HOT_THETA"), hot, opt.info()["groups_in_flight"], opt.info()["hot_nonzeros"], np.round(dev / ref, 3), flush=True)
