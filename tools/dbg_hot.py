import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "graph-embeddings_amd"), os.path.join(REPO, "oracle"), os.path.join(REPO, "tests")]
import numpy as np
import geglove
from geglove import synth
from helpers import make_config
for (V, N, D) in [(10, 100, 8), (10, 128, 200), (10, 500, 200), (50, 3000, 200), (2000, 60000, 50)]:
    I, J, X, xmax = synth.synthetic_coo(V, N, seed=13)
    res = {}
    for hot in ("none", "all"):
        cfg = make_config(D, "glove", mode="hogwild", shuffle="none", seed=42, hot=hot)
        opt = geglove.Adagrad(geglove.CooMatrix(V, I, J, X, xmax), cfg, cfg.costFunction())
        costs = [opt.epoch(e) / len(I) for e in range(3)]
        res[hot] = (costs, opt.state())
    d = {k: float(np.max(np.abs(res["none"][1][k] - res["all"][1][k]))) for k in res["none"][1]}
    print(V, len(I), D, "none", res["none"][0], "all", res["all"][0], "maxdiff", d, flush=True)
