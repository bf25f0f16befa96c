import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "graph-embeddings_amd"), os.path.join(REPO, "oracle"), os.path.join(REPO, "tests")]
import numpy as np
import geglove
from geglove import synth
import oracle as O
from helpers import make_config, OPT_KIND
V, N, D = 20000, 600000, 50
I, J, X, xmax = synth.synthetic_coo(V, N, seed=13); n = len(I)
for opt in ("adam", "amsgrad"):
    ora = O.Glove(V, D, I, J, X, xmax, O.COST_GLOVE, seed=42, threads=1, opt=OPT_KIND[opt])
    r = np.array([ora.epoch() for _ in range(10)]); print(opt, "oracle", np.round(r, 4))
    for hot in ("auto", "none"):
        for w in (0, 16):
            cfg = make_config(D, "glove", opt=opt, mode="hogwild", shuffle="device", seed=42, hot=hot, workers=w)
            dev = geglove.createOptimizer(cfg, geglove.CooMatrix(V, I, J, X, xmax))
            d = np.array([dev.epoch(it) / n for it in range(10)]); print(" ", hot, w, np.round(d, 4), flush=True)
