import sys; sys.path[:0]=['/root/repo/graph-embeddings_amd','/root/repo/oracle','/root/repo/tests']
import numpy as np, geglove, oracle as O
from geglove import synth
from helpers import make_config, OPT_KIND
V, N, D = 90, 2500, 52
I, J, X, xmax = synth.synthetic_coo(V, N, seed=17)
for layout in ([], ["fixed_cuts"]):
  for opt in ("adam","amsgrad"):
    cfg = make_config(D, "glove", opt=opt, mode="hogwild", shuffle="device", seed=5, hot="none", workers=1, hot_theta=0.02, layout=layout)
    dev = geglove.createOptimizer(cfg, geglove.CooMatrix(V, I, J, X, xmax))
    ref = {k: np.ascontiguousarray(v.reshape(V, -1) if v.size == V * D else v, np.float32) for k, v in dev.state().items()}
    for it in range(3):
        order = dev.epoch_order(it).astype(np.int64)
        cost = dev.epoch(it)
        job = O.opt_job(OPT_KIND[opt], it, D, I[order], J[order], X[order], xmax, O.COST_GLOVE, ref)
        out=[]
        for name, got in dev.state().items():
            g, r = got.reshape(-1), np.asarray(ref[name]).reshape(-1)
            err = np.abs(g - r) / (np.abs(r) + 1e-3 * np.max(np.abs(r)))
            out.append("%s med %.1e q95 %.1e max %.1e" % (name, np.median(err), np.quantile(err,0.95), np.max(err)))
        print(layout, opt, it, cost/N, float(job)/N, out[:2], flush=True)
