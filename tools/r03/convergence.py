#!/usr/bin/env python3
"""Long-run convergence of the production path against the sequential reference order (Optimizer.java:96-107).

BASELINE C2 (V = 100 k, 10.0 M nonzeros, dim 100, glove, AdaGrad), 40+ epochs:
  oracle leg  (CPU, no GPU needed; run once, result committed under profiles/): tests/tools/convergence_oracle.py (the scripts that load
      the CPU oracle live under tests/): python3 tests/tools/convergence_oracle.py --epochs 48 --out profiles/r03_convergence_oracle.npz
      -- the sequential restatement of Adagrad.createJob, Java order (Fisher-Yates per epoch, seed 42): per-epoch mean cost and the
      final vectors of a fixed vertex sample.
  device leg  (GPU): python3 tools/r03/convergence.py device --ref profiles/r03_convergence_oracle.npz --out profiles/r03_convergence.json
      the Hogwild kernel, blocked device shuffle, library-default workers, same matrix and seed: per-epoch cost ratio, the epoch at
      which |prev - cur| <= tolerance fires on each side for the shipped tolerance 1e-4 (and for 1e-5, 1e-6), pairwise-cosine
      correlation of the sampled vertices' final vectors.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "graph-embeddings_amd"))

ap = argparse.ArgumentParser()
ap.add_argument("leg", choices=["device"])
ap.add_argument("--V", type=int, default=100_000)
ap.add_argument("--N", type=int, default=12_100_000)        # the generator drops duplicate (i, j): 10.0 M remain
ap.add_argument("--D", type=int, default=100)
ap.add_argument("--epochs", type=int, default=48)
ap.add_argument("--sample", type=int, default=2000)
ap.add_argument("--ref")
ap.add_argument("--device-cfg", default="", help="device leg: extra keys of the YAML's device: block, k=v,k=v (hot_theta, stale_budget, workers ...)")
ap.add_argument("--out", required=True)
a = ap.parse_args()

from geglove import synth            # noqa: E402  (numpy only; no GPU call)
I, J, X, xmax = synth.synthetic_coo_shard(a.V, (0, a.V), a.N, seed=0xC0FFEE)
n = len(I)
sample = np.sort(np.random.default_rng(7).choice(a.V, a.sample, replace=False))


def stop_epoch(costs, tol):
    """Optimizer.java:96-107: stop when |prev - cur| <= tol, prev starting at 0; 1-based epoch, None if it never fires."""
    prev = 0.0
    for k, c in enumerate(costs):
        if abs(prev - c) <= tol:
            return k + 1
        prev = c
    return None


def cos_upper(E):
    nrm = E / np.maximum(np.linalg.norm(E, axis=1, keepdims=True), 1e-30)
    return (nrm @ nrm.T)[np.triu_indices(len(E), 1)]


if True:
    import geglove
    ref = np.load(a.ref)
    assert int(ref["nnz"]) == n and int(ref["V"]) == a.V and int(ref["D"]) == a.D and np.array_equal(ref["sample"], sample)
    epochs = min(a.epochs, len(ref["costs"]))
    cfg = geglove.Configuration({"graph": "s", "method": "glove", "dim": a.D, "threads": 1, "bca": {"alpha": .1, "epsilon": 1e-3},
                                 "opt": {"method": "adagrad", "maxiter": epochs, "tolerance": 0}, "output": {"uri": []},
                                 "device": dict({"mode": "hogwild", "shuffle": "device", "seed": 42},
                                                **{k: (float(v) if "." in v else int(v)) for k, v in (kv.split("=") for kv in a.device_cfg.split(",") if kv)})})
    opt = geglove.Adagrad(geglove.CooMatrix(a.V, I, J, X, xmax), cfg, cfg.costFunction())
    dev, kms = [], []
    for it in range(epochs):
        dev.append(opt.epoch(it) / n); kms.append(opt.last_kernel_ms()[0])
    E = opt.extractResult().reshape(a.V, a.D)[sample]
    rc = ref["costs"][:epochs]
    rho = float(np.corrcoef(cos_upper(E), cos_upper(ref["vectors"].astype(np.float64)))[0, 1])
    out = {"device_cfg": a.device_cfg, "kernel_ms_median": float(np.median(kms)), "workload": "BASELINE C2: V=%d, %d nonzeros, dim %d, glove, AdaGrad, seed 42" % (a.V, n, a.D), "epochs": epochs,
           "workers": opt.info()["groups_in_flight"],
           "device_cost": dev, "oracle_cost": rc.tolist(), "device_over_oracle": (np.array(dev) / rc).tolist(),
           "oracle_threads": int(ref["threads"]),
           "tolerance_stop_epoch": {str(t): {"device": stop_epoch(dev, t), "oracle": stop_epoch(rc.tolist(), t)} for t in (1e-4, 1e-5, 1e-6)},
           "epochs_behind": [float(np.interp(d, rc[::-1], np.arange(epochs, 0, -1.0)) - (k + 1)) for k, d in enumerate(dev)],
           "pairwise_cosine_correlation_final": rho, "sample_vertices": int(a.sample)}
    json.dump(out, open(a.out, "w"), indent=1)
    print(json.dumps({k: out[k] for k in ("tolerance_stop_epoch", "pairwise_cosine_correlation_final")}))
    print("device/oracle:", np.round(out["device_over_oracle"], 4).tolist())
