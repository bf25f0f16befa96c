"""Does the blocked epoch order train the model the reference's order trains -- at the BENCH's own size?  The sequential oracle
cannot run 103 M nonzeros x D = 200 for ten epochs in test time, so both runs are the device's Hogwild kernel, same matrix, same
seed, same init: (a) shuffle: device (chunks of whole rows / hub columns, permuted per epoch -- the headline path), (b) shuffle:
java (a uniformly random permutation of the nonzeros per epoch from the reference's own Fisher-Yates stream; chunks are 128
consecutive nonzeros of it).  Path (b) is tied to the sequential oracle at small sizes by tests/test_glove_parity_gpu.py
(general-order trajectory within 2 %).  Prints per-epoch mean costs, their ratio, and the pairwise-cosine correlation of the two
embeddings over sampled vertices.      python3 tools/r02/order_equivalence.py [epochs] [nnz]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "graph-embeddings_amd"))
import geglove                          # noqa: E402
from geglove import synth               # noqa: E402

epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 10
nnz = int(sys.argv[2]) if len(sys.argv) > 2 else 125_000_000
V, D = 625_000, 200
I, J, X, xmax = synth.synthetic_coo_shard(V, (0, V), nnz, seed=0xC0FFEE)
n = len(I)
out = {"V": V, "nnz": n, "dim": D, "epochs": epochs}
E_ = {}
init = None
for name, shuffle, seed in (("device", "device", 42), ("java", "java", 42), ("java_other_permutations", "java", 43)):
    cfg = geglove.Configuration({"graph": "synthetic", "method": "glove", "dim": D, "threads": 1,
                                 "bca": {"alpha": 0.1, "epsilon": 1e-3, "directed": True},
                                 "opt": {"method": "adagrad", "tolerance": 0, "maxiter": epochs}, "output": {"uri": []},
                                 "device": {"mode": "hogwild", "shuffle": shuffle, "seed": seed}})
    opt = geglove.createOptimizer(cfg, geglove.CooMatrix(V, I, J, X, xmax))
    if init is None:
        init = opt.state()                       # seed 42's parameters: every run starts from them
    else:
        for k, v in init.items():
            opt.set_state(k, v)                  # (the third run keeps seed 43's permutation stream: the noise floor of "another order")
    shuffle = name
    costs, t0 = [], time.perf_counter()
    for it in range(epochs):
        costs.append(opt.epoch(it) / n)
    out[shuffle] = {"cost": [round(c, 6) for c in costs], "seconds": round(time.perf_counter() - t0, 1), "kernel_ms_last": round(opt.last_kernel_ms()[0], 2),
                    "workers": opt.info()["groups_in_flight"]}
    E_[shuffle] = opt.extractResultF32().reshape(V, D)
    opt.close()
    print(shuffle, out[shuffle], flush=True)
rng = np.random.default_rng(1)
pick = rng.choice(V, 2000, replace=False)
def cosm(E):
    nrm = E / np.maximum(np.linalg.norm(E, axis=1, keepdims=True), 1e-30)
    return nrm @ nrm.T
iu = np.triu_indices(len(pick), 1)
out["cost_ratio_device_over_java"] = [round(a / b, 4) for a, b in zip(out["device"]["cost"], out["java"]["cost"])]
out["cost_ratio_other_java_over_java"] = [round(a / b, 4) for a, b in zip(out["java_other_permutations"]["cost"], out["java"]["cost"])]
cj = cosm(E_["java"][pick])[iu]
out["pairwise_cosine_correlation_device_vs_java"] = round(float(np.corrcoef(cosm(E_["device"][pick])[iu], cj)[0, 1]), 5)
out["pairwise_cosine_correlation_java_vs_other_java"] = round(float(np.corrcoef(cosm(E_["java_other_permutations"][pick])[iu], cj)[0, 1]), 5)
print(json.dumps(out), flush=True)
