#!/usr/bin/env python3
"""Would it pay to walk the matrix in COLUMN WINDOWS small enough for the streamed records to stay in the 256 MB Infinity Cache?
Emulated without touching the layout: a block-diagonal matrix (B blocks of V/B rows whose columns lie in the same block), trained once
in the library's order (chunks handed out through the keyed bijection: the workers in flight stream records of all B blocks, 1 GB) and
once with GE_ORDER_WINDOWS=B (the bijection permutes inside B consecutive windows of the chunk list: the workers in flight are all in one
block, so what they stream is V/B records).  Hub columns off and on (with them on, the hub chunks -- the head of the chunk list -- all
fall into the first window).
GE_ORDER_WINDOWS lived in a probe build only (commit 99f9383: a branch at the chunk fetch of k_adagrad_runs; it put spill slots under
the bench instance, which tests/test_kernel_resources.py refuses, and the idea lost, so it was taken out again) -- both legs of
profiles/r03_window_probe.jsonl ran on that one binary.
   python3 tools/r03/window_probe.py [blocks] [dim]"""
import json, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "graph-embeddings_amd"))
import numpy as np
import geglove
from geglove import synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
D = int(sys.argv[2]) if len(sys.argv) > 2 else 200
V, N = 625_000, 125_000_000
Vb = V // B
parts = [synth.synthetic_coo_shard(Vb, (0, Vb), N // B, seed=0xC0FFEE + b) for b in range(B)]
I = np.concatenate([p[0] + np.int32(b * Vb) for b, p in enumerate(parts)])
J = np.concatenate([p[1] + np.int32(b * Vb) for b, p in enumerate(parts)])
X = np.concatenate([p[2] for p in parts])
del parts
co = geglove.CooMatrix(Vb * B, I, J, X, 0.2)
for hot, ident in (("none", "0"), ("none", "1"), ("auto", "0"), ("auto", "1")):
    os.environ["GE_ORDER_WINDOWS"] = str(B) if ident == "1" else "0"
    cfg = geglove.Configuration({"graph": "s", "method": "glove", "dim": D, "threads": 1, "bca": {"alpha": 0.1, "epsilon": 1e-3},
                                 "opt": {"method": "adagrad", "tolerance": 0, "maxiter": 6}, "output": {"uri": []},
                                 "device": {"mode": "hogwild", "shuffle": "device", "seed": 42, "hot": hot}})
    opt = geglove.createOptimizer(cfg, co)
    ms, cost = [], []
    for it in range(5):
        cost.append(opt.epoch(it) / len(I)); ms.append(round(opt.last_kernel_ms()[0], 2))
    info = opt.info()
    print(json.dumps({"blocks": B, "dim": D, "nonzeros": int(len(I)), "hot": hot, "windows": B if ident == "1" else 1, "streamed_window_MB": round((V if ident == "0" else Vb) * info["row_stride"] * 4 / 1e6),
                      "epoch_ms": ms, "mean_cost": [round(c, 6) for c in cost], "hub_columns": info["hot_columns"]}), flush=True)
    opt.close()
