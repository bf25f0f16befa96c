#!/usr/bin/env python3
"""What a segment of ge_sync_epoch costs on one GPU: rank 0 holds (almost) the whole bench-size shard, rank 1 a sliver (so that it does
not compete for the GPU), both threads of this process (ge_local_group: the hub rows meet in host memory).  Per S = exchanges per epoch:
the epoch's wall time on rank 0, the sum of its segment kernels' own times (ge_glove_last_kernel_ms), and a plain ge_glove_epoch beside.
   python3 tools/r03/segment_cost.py [live|segments]      (the form of ge_sync_epoch: GE_SYNC_EPOCH)"""
import json, os, sys, threading, time
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "graph-embeddings_amd"))
import numpy as np
import geglove
from geglove import parallel, synth

FORM = sys.argv[1] if len(sys.argv) > 1 else "live"
os.environ["GE_SYNC_EPOCH"] = FORM
V, D, N = 625_000, 200, 125_000_000
SLIVER = 2_000
ROWS = [(0, V - SLIVER), (V - SLIVER, V)]
grp = parallel.LocalGroup(2)
bar = threading.Barrier(2)
res, err = {}, [None, None]
SEGS = [0, 8, 16, 32, 64, 128, 8] if FORM == "live" else [0, 8, 16, 32, 64, 8]

def body(r):
    try:
        rows = ROWS[r]
        n_local = int(N * (rows[1] - rows[0]) / V)
        I, J, X, xmax = synth.synthetic_coo_shard(V, rows, n_local, seed=0xC0FFEE)
        cfg = geglove.Configuration({"graph": "s", "method": "glove", "dim": D, "threads": 1, "bca": {"alpha": 0.1, "epsilon": 1e-3},
                                     "opt": {"method": "adagrad", "tolerance": 0, "maxiter": 99}, "output": {"uri": []},
                                     "device": {"mode": "hogwild", "shuffle": "device", "seed": 42, "row_range": list(rows)}})
        opt = geglove.createOptimizer(cfg, geglove.CooMatrix(V, I, J, X, xmax))
        bar.wait(timeout=600)
        sync = parallel.ContextSync(opt, 2, r, wire="bf16", accum_every=4, local_group=grp)
        it = 0
        for S in SEGS:
            wall, kern = [], []
            for rep in range(3):
                bar.wait(timeout=600)
                t = time.perf_counter()
                c = sync.epoch(it, S) if S > 0 else opt.epoch(it)
                wall.append((time.perf_counter() - t) * 1e3)
                ms, nl = opt.last_kernel_ms()
                kern.append(ms)
                sync.sync(); it += 1
            if r == 0:
                print(json.dumps({"form": FORM, "plan": sync.hub_plan(S) if S else None, "segments": S if S else "plain ge_glove_epoch", "rank0_nonzeros": int(len(I)), "hub_rows": int(len(sync.hub_rows())), "launches": nl,
                                  "epoch_wall_ms": [round(x, 2) for x in wall], "kernels_sum_ms": [round(x, 2) for x in kern], "mean_cost": round(c / len(I), 5)}), flush=True)
        sync.close(); opt.close()
    except Exception as e:
        err[r] = e; grp.abort(); bar.abort()

th = [threading.Thread(target=body, args=(r,)) for r in range(2)]
for t in th: t.start()
for t in th: t.join()
if any(err): raise SystemExit(str(err))
