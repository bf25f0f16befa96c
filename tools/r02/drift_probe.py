"""Does the epoch time follow the box's memory system?  One process, ~2 minutes: per sample a 2 GiB device copy, a random
816-byte-row gather (torch), one epoch of the bench-size handle, and rocm-smi's clocks / power.   python3 tools/r02/drift_probe.py"""
import json
import os
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "graph-embeddings_amd"))
import geglove                          # noqa: E402
from geglove import synth               # noqa: E402

V, D, nnz = 625_000, 200, 125_000_000
cache = "/tmp/ge_mode_probe_%d_%d.npz" % (V, nnz)
if os.path.exists(cache):
    z = np.load(cache); I, J, X, xmax = z["I"], z["J"], z["X"], float(z["xmax"])
else:
    I, J, X, xmax = synth.synthetic_coo_shard(V, (0, V), nnz, seed=0xC0FFEE)
    np.savez(cache, I=I, J=J, X=X, xmax=xmax)
dev = torch.device("cuda", 0)
src = torch.empty(512 * 1024 * 1024, dtype=torch.float32, device=dev).normal_()
dst = torch.empty_like(src)
table = torch.empty(2_500_000, 204, dtype=torch.float32, device=dev).normal_()      # 2 GB of 816-byte rows
idx = torch.randint(0, table.shape[0], (4_000_000,), device=dev)
cfg = geglove.Configuration({"graph": "synthetic", "method": "glove", "dim": D, "threads": 1,
                             "bca": {"alpha": 0.1, "epsilon": 1e-3, "directed": True},
                             "opt": {"method": "adagrad", "tolerance": 0, "maxiter": 1000}, "output": {"uri": []},
                             "device": {"mode": "hogwild", "shuffle": "device", "seed": 42}})
opt = geglove.createOptimizer(cfg, geglove.CooMatrix(V, I, J, X, xmax))


def timed(fn, reps=3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(); torch.cuda.synchronize()
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def smi():
    try:
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showtemp", "--json"], capture_output=True, text=True, timeout=20).stdout
        d = json.loads(out); c = d[sorted(d)[0]]
        keep = {}
        for k, v in c.items():
            kl = k.lower()
            if any(t in kl for t in ("sclk", "mclk", "fclk", "socclk", "power", "junction", "memory")): keep[k] = v
        return keep
    except Exception as e:
        return {"error": str(e)}


t_start = time.time()
it = 0
for s in range(70):
    copy_ms = timed(lambda: dst.copy_(src))
    gather_ms = timed(lambda: torch.index_select(table, 0, idx))
    opt.epoch(it); it += 1
    ep = opt.last_kernel_ms()[0]
    rec = {"t": round(time.time() - t_start, 1), "copy_TBps": round(2 * src.numel() * 4 / copy_ms / 1e9, 3),
           "gather_TBps": round(2 * idx.numel() * 816 / gather_ms / 1e9, 3), "epoch_ms": round(ep, 2)}
    if s % 5 == 0: rec["smi"] = smi()
    print(json.dumps(rec), flush=True)
    if 25 <= s < 40: time.sleep(2.0)          # an idle stretch in the middle
opt.close()
