"""Cost of the MERGED model after E epochs: one GPU process vs two ranks (sharing this GPU over gloo) with the
synchronous and with the overlapped context exchange.  The cost is evaluated on the host over all nonzeros
(GloveCost.java:9-20 without the update, fp64), so the three runs are compared on the same footing.
    python tools/exchange_quality.py [epochs=30] [V=100000] [nnz=13000000] [dim=64]"""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "graph-embeddings_amd"))


def model_cost(I, J, X, xmax, focus, context, fb, cb, block=1 << 21):
    tot = 0.0
    for a in range(0, len(I), block):
        i, j, x = I[a:a + block], J[a:a + block], X[a:a + block].astype(np.float64)
        inner = np.einsum("nd,nd->n", focus[i].astype(np.float64), context[j].astype(np.float64)) + fb[i] + cb[j] - np.log(x)
        tot += float(np.sum(0.5 * np.minimum(1.0, (x / xmax) ** 0.75) * inner * inner))
    return tot / len(I)


def rank_main(rank, world, port, q, exchange, V, N, D, epochs):
    import geglove
    from geglove import parallel, synth
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    I, J, X, xmax = synth.synthetic_coo(V, N, seed=13)
    rows = parallel.shard_rows(V, world, rank)
    si, sj, sx = parallel.shard_nonzeros(I, J, X, rows) if world > 1 else (I, J, X)
    cfg = geglove.Configuration({"graph": "synthetic", "method": "glove", "dim": D, "threads": 1,
                                 "bca": {"alpha": 0.1, "epsilon": 1e-3, "directed": True},
                                 "opt": {"method": "adagrad", "tolerance": 0, "maxiter": epochs}, "output": {"uri": []},
                                 "device": {"mode": "hogwild", "shuffle": "device", "seed": 42,
                                            "workers": -256 if exchange == "overlap" else 0,
                                            "row_range": rows if world > 1 else (0, 0)}})
    opt = geglove.createOptimizer(cfg, geglove.CooMatrix(V, si, sj, sx, xmax))
    sync = None
    if world > 1:
        dev = torch.device("cuda", 0)
        sync = parallel.context_sync_for(opt, dev, lazy_every=4)
    for it in range(epochs):
        opt.epoch(it)
        if sync is not None:
            sync.turn() if exchange == "overlap" else sync.sync()
    if sync is not None and exchange == "overlap":
        sync.replicate()
    torch.cuda.synchronize()
    q.put((rank, rows, opt.get_state("focus"), opt.get_state("fbias"), opt.get_state("context") if rank == 0 else None, opt.get_state("cbias") if rank == 0 else None))
    if world > 1:
        dist.barrier(); dist.destroy_process_group()
    opt.close()


def run(world, exchange, V, N, D, epochs):
    ctx = mp.get_context("spawn"); q = ctx.Queue()
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = [ctx.Process(target=rank_main, args=(r, world, port, q, exchange, V, N, D, epochs)) for r in range(world)]
    for p in procs: p.start()
    got = [q.get(timeout=900) for _ in range(world)]
    for p in procs: p.join(timeout=120)
    got.sort(key=lambda g: g[0])
    focus = np.concatenate([g[2].reshape(-1, D) for g in got]); fb = np.concatenate([g[3] for g in got])
    return focus, got[0][4].reshape(V, D), fb, got[0][5]


if __name__ == "__main__":
    from geglove import synth
    epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    V = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000
    N = int(sys.argv[3]) if len(sys.argv) > 3 else 13_000_000
    D = int(sys.argv[4]) if len(sys.argv) > 4 else 64
    I, J, X, xmax = synth.synthetic_coo(V, N, seed=13)
    base = None
    for world, exchange in ((1, "none"), (2, "sync"), (2, "overlap")):
        c = model_cost(I, J, X, xmax, *run(world, exchange, V, N, D, epochs))
        base = base or c
        print("%d rank(s), exchange %-7s: merged-model cost after %d epochs %.6f  (x %.3f of one GPU)" % (world, exchange, epochs, c, c / base), flush=True)
