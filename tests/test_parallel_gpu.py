"""Two ranks on ONE GPU (gloo over device tensors; RCCL refuses two ranks per device): exercises the real
multi-GPU plumbing of bench.py -- row-sharded handles (row_begin/row_end), zero-copy torch views of the
library's device tables, ContextSync on device memory -- and checks the sharded run against the oracle."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
V, N, D, EPOCHS = 6000, 300000, 32, 4


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _rank_main(rank, world, port, q):
    import geglove
    from geglove import parallel, synth
    from helpers import make_config
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    I, J, X, xmax = synth.synthetic_coo(V, N, seed=13)
    rows = parallel.shard_rows(V, world, rank)
    si, sj, sx = parallel.shard_nonzeros(I, J, X, rows)
    cfg = make_config(D, "glove", mode="hogwild", shuffle="device", seed=42, row_range=rows)
    opt = geglove.Adagrad(geglove.CooMatrix(V, si, sj, sx, xmax), cfg, cfg.costFunction())
    dev = torch.device("cuda", 0)

    def wrap(name):
        ptr, cnt = opt.device_ptr(name)
        return torch.as_tensor(parallel.DeviceArray(ptr, cnt), device=dev)

    ctx = wrap("context")
    assert ctx.data_ptr() == opt.device_ptr("context")[0]                       # zero copy
    sync = parallel.ContextSync(sums=[ctx], means=[wrap("cbias")], lazy_sums=[wrap("gsq_context"), wrap("gsq_cbias")], lazy_every=2)
    costs = []
    for it in range(EPOCHS):
        c = opt.epoch(it)
        sync.sync()
        t = torch.tensor([c], dtype=torch.float64); dist.all_reduce(t)
        costs.append(float(t.item()) / len(I))
    torch.cuda.synchronize()
    digest = torch.tensor([float(np.float64(opt.get_state(k).astype(np.float64).sum())) for k in ("context", "cbias", "gsq_context")], dtype=torch.float64)
    gathered = [torch.zeros_like(digest) for _ in range(world)]
    dist.all_gather(gathered, digest)
    fshape = opt.get_state("focus").shape[0]
    if rank == 0:
        q.put((costs, [g.tolist() for g in gathered], fshape, rows))
    opt.close()
    dist.destroy_process_group()


def test_two_ranks_share_one_gpu(gpu):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank_main, args=(r, 2, port, q)) for r in range(2)]
    for p in procs: p.start()
    costs, digests, fshape, rows = q.get(timeout=600)
    for p in procs: p.join(timeout=120)
    assert all(p.exitcode == 0 for p in procs)
    assert digests[0] == digests[1]                                  # replicated tables agree after every sync
    assert fshape == (rows[1] - rows[0]) * D                         # each rank holds only its focus rows
    import oracle as O
    from geglove import synth
    I, J, X, xmax = synth.synthetic_coo(V, N, seed=13)
    ora = O.Glove(V, D, I, J, X, xmax, O.COST_GLOVE, seed=42, threads=1)
    ref = [ora.epoch() for _ in range(EPOCHS)]
    assert costs[-1] < costs[0]
    np.testing.assert_allclose(costs, ref, rtol=0.10)
