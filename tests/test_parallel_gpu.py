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
D, EPOCHS = 32, 4


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _rank_main(rank, world, port, q, exchange, V, N):
    import geglove
    from geglove import parallel, synth
    from helpers import make_config
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    I, J, X, xmax = synth.synthetic_coo(V, N, seed=13)
    rows = parallel.shard_rows(V, world, rank)
    si, sj, sx = parallel.shard_nonzeros(I, J, X, rows)
    cfg = make_config(D, "glove", mode="hogwild", shuffle="device", seed=42, row_range=rows,
                      workers=-64 if exchange == "overlap" else 0)        # the overlapped form leaves slots to the collective
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
        if exchange == "overlap":
            sync.turn()                                                             # lands one step late, as bench.py does
        else:
            sync.sync()
        t = torch.tensor([c], dtype=torch.float64); dist.all_reduce(t)
        costs.append(float(t.item()) / len(I))
    fused = [e["fused"] for e in sync._entries()] if exchange == "overlap" else []
    if exchange == "overlap":
        sync.replicate()
    torch.cuda.synchronize()
    digest = torch.tensor([float(np.float64(opt.get_state(k).astype(np.float64).sum())) for k in ("context", "cbias", "gsq_context")], dtype=torch.float64)
    gathered = [torch.zeros_like(digest) for _ in range(world)]
    dist.all_gather(gathered, digest)
    fshape = opt.get_state("focus").shape[0]
    if rank == 0:
        q.put((costs, [g.tolist() for g in gathered], fshape, rows, fused))
    opt.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("exchange,V,N", [("sync", 6000, 300000), ("overlap", 6000, 300000),
                                          ("sync", 40000, 2000000), ("overlap", 40000, 2000000)])   # 40000 x 32 floats: bf16 wire, fused device pass
def test_two_ranks_share_one_gpu(gpu, exchange, V, N):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank_main, args=(r, 2, port, q, exchange, V, N)) for r in range(2)]
    for p in procs: p.start()
    costs, digests, fshape, rows, fused = q.get(timeout=600)
    for p in procs: p.join(timeout=120)
    assert all(p.exitcode == 0 for p in procs)
    assert digests[0] == digests[1]                                  # replicated tables agree after every sync
    if exchange == "overlap":
        assert fused == [V * D >= (1 << 20), V * D >= (1 << 20), False, False]   # context, gsq_context | gsq_cbias, cbias
    assert fshape == (rows[1] - rows[0]) * D                         # each rank holds only its focus rows
    import oracle as O
    from geglove import synth
    I, J, X, xmax = synth.synthetic_coo(V, N, seed=13)
    ora = O.Glove(V, D, I, J, X, xmax, O.COST_GLOVE, seed=42, threads=1)
    ref = [ora.epoch() for _ in range(EPOCHS)]
    assert costs[-1] < costs[0]
    # the overlapped form sees the other rank's first epoch only after its own second one: a bump there, then it closes
    np.testing.assert_allclose(costs, ref, rtol=0.10 if exchange == "sync" else 0.15)
    np.testing.assert_allclose(costs[-1], ref[-1], rtol=0.05)


@pytest.mark.parametrize("n", [8 * 4096, 8 * 4096 + 5, 3, 1_000_003])
@pytest.mark.parametrize("land,take", [(0, 1), (1, 0), (1, 1)])
def test_exchange_turn_kernel_matches_the_written_out_passes(gpu, n, land, take):
    """ge_exchange_turn against the same arithmetic in torch fp32 (bit-exact: one subtraction, one addition, RNE narrowing)."""
    from geglove import capi
    g = torch.Generator().manual_seed(n * 4 + land * 2 + take)
    dev = torch.device("cuda", 0)
    # 16-byte aligned views even for ragged n: allocate whole tensors
    t = torch.randn(n, generator=g).to(dev); b = (t + 0.01 * torch.randn(n, generator=g).to(dev)).contiguous()
    w = (0.02 * torch.randn(n, generator=g)).to(torch.bfloat16).to(dev); own = (0.01 * torch.randn(n, generator=g)).to(torch.bfloat16).to(dev)
    t0, b0, w0, o0 = t.clone(), b.clone(), w.clone(), own.clone()
    capi.check(capi.lib().ge_exchange_turn(t.data_ptr(), b.data_ptr(), w.data_ptr(), own.data_ptr(), n, land, take,
                                           torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    d = (t0 - b0).to(torch.bfloat16)
    r = w0.float() - o0.float()
    t_ref = t0 + r if land else t0
    if take:
        b_ref, w_ref, o_ref = t_ref, d, d
    else:
        b_ref, w_ref, o_ref = b0 + r, w0, o0
    for name, got, ref in (("table", t, t_ref), ("base", b, b_ref), ("wire", w, w_ref), ("own", own, o_ref)):
        assert torch.equal(got, ref), name


def test_exchange_turn_rejects_bad_arguments(gpu):
    from geglove import capi
    t = torch.zeros(64, device="cuda")
    w = torch.zeros(64, dtype=torch.bfloat16, device="cuda")
    L = capi.lib()
    assert L.ge_exchange_turn(None, t.data_ptr(), w.data_ptr(), w.data_ptr(), 64, 1, 1, None) == capi.GE_ERR_ARG
    assert L.ge_exchange_turn(t.data_ptr() + 4, t.data_ptr(), w.data_ptr(), w.data_ptr(), 8, 1, 1, None) == capi.GE_ERR_ARG
    assert L.ge_exchange_turn(t.data_ptr(), t.data_ptr(), w.data_ptr(), w.data_ptr(), -1, 1, 1, None) == capi.GE_ERR_ARG
    assert L.ge_exchange_turn(t.data_ptr(), t.data_ptr(), w.data_ptr(), w.data_ptr(), 0, 1, 1, None) == capi.GE_OK
