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


def _rank_main(rank, world, port, q, exchange, V, N, D=D):
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
    sync = parallel.context_sync_for(opt, dev, lazy_every=2)
    assert sync._entries()[0]["t"].data_ptr() == opt.device_ptr("context")[0]   # zero copy: the library's own table
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
    torch.cuda.synchronize()
    c64 = opt.get_state("context").astype(np.float64)
    pre = torch.tensor([float(c64.sum()), float(np.abs(c64).sum())], dtype=torch.float64)
    pre_all = [torch.zeros_like(pre) for _ in range(world)]
    dist.all_gather(pre_all, pre)
    sync.replicate()                    # lands what is in flight (overlap) and makes the replicas bit-identical
    torch.cuda.synchronize()
    digest = torch.tensor([float(np.float64(opt.get_state(k).astype(np.float64).sum())) for k in ("context", "cbias", "gsq_context")], dtype=torch.float64)
    gathered = [torch.zeros_like(digest) for _ in range(world)]
    dist.all_gather(gathered, digest)
    fshape = opt.get_state("focus").shape[0]
    if rank == 0:
        q.put((costs, [g.tolist() for g in gathered], fshape, rows, fused, [x.tolist() for x in pre_all]))
    opt.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("exchange,V,N,D", [("sync", 6000, 300000, 32), ("overlap", 6000, 300000, 32),
                                            ("sync", 40000, 2000000, 32), ("overlap", 40000, 2000000, 32),   # 40000 x 32 floats: bf16 wire, fused device pass
                                            ("overlap", 6000, 300000, 256)])                                  # dim 256: plain rows, separate bias tables
def test_two_ranks_share_one_gpu(gpu, exchange, V, N, D):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank_main, args=(r, 2, port, q, exchange, V, N, D)) for r in range(2)]
    for p in procs: p.start()
    costs, digests, fshape, rows, fused, pre = q.get(timeout=600)
    for p in procs: p.join(timeout=120)
    assert all(p.exitcode == 0 for p in procs)
    assert digests[0] == digests[1]                                  # replicated tables are identical after replicate()
    if exchange == "sync":                                           # ... and before it only the bf16 rounding of the last deltas apart
        assert abs(pre[0][0] - pre[1][0]) <= 1e-5 * pre[0][1]           # difference of the sums against the sum of magnitudes
    if exchange == "overlap":
        if D % 256:
            assert fused == [True, V * (D + 4) >= (1 << 20), False]               # context rows (fat), accumulator table, cbias column
        else:
            assert fused == [V * D >= (1 << 20)] * 2 + [False, False]             # plain rows: context, gsq_context | gsq_cbias, cbias
    assert fshape == (rows[1] - rows[0]) * D                         # each rank holds only its focus rows
    import oracle as O
    from geglove import synth
    I, J, X, xmax = synth.synthetic_coo(V, N, seed=13)
    ora = O.Glove(V, D, I, J, X, xmax, O.COST_GLOVE, seed=42, threads=1)
    ref = [ora.epoch() for _ in range(EPOCHS)]
    assert costs[-1] < costs[0]
    # the overlapped form sees the other rank's first epoch only after its own second one: a bump there (on top of the
    # shift the blocked order itself gives the first two epochs), then it closes
    np.testing.assert_allclose(costs[:2], ref[:2], rtol=0.12 if exchange == "sync" else 0.20)
    np.testing.assert_allclose(costs[2:], ref[2:], rtol=0.05)


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
    b_ref = b0 + r if land else b0
    w_ref, o_ref = w0, o0
    if take:
        b_ref, w_ref, o_ref = b_ref + d.float(), d, d                 # the base advances by what is sent
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


def _np_bf16_rne(x):
    u = x.astype(np.float32).view(np.uint32).astype(np.uint64)
    return (((u + 0x7FFF + ((u >> 16) & 1)) >> 16) & 0xFFFF).astype(np.uint16)


def _np_bf16_to_f32(h):
    return (h.astype(np.uint32) << 16).view(np.float32)


def _np_mix32(x):
    x = x.astype(np.uint64)
    M = np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(15); x = (x * np.uint64(0x85EBCA77)) & M
    x ^= x >> np.uint64(13); x = (x * np.uint64(0xC2B2AE3D)) & M
    x ^= x >> np.uint64(16)
    return x


@pytest.mark.parametrize("land,take", [(0, 1), (1, 0), (1, 1)])
def test_exchange_turn_bf16_matches_a_numpy_model(gpu, land, take):
    """ge_exchange_turn_bf16 on a table whose rows 3, 10 and 11 are hubs (fp32 master rows) against the same arithmetic in numpy."""
    from geglove import capi
    rng = np.random.default_rng(land * 2 + take)
    V, D, seed = 37, 12, 0xABCDEF
    hub_index = np.full(V, -1, np.int32); hub_index[[3, 10, 11]] = [0, 1, 2]
    t16 = _np_bf16_rne(rng.standard_normal(V * D).astype(np.float32))
    hub = rng.standard_normal(3 * D).astype(np.float32)
    rows = np.repeat(np.arange(V), D)
    is_hub = hub_index[rows] >= 0
    hpos = (hub_index[rows].astype(np.int64) * D + np.tile(np.arange(D), V))[is_hub]
    t = _np_bf16_to_f32(t16).copy(); t[is_hub] = hub[hpos]                 # the row values
    b = (t + 0.01 * rng.standard_normal(V * D)).astype(np.float32)
    w16 = _np_bf16_rne(0.02 * rng.standard_normal(V * D).astype(np.float32)); o16 = _np_bf16_rne(0.01 * rng.standard_normal(V * D).astype(np.float32))
    w16[5 * D:6 * D] = o16[5 * D:6 * D]                                  # row 5: nobody else moved it
    dev = torch.device("cuda", 0)
    as_dev = lambda a: torch.from_numpy(a.view(np.int16) if a.dtype == np.uint16 else a).to(dev)
    d_t, d_b, d_w, d_o, d_hub, d_idx = map(as_dev, (t16.copy(), b.copy(), w16.copy(), o16.copy(), hub.copy(), hub_index))
    capi.check(capi.lib().ge_exchange_turn_bf16(d_t.data_ptr(), d_hub.data_ptr(), d_idx.data_ptr(), V, D, d_b.data_ptr(),
                                                d_w.data_ptr(), d_o.data_ptr(), land, take, seed, None))
    torch.cuda.synchronize()
    back = lambda x: x.cpu().numpy().view(np.uint16) if x.dtype == torch.int16 else x.cpu().numpy()
    # ---- model ----
    d = (t - b).astype(np.float32)
    r = (_np_bf16_to_f32(w16) - _np_bf16_to_f32(o16)).astype(np.float32)
    tn, bn = t.copy(), b.copy()
    exp_t16, exp_hub = t16.copy(), hub.copy()
    if land:
        tn = (t + r).astype(np.float32)
        bn = (b + r).astype(np.float32)
        rnd = _np_mix32(((np.arange(V * D, dtype=np.uint64) * np.uint64(0x9E3779B1)) + np.uint64(seed)) & np.uint64(0xFFFFFFFF)) >> np.uint64(16)
        stored16 = (((tn.view(np.uint32).astype(np.uint64) + rnd) >> np.uint64(16)) & np.uint64(0xFFFF)).astype(np.uint16)
        exp_t16 = np.where(is_hub, t16, stored16)                          # hub rows: the bf16 copy is not touched
        exp_hub[hpos] = tn[is_hub]
    if take:
        bn = (bn + _np_bf16_to_f32(_np_bf16_rne(d))).astype(np.float32)
    np.testing.assert_array_equal(back(d_t), exp_t16, err_msg="table")
    np.testing.assert_array_equal(back(d_b), bn, err_msg="base")
    np.testing.assert_array_equal(back(d_hub), exp_hub, err_msg="hub rows")
    if take:
        np.testing.assert_array_equal(back(d_w), _np_bf16_rne(d), err_msg="wire")
        np.testing.assert_array_equal(back(d_o), _np_bf16_rne(d), err_msg="own")
    else:
        np.testing.assert_array_equal(back(d_w), w16); np.testing.assert_array_equal(back(d_o), o16)
    if land:
        assert np.array_equal(back(d_t)[5 * D:6 * D], t16[5 * D:6 * D])   # nothing landed on row 5: its bf16 value is untouched by the rounding


def _bf16_rank_main(rank, world, port, q, exchange, V, N):
    import geglove
    from geglove import parallel, synth
    from helpers import make_config
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    I, J, X, xmax = synth.synthetic_coo(V, N, seed=13)
    rows = parallel.shard_rows(V, world, rank)
    si, sj, sx = parallel.shard_nonzeros(I, J, X, rows)
    cfg = make_config(D, "glove", mode="hogwild", shuffle="device", seed=42, row_range=rows, dtype="bf16")
    opt = geglove.Adagrad(geglove.CooMatrix(V, si, sj, sx, xmax), cfg, cfg.costFunction())
    dev = torch.device("cuda", 0)

    def wrap(name):
        ptr, cnt = opt.device_ptr(name)
        return torch.as_tensor(parallel.DeviceArray(ptr, cnt), device=dev)

    sync = parallel.context_sync_for(opt, dev, lazy_every=2)
    ctx = sync.bf16[0]
    costs = []
    for it in range(EPOCHS):
        c = opt.epoch(it)
        sync.turn() if exchange == "overlap" else sync.sync()
        t = torch.tensor([c], dtype=torch.float64); dist.all_reduce(t)
        costs.append(float(t.item()) / len(I))
    if exchange == "overlap":
        sync.replicate()
    torch.cuda.synchronize()
    context = opt.get_state("context").astype(np.float64)
    other = torch.from_numpy(context.copy()); gathered = [torch.zeros_like(other) for _ in range(world)]
    dist.all_gather(gathered, other)
    hubs = ctx.hub_index.cpu().to(torch.int64); hub_g = [torch.zeros_like(hubs) for _ in range(world)]
    dist.all_gather(hub_g, hubs)
    if rank == 0:
        diff = np.abs(gathered[0].numpy() - gathered[1].numpy()).reshape(V, D)
        worst = np.argsort(diff.max(axis=1))[-5:]
        if os.environ.get("GE_TEST_DEBUG"):
            for v in worst:
                d = int(diff[v].argmax())
                print("row", v, "hub idx", int(hub_g[0][v]), int(hub_g[1][v]), "diff", diff[v].max(), "values", gathered[0].numpy().reshape(V, D)[v, d], gathered[1].numpy().reshape(V, D)[v, d],
                      "q99.9 of all", np.quantile(diff, 0.999), flush=True)
        q.put((costs, float(diff.max()), float(np.abs(context).max()), ctx.n_hub))
    opt.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("exchange", ["sync", "overlap"])
def test_two_ranks_with_bf16_rows(gpu, exchange):
    """BASELINE C5's storage (bf16 rows + fp32 accumulators) sharded over two ranks: the cost follows the single-process
    fp32 oracle and the two context replicas agree to bf16 precision."""
    V, N = 40000, 2000000
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bf16_rank_main, args=(r, 2, port, q, exchange, V, N)) for r in range(2)]
    for p in procs: p.start()
    costs, gap, scale, n_hub = q.get(timeout=600)
    for p in procs: p.join(timeout=120)
    assert all(p.exitcode == 0 for p in procs)
    import oracle as O
    from geglove import synth
    I, J, X, xmax = synth.synthetic_coo(V, N, seed=13)
    ora = O.Glove(V, D, I, J, X, xmax, O.COST_GLOVE, seed=42, threads=1)
    ref = [ora.epoch() for _ in range(EPOCHS)]
    assert n_hub > 0                                                    # the hub rows (fp32 masters) took part
    np.testing.assert_allclose(costs, ref, rtol=0.10 if exchange == "sync" else 0.15)
    np.testing.assert_allclose(costs[-1], ref[-1], rtol=0.05)
    assert gap <= scale * 2.0 ** -8                                     # replicas: one bf16 rounding of a delta / of a stored value apart


@pytest.mark.parametrize("rows,stride,cols", [(1000, 36, 32), (257, 204, 200), (300, 11, 7), (5, 4, 4)])
@pytest.mark.parametrize("land,take", [(0, 1), (1, 0), (1, 1)])
def test_exchange_turn_rows_touches_only_the_row_part(gpu, rows, stride, cols, land, take):
    """ge_exchange_turn_rows on a table of fat rows: columns < cols behave like ge_exchange_turn, the bias column and the
    padding keep table / base and get zero wire / own slots on a take (4-wide and scalar variants)."""
    from geglove import capi
    g = torch.Generator().manual_seed(rows * 7 + stride + land * 2 + take)
    dev = torch.device("cuda", 0)
    n = rows * stride
    t = torch.randn(n, generator=g).to(dev); b = (t + 0.01 * torch.randn(n, generator=g).to(dev)).contiguous()
    w = (0.02 * torch.randn(n, generator=g)).to(torch.bfloat16).to(dev); own = (0.01 * torch.randn(n, generator=g)).to(torch.bfloat16).to(dev)
    t0, b0, w0, o0 = t.clone(), b.clone(), w.clone(), own.clone()
    capi.check(capi.lib().ge_exchange_turn_rows(t.data_ptr(), b.data_ptr(), w.data_ptr(), own.data_ptr(), rows, stride, cols, land, take,
                                                torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    part = (torch.arange(n, device=dev) % stride) < cols
    d = (t0 - b0).to(torch.bfloat16)
    r = w0.float() - o0.float()
    t_ref = torch.where(part, t0 + r if land else t0, t0)
    b_ref = b0 + r if land else b0
    w_ref, o_ref = w0, o0
    if take:
        b_ref = b_ref + d.float()
        w_ref = torch.where(part, d, torch.zeros_like(d)); o_ref = w_ref
    b_ref = torch.where(part, b_ref, b0)
    for name, got, ref in (("table", t, t_ref), ("base", b, b_ref), ("wire", w, w_ref), ("own", own, o_ref)):
        assert torch.equal(got, ref), name
