"""Two ranks on ONE GPU: the multi-GPU trainer through the C ABI (ge_sync_*, csrc/sync.hip) with the host transport
(callbacks into torch.distributed / gloo on device memory -- RCCL refuses two ranks per device; the all-reduce itself is the
only thing RCCL would do differently).  Row-sharded handles (row_begin / row_end), the library's own take / land kernels on
its device tables (fat rows, interleaved records, bf16 rows + fp32 hub masters), both exchange forms; checked against the
sequential oracle (cost), against tests/sync_model.py (the merge rule, bit for bit) and for replica identity."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
EPOCHS = 4
CTX = ("context", "cbias", "gsq_context", "gsq_cbias")


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _spawn(target, args, world=2, timeout=900):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=target, args=(r, world, port, q) + tuple(args)) for r in range(world)]
    for p in procs: p.start()
    import queue, time
    t0, out = time.time(), None
    while out is None:
        try:
            out = q.get(timeout=5)
        except queue.Empty:
            assert all(p.is_alive() or p.exitcode == 0 for p in procs), "a rank died: %s" % [p.exitcode for p in procs]
            assert time.time() - t0 < timeout, "ranks timed out"
    for p in procs: p.join(timeout=120)
    assert all(p.exitcode == 0 for p in procs)
    return out


def _rank_main(rank, world, port, q, exchange, V, N, D, dtype, layout, wire):
    import geglove
    from geglove import parallel, synth
    from helpers import make_config
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    I, J, X, xmax = synth.synthetic_coo(V, N, seed=13)
    rows = parallel.shard_rows(V, world, rank)
    si, sj, sx = parallel.shard_nonzeros(I, J, X, rows)
    cfg = make_config(D, "glove", mode="hogwild", shuffle="device", seed=42, row_range=rows, dtype=dtype, layout=layout,
                      workers=-64 if exchange == "overlap" else 0)        # the overlapped form leaves slots to the collective
    opt = geglove.Adagrad(geglove.CooMatrix(V, si, sj, sx, xmax), cfg, cfg.costFunction())
    sync = parallel.context_sync_for(opt, torch.device("cuda", 0), lazy_every=2, wire=wire)     # gloo backend -> the callback transport
    costs, same_after_sync = [], []
    for it in range(EPOCHS):
        c = opt.epoch(it)
        sync.turn() if exchange == "overlap" else sync.sync()
        t = torch.tensor([c], dtype=torch.float64); dist.all_reduce(t)
        costs.append(float(t.item()) / len(I))
        if exchange == "sync" and it % 2 == 1:                          # accumulators were exchanged too (lazy_every = 2)
            dig = torch.tensor([float(opt.get_state(k).astype(np.float64).sum()) for k in CTX], dtype=torch.float64)
            both = [torch.zeros_like(dig) for _ in range(world)]; dist.all_gather(both, dig)
            same_after_sync.append([b.tolist() for b in both])
    c64 = opt.get_state("context").astype(np.float64)
    ctx_all = [torch.zeros(c64.size, dtype=torch.float64) for _ in range(world)]
    dist.all_gather(ctx_all, torch.from_numpy(c64.copy()))
    gap = float((ctx_all[0] - ctx_all[1]).abs().max()); scale = float(np.abs(c64).max())
    sync.replicate()
    digest = torch.tensor([float(opt.get_state(k).astype(np.float64).sum()) for k in CTX], dtype=torch.float64)
    gathered = [torch.zeros_like(digest) for _ in range(world)]
    dist.all_gather(gathered, digest)
    fshape = opt.get_state("focus").shape[0]
    info = opt.info()
    if rank == 0:
        q.put((costs, [g.tolist() for g in gathered], fshape, rows, same_after_sync, gap, scale, info))
    sync.close(); opt.close()
    dist.destroy_process_group()


CASES = [("sync", 6000, 300000, 32, "f32", [], "f32"), ("overlap", 6000, 300000, 32, "f32", [], "bf16"),
         ("sync", 40000, 2000000, 32, "f32", ["separate_tables"], "bf16"), ("overlap", 40000, 2000000, 32, "f32", [], "bf16"),
         ("overlap", 6000, 300000, 256, "f32", [], "bf16"),                 # dim 256: plain rows, separate bias tables
         ("sync", 20000, 1000000, 200, "f32", [], "f32"),      # BASELINE C4's shape (dim 200 fp32), reduced V
         ("overlap", 20000, 1000000, 200, "f32", ["separate_tables"], "bf16"),
         ("sync", 20000, 1000000, 300, "bf16", [], "bf16"),                # BASELINE C5's shape (dim 300, bf16 rows + fp32 accumulators)
         ("overlap", 20000, 1000000, 300, "bf16", [], "bf16"),
         ("overlap", 40000, 2000000, 32, "bf16", [], "bf16")]


@pytest.mark.parametrize("exchange,V,N,D,dtype,layout,wire", CASES)
def test_two_ranks_share_one_gpu(gpu, exchange, V, N, D, dtype, layout, wire):
    costs, digests, fshape, rows, same, gap, scale, info = _spawn(_rank_main, (exchange, V, N, D, dtype, layout, wire))
    if dtype == "f32":
        assert digests[0] == digests[1]                              # fp32 tables are identical after replicate()
        if wire == "f32":
            assert same and all(a == b for a, b in same)             # ... and after EVERY synchronous exchange on an fp32 wire: the tables are the consensus
    if exchange == "sync":
        assert gap <= scale * 2.0 ** -7                              # before replicate(): one bf16 rounding apart (overlap: plus a step's deltas in flight)
    assert fshape == (rows[1] - rows[0]) * D                         # each rank holds only its focus rows
    import oracle as O
    from geglove import synth
    I, J, X, xmax = synth.synthetic_coo(V, N, seed=13)
    ora = O.Glove(V, D, I, J, X, xmax, O.COST_GLOVE, seed=42, threads=1)
    ref = [ora.epoch() for _ in range(EPOCHS)]
    print("two ranks %s D=%d %s %s wire %s: cost / oracle %s" % (exchange, D, dtype, layout, wire, np.round(np.array(costs) / np.array(ref), 3).tolist()))
    assert costs[-1] < costs[0]
    # the overlapped form sees the other rank's first epoch only after its own second one: a bump there (on top of the
    # shift the blocked order itself gives the first two epochs), then it closes
    np.testing.assert_allclose(costs[:2], ref[:2], rtol=0.12 if exchange == "sync" else 0.20)
    np.testing.assert_allclose(costs[2:], ref[2:], rtol=0.06)


# ---- the library's exchange against the model, bit for bit --------------------------------------------------------------
def _model_rank_main(rank, world, port, q, D, layout, steps, with_hubs=False):
    """Each rank perturbs its context-side tables identically on the device (set_state) and on a host copy, then runs ge_sync
    on the device and SyncModel on the host: after every step the two must hold the same bits (fp32 wire)."""
    import geglove
    from geglove import parallel, synth
    from helpers import make_config
    from sync_model import SyncModel
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    V = 2000 if with_hubs else 300                                  # (busy columns: at least 256 nonzeros on a rank, so 1 000 rows per rank)
    I, J, X, xmax = synth.synthetic_coo(V, 150000 if with_hubs else 3000, seed=3)
    rows = parallel.shard_rows(V, world, rank)
    si, sj, sx = parallel.shard_nonzeros(I, J, X, rows)
    cfg = make_config(D, "glove", mode="hogwild", shuffle="device", seed=42, row_range=rows, layout=layout)
    opt = geglove.Adagrad(geglove.CooMatrix(V, si, sj, sx, xmax), cfg, cfg.costFunction())
    host = {k: torch.from_numpy(opt.get_state(k).copy()) for k in CTX}
    sync = parallel.context_sync_for(opt, torch.device("cuda", 0), lazy_every=2, wire="f32")
    hubs = sync.hub_rows() if with_hubs else None
    if with_hubs:
        assert len(hubs) > 0, "the test matrix has no hub columns"
    model = SyncModel(sums=[host["context"]], means=[host["cbias"]], lazy_sums=[host["gsq_context"], host["gsq_cbias"]], lazy_every=2, wire="f32")
    rng = np.random.default_rng(7 + rank)
    ok = True
    for step in range(steps):
        for k in CTX:                                                  # a "local pass": sparse moves, some elements untouched by either rank
            cur = opt.get_state(k)
            mv = (rng.standard_normal(cur.size) * 0.01 * (rng.random(cur.size) < 0.3)).astype(np.float32)
            new = (cur + mv).astype(np.float32)
            opt.set_state(k, new); host[k].copy_(torch.from_numpy(new))
        if hubs is not None:
            # ge_sync_epoch's order: the hub rows are reconciled by their small exchange (twice here, with moves in between), and
            # only rows that are NOT hubs still carry moves when the large exchange takes
            for rep in range(2):
                sync.hub_exchange(); model.hub_exchange(hubs, V)
                torch.cuda.synchronize()
                for k in CTX:
                    same = np.array_equal(opt.get_state(k), host[k].numpy())
                    if not same and os.environ.get("GE_TEST_DEBUG"):
                        a, b = opt.get_state(k).reshape(V, -1), host[k].numpy().reshape(V, -1)
                        bad = np.nonzero((a != b).any(axis=1))[0]
                        print("rank %d step %d hub exchange %d: %s differs in %d rows (hub rows among them: %d), max abs %.3g" % (rank, step, rep, k, len(bad), int(np.isin(bad, hubs).sum()), float(np.abs(a - b).max())), flush=True)
                    ok = ok and same
                keep = np.ones(V, bool); keep[hubs] = rep == 0            # first: everybody moves again; then: everybody but the hub rows
                for k in CTX:
                    cur = opt.get_state(k)
                    mv = (rng.standard_normal(cur.size) * 0.01 * (rng.random(cur.size) < 0.3)).astype(np.float32)
                    mv = (mv.reshape(V, -1) * keep[:, None]).reshape(-1).astype(np.float32)
                    new = (cur + mv).astype(np.float32)
                    opt.set_state(k, new); host[k].copy_(torch.from_numpy(new))
        if step % 3 == 2:
            sync.sync(); model.sync()
        else:
            sync.turn(); model.turn()
        torch.cuda.synchronize()
        for k in CTX:
            ok = ok and np.array_equal(opt.get_state(k), host[k].numpy())
    sync.replicate(); model.replicate()
    for k in CTX:
        ok = ok and np.array_equal(opt.get_state(k), host[k].numpy())
    flags = [torch.zeros(1) for _ in range(world)]
    dist.all_gather(flags, torch.tensor([1.0 if ok else 0.0]))
    if rank == 0:
        q.put([float(f.item()) for f in flags])
    sync.close(); opt.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("D,layout", [(32, []), (32, ["separate_tables"]), (256, []), (256, ["separate_tables"]), (200, [])])
def test_ge_sync_equals_the_model_bit_for_bit(gpu, D, layout):
    """Sum rule, mean rule (elements one rank, both ranks or no rank moved), lazy accumulators, take / land / turn / sync /
    replicate, over fat rows, interleaved records and plain rows with separate bias tables."""
    assert _spawn(_model_rank_main, (D, layout, 7)) == [1.0, 1.0]


@pytest.mark.parametrize("D,layout", [(32, []), (256, []), (200, [])])
def test_hub_exchange_equals_the_model_bit_for_bit(gpu, D, layout):
    """ge_sync_hub_exchange (k_hub_take / k_hub_land, what ge_sync_epoch runs behind every segment) between the large exchanges: the hub
    rows of all four tables become base + the summed deltas exactly, nothing else moves, and the large exchanges that follow stay
    bit-equal to the model (they find nothing left to do for those rows)."""
    assert _spawn(_model_rank_main, (D, layout, 7, True)) == [1.0, 1.0]


@pytest.mark.parametrize("D,world,dtype", [(32, 2, "f32"), (200, 3, "f32"), (32, 2, "bf16")])
def test_live_hub_exchange_equals_its_model_bit_for_bit(gpu, D, world, dtype):
    """ge_sync_hub_exchange_live (k_live_take / k_live_land: what ge_sync_epoch runs BESIDE the epoch kernel), here with nothing running
    beside it, so that it is a deterministic function: per live row and per element of the row and its accumulator row
    own = table - base, sum = own_0 + own_1 + ... (rank order, from 0.0f), merged = sum (the row table: x merge_scale of the accumulators),
    table += merged - own, base += merged -- the other ranks' moves are
    ADDED to whatever the table holds, nothing is stored over it; rows outside the live set, and every scalar, stay as they are (bf16
    handles: the live rows are fp32 master rows, the arithmetic is the same).  Two
    rounds with moves in between (the second one checks the base), then the exact exchange and replicate(): identical replicas."""
    import threading
    import geglove
    from geglove import parallel, synth
    from helpers import make_config
    V, N = 2000, 150000
    I, J, X, xmax = synth.synthetic_coo(V, N, seed=3)
    grp = parallel.LocalGroup(world)
    bar = threading.Barrier(world)
    snap = [dict() for _ in range(world)]          # per rank: what it held before / after every round
    err = [None] * world
    live = [None] * world

    def body(r):
        opt = sync = None
        try:
            rows = parallel.shard_rows(V, world, r)
            si, sj, sx = parallel.shard_nonzeros(I, J, X, rows)
            cfg = make_config(D, "glove", mode="hogwild", shuffle="device", seed=42, row_range=rows, dtype=dtype)
            opt = geglove.Adagrad(geglove.CooMatrix(V, si, sj, sx, xmax), cfg, cfg.costFunction())
            bar.wait(timeout=300)
            sync = parallel.ContextSync(opt, world, r, wire="f32", accum_every=2, local_group=grp)
            live[r] = sync.live_rows()
            rng = np.random.default_rng(11 + r)
            snap[r]["base"] = {k: opt.get_state(k).copy() for k in CTX}
            for rnd in range(2):
                for k in CTX:
                    cur = opt.get_state(k)
                    mv = (rng.standard_normal(cur.size) * 0.01 * (rng.random(cur.size) < 0.4)).astype(np.float32)
                    opt.set_state(k, (cur + mv).astype(np.float32))
                snap[r]["before%d" % rnd] = {k: opt.get_state(k).copy() for k in CTX}
                bar.wait(timeout=300)
                sync.hub_exchange_live()
                snap[r]["after%d" % rnd] = {k: opt.get_state(k).copy() for k in CTX}
            sync.hub_exchange(); sync.sync(); sync.replicate()
            snap[r]["final"] = {k: opt.get_state(k).copy() for k in CTX}
        except Exception as e:              # noqa: BLE001
            err[r] = e; grp.abort(); bar.abort()
        finally:
            if sync is not None: sync.close()
            if opt is not None: opt.close()

    th = [threading.Thread(target=body, args=(r,)) for r in range(world)]
    for t in th: t.start()
    for t in th: t.join(timeout=600)
    assert all(e is None for e in err), err
    grp.close()
    rows_live = live[0]
    assert len(rows_live) > 0 and all(np.array_equal(rows_live, l) for l in live)
    mask = np.zeros(V, bool); mask[rows_live] = True
    base = {k: snap[0]["base"][k].copy() for k in ("context", "gsq_context")}       # the same on every rank (same seed, nothing exchanged yet)
    inv_w = np.float32(1.0) / np.float32(world)
    for rnd in range(2):
        own, total = {}, {}
        for k in ("context", "gsq_context"):
            own[k] = [(snap[r]["before%d" % rnd][k] - base[k]).astype(np.float32) for r in range(world)]
            total[k] = np.zeros_like(own[k][0])
            for r in range(world):
                total[k] = (total[k] + own[k][r]).astype(np.float32)
        # the row table's sum is scaled by merge_scale of the accumulators' consensus BEFORE this exchange and their summed delta
        e = np.maximum(total["gsq_context"], np.float32(0))
        scale = np.sqrt(((base["gsq_context"] + e * inv_w).astype(np.float32) / (base["gsq_context"] + e).astype(np.float32)).astype(np.float32)).astype(np.float32)
        merged = {"context": (scale * total["context"]).astype(np.float32), "gsq_context": total["gsq_context"]}
        for k in ("context", "gsq_context"):
            for r in range(world):
                want = snap[r]["before%d" % rnd][k].copy().reshape(V, -1)
                add = (merged[k] - own[k][r]).astype(np.float32).reshape(V, -1)
                want[mask] = (want[mask] + add[mask]).astype(np.float32)
                assert np.array_equal(snap[r]["after%d" % rnd][k].reshape(V, -1), want), (rnd, k, r)
        for k in ("context", "gsq_context"):
            nb = base[k].reshape(V, -1).copy()
            nb[mask] = (nb[mask] + merged[k].reshape(V, -1)[mask]).astype(np.float32)
            base[k] = nb.reshape(-1)
        for k in ("cbias", "gsq_cbias"):                                              # the scalars are not the live exchange's
            for r in range(world):
                assert np.array_equal(snap[r]["after%d" % rnd][k], snap[r]["before%d" % rnd][k])
    for r in range(1, world):
        for k in CTX:
            if dtype == "bf16" and k == "context":            # bf16 rows live partly in per-rank fp32 master rows: equal up to one bf16 rounding
                np.testing.assert_allclose(snap[0]["final"][k], snap[r]["final"][k], rtol=2.0 ** -7, atol=1e-4)   # (random moves of 1e-2: an element near zero is a rounding of its neighbours' scale apart)
            else:
                assert np.array_equal(snap[0]["final"][k], snap[r]["final"][k]), (r, k)
    assert all(np.all(np.isfinite(snap[0]["final"][k])) for k in CTX)


def test_ge_sync_argument_errors(gpu):
    import ctypes as C
    import geglove
    from geglove import capi, synth
    from helpers import make_config
    I, J, X, xmax = synth.synthetic_coo(50, 400, seed=3)
    L = capi.lib()
    opt = geglove.Adagrad(geglove.CooMatrix(50, I, J, X, xmax), make_config(8, "glove", mode="hogwild"), geglove.GloveCost())
    h = C.c_void_p()
    cfg = capi.SyncCfg(); cfg.world, cfg.rank, cfg.wire = 2, 0, capi.GE_DTYPE_F32
    assert L.ge_sync_create(opt._h, C.byref(cfg), C.byref(h)) == capi.GE_ERR_ARG          # world 2 without transport or RCCL id
    assert b"RCCL unique id" in L.ge_last_error()
    cfg.world, cfg.rank = 1, 3
    assert L.ge_sync_create(opt._h, C.byref(cfg), C.byref(h)) == capi.GE_ERR_ARG
    cfg.world, cfg.rank = 1, 0
    capi.check(L.ge_sync_create(opt._h, C.byref(cfg), C.byref(h)))                        # world 1: every call is a no-op
    for fn in (L.ge_sync_turn, L.ge_sync_sync, L.ge_sync_finish):
        capi.check(fn(h))
    capi.check(L.ge_sync_replicate(h, 0))
    L.ge_sync_destroy(h)
    det = geglove.Adagrad(geglove.CooMatrix(50, I, J, X, xmax), make_config(8, "glove", mode="deterministic", shuffle="java"), geglove.GloveCost())
    assert L.ge_sync_create(det._h, C.byref(cfg), C.byref(h)) == capi.GE_ERR_STATE
    adam = geglove.createOptimizer(make_config(8, "glove", opt="adam", mode="hogwild"), geglove.CooMatrix(50, I, J, X, xmax))
    assert L.ge_sync_create(adam._h, C.byref(cfg), C.byref(h)) == capi.GE_ERR_STATE
    buf = (C.c_char * 128)()
    assert L.ge_rccl_unique_id(buf) == capi.GE_OK and any(buf.raw)                        # RCCL loads and hands out an id


def test_rccl_path_runs_on_one_gpu(gpu):
    """RCCL refuses two ranks on one device, so a 1-GPU box cannot run the two-rank exchange over it; what it can run is the
    library's RCCL binding itself: open librccl, ncclCommInitRank (one rank), ncclAllReduce + ncclBroadcast on a side stream,
    data intact.  The N > 1 data path over RCCL is the driver's 8-GPU bench (bench.py --gpus N)."""
    from geglove import capi
    capi.check(capi.lib().ge_rccl_selftest(0))


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


@pytest.mark.parametrize("pad", [0, 20])
@pytest.mark.parametrize("land,take", [(0, 1), (1, 0), (1, 1)])
def test_exchange_turn_bf16_matches_a_numpy_model(gpu, land, take, pad):
    """ge_exchange_turn_bf16 on a table whose rows 3, 10 and 11 are hubs (fp32 master rows) against the same arithmetic in numpy;
    pad > 0: the rows sit `dim + pad` elements apart (records), the elements between them must stay untouched."""
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
    S = D + pad
    wide = np.full((V, S), 0x7777, np.uint16); wide[:, :D] = t16.reshape(V, D)                  # the table as stored: rows S elements apart
    d_t, d_b, d_w, d_o, d_hub, d_idx = map(as_dev, (wide.reshape(-1).copy(), b.copy(), w16.copy(), o16.copy(), hub.copy(), hub_index))
    capi.check(capi.lib().ge_exchange_turn_bf16(d_t.data_ptr(), S, d_hub.data_ptr(), d_idx.data_ptr(), V, D, d_b.data_ptr(),
                                                d_w.data_ptr(), d_o.data_ptr(), land, take, seed, None))
    torch.cuda.synchronize()
    back = lambda x: x.cpu().numpy().view(np.uint16) if x.dtype == torch.int16 else x.cpu().numpy()
    stored = back(d_t).reshape(V, S)
    assert np.all(stored[:, D:] == 0x7777)                                                      # nothing outside the rows was touched
    d_t = torch.from_numpy(stored[:, :D].reshape(-1).copy().view(np.int16))
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
    np.testing.assert_array_equal(back(d_w), w16, err_msg="wire")        # the receive buffer of the all-reduce: never written here
    if take:
        np.testing.assert_array_equal(back(d_o), _np_bf16_rne(d), err_msg="own")
    else:
        np.testing.assert_array_equal(back(d_o), o16)
    if land:
        assert np.array_equal(back(d_t)[5 * D:6 * D], t16[5 * D:6 * D])   # nothing landed on row 5: its bf16 value is untouched by the rounding


# ---- eight ranks on ONE GPU: rank threads of this process meeting in a ge_local_group (VERDICT r02 #1b) ------------------
# Eight contributors through the library's own mean-over-contributors (cBias), summed rows / lazily summed accumulators and
# the bf16 wire with error feedback -- what an 8-GPU node runs, minus RCCL (the group sums in host memory, in rank order).
W8 = dict(world=8, V=100_000, N=3_000_000, D=200, epochs=7)
_ORACLE8 = {}


def _oracle8():
    if not _ORACLE8:
        import oracle as O
        from geglove import synth
        I, J, X, xmax = synth.synthetic_coo(W8["V"], W8["N"], seed=13)
        ora = O.Glove(W8["V"], W8["D"], I, J, X, xmax, O.COST_GLOVE, seed=42, threads=1)
        _ORACLE8["ref"] = [ora.epoch() for _ in range(W8["epochs"])]
        ora.close()
    return _ORACLE8["ref"]


def _run_rank_threads(world, V, N, D, epochs, exchange, wire, dtype="f32", lazy_every=2, fail_rank=None, hub_segments=None, plans=None, workers=None, own_streams=False):
    """One thread per rank; every rank owns a ge_glove handle (its block of focus rows) on device 0 and a ge_sync of the group."""
    import threading
    import geglove
    from geglove import parallel, synth
    from helpers import make_config
    I, J, X, xmax = synth.synthetic_coo(V, N, seed=13)
    if plans is not None: plans.append((I, J, X, xmax))
    grp = parallel.LocalGroup(world)
    bar = threading.Barrier(world)
    cost = np.zeros((epochs, world))
    out, err = [None] * world, [None] * world

    def body(r):
        opt = sync = None
        try:
            rows = parallel.shard_rows(V, world, r)
            si, sj, sx = parallel.shard_nonzeros(I, J, X, rows)
            # (library-default workers in both forms: the slots a real run reserves for RCCL's kernels, workers = -256, would leave a
            # matrix this small four workers and -- the hub set being relative to the worker count -- no hub columns at all)
            cfg = make_config(D, "glove", mode="hogwild", shuffle="device", seed=42, row_range=rows, dtype=dtype, **({"workers": workers} if workers else {}))
            # own_streams: every rank's handle on a stream of its own, as one rank per GPU has it (the ranks of this process otherwise all
            # launch on the null stream, where their epochs run one after the other)
            stream = torch.cuda.Stream(device=0) if own_streams else None
            opt = geglove.Adagrad(geglove.CooMatrix(V, si, sj, sx, xmax), cfg, cfg.costFunction(), **({"stream": stream.cuda_stream} if stream is not None else {}))
            bar.wait(timeout=600)
            sync = parallel.ContextSync(opt, world, r, wire=wire, accum_every=lazy_every, local_group=grp)
            if plans is not None and r == 0: plans.append(sync.hub_plan(hub_segments or 0))
            for it in range(epochs):
                if fail_rank == r and it == 1:
                    raise RuntimeError("rank %d leaves" % r)
                cost[it, r] = sync.epoch(it) if hub_segments is None else (sync.epoch(it, hub_segments) if hub_segments > 0 else opt.epoch(it))
                sync.turn() if (exchange == "overlap" and it >= 2) else sync.sync()      # (the hosts' policy: two synchronous epochs first)
            if plans is not None and r == 0: plans.append(sync.hub_plan(hub_segments or 0))
            sync.replicate()
            out[r] = {k: opt.get_state(k) for k in CTX}
        except Exception as e:              # noqa: BLE001 -- reported by the caller
            err[r] = e
            grp.abort()
            bar.abort()
        finally:
            if sync is not None: sync.close()
            if opt is not None: opt.close()

    th = [threading.Thread(target=body, args=(r,)) for r in range(world)]
    for t in th: t.start()
    for t in th: t.join(timeout=1500)
    alive = [t.is_alive() for t in th]
    grp_close = not any(alive)
    if grp_close: grp.close()
    return cost.sum(axis=1) / len(I), out, err, alive


@pytest.mark.parametrize("exchange,dtype,form", [("sync", "f32", "live"), ("overlap", "f32", "live"), ("sync", "f32", "segments"), ("overlap", "f32", "segments"), ("sync", "bf16", "segments"), ("sync", "bf16", "live")])
def test_eight_ranks_share_one_gpu(gpu, exchange, dtype, form, monkeypatch):
    """C4's shape (dim 200, fp32 rows, bf16 wire, accumulators every 2nd exchange) with EIGHT contributors per element, through
    ge_sync_epoch + the large exchange, in both forms of ge_sync_epoch: the hub rows exchanged LIVE beside the one launch of the epoch
    (what a run gets over RCCL and in a local group) and between the SEGMENTS of the epoch; and the same with bf16 rows (C5; live: on the fp32
    master rows of the columns that are hubs on every rank)."""
    if form == "segments":
        monkeypatch.setenv("GE_SYNC_EPOCH", "segments")
    plans = []
    costs, out, err, alive = _run_rank_threads(W8["world"], W8["V"], W8["N"], W8["D"], W8["epochs"], exchange, "bf16", dtype=dtype, plans=plans)
    assert not any(alive) and all(e is None for e in err), err
    assert len(plans) == 3 and plans[1]["live"] == (form == "live") and plans[1]["exchanges"] >= 8, plans[1:]
    if form == "live":
        assert plans[1]["live_rows"] > 0           # (eight ranks' epochs of a few ms on one GPU: the live form may hand over to segments, plans[2])
    for r in range(1, W8["world"]):
        for k in CTX:
            if dtype == "bf16" and k == "context":            # bf16 rows live partly in per-rank fp32 master rows: equal up to one bf16 rounding
                np.testing.assert_allclose(out[0][k], out[r][k], rtol=2.0 ** -7, atol=1e-5)
                continue
            assert np.array_equal(out[0][k], out[r][k]), "rank %d's %s differs from rank 0's after replicate()" % (r, k)
    ref = _oracle8()
    ratio = np.array(costs) / np.array(ref)
    print("eight ranks %s %s %s %s -> %s D=%d: cost / oracle %s" % (exchange, dtype, form, plans[1], plans[2], W8["D"], np.round(ratio, 3).tolist()))
    assert np.all(np.isfinite(costs)) and costs[-1] < costs[2] < costs[0]
    # bands: the blocked order and eight shards shift the first two epochs; from the third the sharded run tracks the
    # single-process oracle (synchronous: every rank sees the others' moves after each step; overlapped: one step late)
    np.testing.assert_allclose(costs[:2], ref[:2], rtol=0.25)
    # (the first overlapped epoch reads 8 - 10 % BELOW the oracle: a rank reports the training cost of its own shard against a context side
    # the synchronous epochs before it have just improved)
    np.testing.assert_allclose(costs[2:], ref[2:], rtol=0.06 if (exchange == "sync" and dtype == "f32") else (0.10 if exchange == "sync" else 0.13))


def _live_scenario():
    """Body of test_two_ranks_live_exchange_beside_a_running_epoch, run in a process of its own (prints one JSON line)."""
    import json
    import geglove
    from helpers import make_config
    V, N, D, E = 100_000, 30_000_000, 200, 5        # (epochs of ~ 10 ms per rank: an exchange through the host takes ~ 1 ms)
    plans = []
    # (a stream per rank, and 2 000 workers per rank: on a GPU of its own a rank's epoch leaves 128 wavefront slots to the small kernels; two
    # ranks that share one GPU must not fill it between them, or the small kernels queue behind the other rank's workgroups)
    costs, out, err, alive = _run_rank_threads(2, V, N, D, E, "sync", "bf16", plans=plans, hub_segments=4, workers=2000, own_streams=True)
    (I, J, X, xmax), before, after = plans
    same = bool(all(np.array_equal(out[0][k], out[1][k]) for k in CTX)) if out[0] is not None and out[1] is not None else False
    cfg = make_config(D, "glove", mode="hogwild", shuffle="device", seed=42)
    one = geglove.Adagrad(geglove.CooMatrix(V, I, J, X, xmax), cfg, cfg.costFunction())
    ref = [one.epoch(it) / len(I) for it in range(E)]
    one.close()
    print("LIVE " + json.dumps({"alive": [bool(a) for a in alive], "err": [None if e is None else str(e) for e in err], "before": before, "after": after,
                                "replicas_identical": same, "costs": [float(c) for c in costs], "single_handle": [float(c) for c in ref]}), flush=True)


def test_two_ranks_live_exchange_beside_a_running_epoch(gpu):
    """The live form for real: two rank threads with 15 M nonzeros each (epochs of several milliseconds), four exchanges of the hub rows per
    epoch BESIDE the epoch kernel -- take, host sum, atomic land on a stream of their own while the kernel's workers publish their own
    deltas into the same rows.  The run must stay live (its exchanges keep the epoch's pace), end in identical replicas and follow a
    single-handle run of the same library on the whole matrix (same seed; the sharded run's order differs, so: a band).  In a process of
    its own: whether an exchange is late is a matter of host timing, and a test process that has run hundreds of GPU tests is no measure."""
    import json
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, "-c", "import conftest, test_parallel_gpu as T; T._live_scenario()"], cwd=here, capture_output=True, text=True, timeout=360,
                       env=dict(os.environ, GE_SYNC_DEBUG="1", GE_LOCAL_GROUP_TIMEOUT_S="120"))
    line = [l for l in r.stdout.splitlines() if l.startswith("LIVE ")]
    assert r.returncode == 0 and line, (r.returncode, r.stdout[-2000:], r.stderr[-4000:])
    d = json.loads(line[-1][5:])
    assert not any(d["alive"]) and all(e is None for e in d["err"]), d
    assert d["before"]["live"] and d["before"]["exchanges"] == 4 and d["before"]["live_rows"] > 0, d["before"]
    assert d["after"]["live"], "the live exchange fell behind the epoch and handed over to segments:\n" + r.stderr[-6000:]
    assert d["replicas_identical"]
    costs, ref = np.array(d["costs"]), np.array(d["single_handle"])
    print("two ranks live: cost / single handle %s (%d live rows)" % (np.round(costs / ref, 3).tolist(), d["before"]["live_rows"]))
    assert np.all(np.isfinite(costs)) and costs[-1] < costs[1] < costs[0]
    np.testing.assert_allclose(costs[2:], ref[2:], rtol=0.06)
    np.testing.assert_allclose(costs[:2], ref[:2], rtol=0.15)


def test_eight_ranks_need_the_hub_rows_reconciled_inside_the_epoch(gpu):
    """What ge_sync_epoch is for: the same eight-rank run with ONE exchange per epoch for every row (ge_glove_epoch + ge_sync_sync)
    overshoots on the busiest context rows -- eight ranks each push them for a whole epoch from the same start, the pushes add up --
    and leaves the oracle's trajectory within a few epochs (measured: NaN in epoch 7; the CPU simulation of the merge rule shows the
    same, tests/tools/multirank_sim.py), while with the hub rows reconciled between the segments of the epoch it follows it."""
    costs, out, err, alive = _run_rank_threads(W8["world"], W8["V"], W8["N"], W8["D"], W8["epochs"], "sync", "bf16", hub_segments=0)
    assert not any(alive) and all(e is None for e in err), err
    ref = _oracle8()
    ratio = np.array(costs) / np.array(ref)
    print("eight ranks, one exchange per epoch for every row: cost / oracle %s" % np.round(ratio, 3).tolist())
    assert not np.all(np.isfinite(costs)) or ratio[-1] > 1.15


def test_a_failing_rank_releases_its_local_group(gpu):
    """a rank that dies must not leave its peers inside the group's barrier.  Rank 1 leaves before its second epoch;
    ranks 0 and 2 get GE_ERR_STATE from their next exchange and every thread ends."""
    costs, out, err, alive = _run_rank_threads(3, 3000, 60_000, 16, 4, "sync", "f32", fail_rank=1)
    assert not any(alive)
    assert isinstance(err[1], RuntimeError)
    from geglove import capi
    for r in (0, 2):
        assert isinstance(err[r], capi.GeError) and err[r].status == capi.GE_ERR_STATE, err


def test_bench_starts_its_own_ranks(gpu):
    """`python3 bench.py --gpus 2` from a plain shell (no torchrun): the launcher starts both ranks before any GPU call.  On this
    one-GPU box they share device 0 over gloo (GE_BENCH_ONE_DEVICE / GE_BENCH_BACKEND, rehearsal switches); the line says n_gpus 2
    and carries both exchange forms and the per-step cost of the whole job."""
    import json, subprocess, sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(GE_BENCH_BACKEND="gloo", GE_BENCH_ONE_DEVICE="1")
    r = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--rows-per-gpu", "20000", "--nnz-per-gpu", "1500000"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["config"]["vocab"] == 40000
    assert len(d["mean_cost_per_step"]) == 4 and d["mean_cost_per_step"][-1] < d["mean_cost_per_step"][0]
    assert d["exchange"]["form"] == "overlap" and d["exchange"]["other_form"]["form"] == "sync"
    assert len(d["exchange"]["other_form"]["mean_cost_per_step"]) == 3
