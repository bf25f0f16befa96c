"""The BASELINE.json configurations at (or near) their full sizes, under -m gpu.

C2  synthetic 100 k vertices / 10 M nonzeros / dim 100 fp32, library-default workers: cost trajectory against the
    sequential oracle, tolerance stated per epoch.
C3' DBLP-like graph (a stand-in: the DBLP data is not in the reference) through the device builder at 50 k vertices,
    dim 200, the shipped YAML's pglove: the TRAINED VECTORS of the production (Hogwild, blocked order) path against the
    sequential oracle's, through the pairwise-cosine matrix of a vertex sample, plus the final cost.
Lost runs: no focus row is resident in two workers at once -- measured, and shown to be what the
    round-1 layout (layout: fixed_cuts) loses.
The oracle is single-threaded here; these are the slow tests of the suite (about a minute in all)."""
import os

import numpy as np
import pytest

import geglove
from geglove import synth
import oracle as O
from helpers import make_config

pytestmark = pytest.mark.gpu


def test_c2_full_size_cost_trajectory(gpu):
    """BASELINE C2.  Per-epoch mean cost, device (thousands of racing workers, blocked order) / sequential oracle
    (Java order, same seed), TEN epochs: within 10 % in the first two (the blocked order alone shifts them), within 4 % after."""
    V, D = 100_000, 100
    I, J, X, xmax = synth.synthetic_coo_shard(V, (0, V), 12_100_000, seed=0xC0FFEE)     # the generator drops duplicate (i, j): 10.0 M remain
    n = len(I)
    assert 9_900_000 < n < 10_100_000
    cfg = make_config(D, "glove", mode="hogwild", shuffle="device", seed=42)
    opt = geglove.Adagrad(geglove.CooMatrix(V, I, J, X, xmax), cfg, cfg.costFunction())
    info = opt.info()
    assert info["groups_in_flight"] >= 1024 and info["long_rows"] >= 0
    EP = 10
    dev = np.array([opt.epoch(it) / n for it in range(EP)])
    # the sequential oracle's trajectory on this matrix and seed is a committed fixture (tests/golden/c2_oracle_costs.npz, written by
    # tests/tools/convergence_oracle.py: 20 s per epoch on one core); its first two epochs are recomputed here to tie the file to the code
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "c2_oracle_costs.npz"))
    assert int(gold["nnz"]) == n and int(gold["V"]) == V and int(gold["D"]) == D
    ora = O.Glove(V, D, I, J, X, xmax, O.COST_GLOVE, seed=42, threads=1)
    live = np.array([ora.epoch() for _ in range(2)])
    np.testing.assert_array_equal(live, gold["costs"][:2])
    ref = gold["costs"][:EP]
    print("C2 device/oracle per epoch:", np.round(dev / ref, 4).tolist(), "workers", info["groups_in_flight"])
    assert np.all(np.isfinite(dev)) and np.all(np.diff(dev) < 0)
    np.testing.assert_allclose(dev[:2], ref[:2], rtol=0.10)
    np.testing.assert_allclose(dev[2:], ref[2:], rtol=0.04)       # the lag grows to 6.8 % by epoch 27 and holds there


def _cos_matrix(E):
    n = E / np.maximum(np.linalg.norm(E, axis=1, keepdims=True), 1e-30)
    return n @ n.T


def test_c3_standin_trained_vectors_at_dim_200(gpu):
    """DBLP-like graph, 50 k vertices -> ge_bca_build (bit-exact vs the oracle's builder) -> pglove, dim 200, Hogwild with
    the library's worker count.  Vectors are compared through the pairwise-cosine matrix of 1500 sampled vertices
    (invariant to what SGD leaves undetermined): correlation with the sequential Java-order oracle >= 0.995, final
    cost within 3 %."""
    g = synth.dblp_like_graph(10000, 15000, 40)
    V, D, EP = g["V"], 200, 4
    assert V >= 50_000
    cfgb = make_config(D, "pglove")
    bca = geglove.BookmarkColoring(g, cfgb)
    ref = O.bca_build(V, g["out"], g["inn"], 0.1, 1e-3, True, O.NORM_NONE)
    np.testing.assert_array_equal(bca.I, ref["I"]); np.testing.assert_array_equal(bca.J, ref["J"])
    assert np.array_equal(bca.X.view(np.uint32), ref["X"].view(np.uint32)) and bca.max() == ref["max"]
    I, J, X, xmax = bca.I, bca.J, bca.X, bca.max()
    n = len(I)
    cfg = make_config(D, "pglove", mode="hogwild", shuffle="device", seed=42)
    opt = geglove.Adagrad(bca, cfg, cfg.costFunction())
    info = opt.info()
    for it in range(EP):
        c_dev = opt.epoch(it) / n
    E_dev = opt.extractResult().reshape(V, D)
    ora = O.Glove(V, D, I, J, X, xmax, O.COST_PGLOVE, seed=42, threads=1)
    for _ in range(EP):
        c_ref = ora.epoch()
    E_ref = ora.extract()
    pick = np.random.default_rng(1).choice(V, 1500, replace=False)
    iu = np.triu_indices(len(pick), 1)
    rho = np.corrcoef(_cos_matrix(E_ref[pick])[iu], _cos_matrix(E_dev[pick])[iu])[0, 1]
    print("C3 stand-in: %d vertices, %d nonzeros, %d workers, %d long rows; cost device %.6f oracle %.6f; cosine correlation %.5f"
          % (V, n, info["groups_in_flight"], info["long_rows"], c_dev, c_ref, rho))
    assert info["long_rows"] > 0                                      # rows of several hundred entries exist: the delta-publishing pieces ran
    assert abs(c_dev / c_ref - 1) <= 0.03, (c_dev, c_ref)
    assert rho >= 0.995, rho


def _private_columns_matrix(rows, per_row, seed):
    """`rows` focus rows of `per_row` nonzeros each, every nonzero on a column of its own: no context row is ever shared,
    so what happens to focus row i depends on row i's own updates alone -- whatever the other workers do."""
    rng = np.random.default_rng(seed)
    n = rows * per_row
    I = np.repeat(np.arange(rows, dtype=np.int32), per_row)
    J = rng.permutation(n).astype(np.int32)
    X = (10 ** rng.uniform(-3.5, -0.7, size=n)).astype(np.float32)
    return n, I, J, X, 0.2                                             # vocabulary = n: one column per nonzero


@pytest.mark.parametrize("per_row,dtype", [(200, "f32"), (90, "f32"), (200, "bf16"), (90, "bf16"), (700, "f32")])
def test_no_focus_row_is_resident_in_two_workers(gpu, per_row, dtype):
    """Known limit 1 of round 1: a focus row cut by a chunk boundary could be resident in two workers at once; the later
    store discarded the other worker's WHOLE run (up to 128 updates and their gradSq increments).
    Made exactly countable: 3000 rows, every nonzero on a private column, learning rate 0 and the bias accumulators at
    1e30, so no parameter moves and the weighted cost wc of every nonzero is a constant.  gradSqFocus[i] then grows by
    sum over row i's nonzeros of (wc * context)^2 -- the same number whatever the order and the concurrency, unless an
    update's increment is discarded.
      * default layout, 2048 racing workers: every row's growth equals the one-worker run's (fp32 summation order only;
        90 per row = rows packed whole, bit for bit; 200 / 700 per row = 2 / 6 pieces that publish by delta).
      * layout: fixed_cuts (round 1) on the same matrix loses whole runs -- which is what makes this test a test."""
    rows, D = 3000, 64
    V, I, J, X, xmax = _private_columns_matrix(rows, per_row, seed=per_row)

    def run(layout, workers):
        cfg = make_config(D, "glove", mode="hogwild", shuffle="device", seed=42, hot="none", workers=workers, layout=layout, dtype=dtype,
                          learning_rate=0.0)
        opt = geglove.Adagrad(geglove.CooMatrix(V, I, J, X, xmax), cfg, cfg.costFunction())
        opt.set_state("gsq_fbias", np.full(V, 1e30, np.float32)); opt.set_state("gsq_cbias", np.full(V, 1e30, np.float32))
        f0 = opt.get_state("focus")
        opt.epoch(0)
        assert np.array_equal(opt.get_state("focus"), f0)               # nothing moves
        return (opt.get_state("gsq_focus").reshape(V, D)[:rows].astype(np.float64) - 1).sum(1), opt.info()

    g1, _ = run([], 1)
    gp, info = run([], 2048)
    go, _ = run(["fixed_cuts"], 2048)
    print("per_row %d %s: gradSq growth / one worker: default layout %.6f .. %.6f (long rows %d, shared chunks %d); fixed cuts min %.3f, %d rows below 0.9"
          % (per_row, dtype, (gp / g1).min(), (gp / g1).max(), info["long_rows"], info["shared_chunks"], (go / g1).min(), ((go / g1) < 0.9).sum()))
    assert (info["long_rows"] == rows) == (per_row > 128)
    np.testing.assert_allclose(gp, g1, rtol=1e-4)                       # no update's increment was discarded
    assert ((go / g1) < 0.9).sum() > 20                                 # the round-1 layout does lose runs here


def test_record_tables_are_placed_by_probing_and_the_choice_changes_no_result(gpu):
    """ge_glove_create allocates a large record table several times and keeps the candidate whose probe -- records read and written
    back unchanged at random rows -- runs fastest.  The search is reported, `layout: first_placement` turns it off, and
    the handle that went through it holds exactly the state of the one that did not (the probe touches the tables before they are
    initialised and writes back what it read): one worker, same seed, bit for bit after an epoch."""
    V, D = 60000, 200                                           # 60 000 records of 1 664 bytes: 100 MB per side, above the 64 MB floor
    I, J, X, xmax = synth.synthetic_coo(V, 400000, seed=23)

    def run(layout):
        cfg = make_config(D, "glove", mode="hogwild", shuffle="device", seed=42, workers=1, layout=layout)
        opt = geglove.createOptimizer(cfg, geglove.CooMatrix(V, I, J, X, xmax))
        info = opt.info()
        cost = opt.epoch(0)
        state = opt.state()
        opt.close()
        return info, cost, state

    i0, c0, s0 = run(["first_placement"])
    i1, c1, s1 = run([])
    assert i0["placements"] == 2 and i0["placement_best_ms"] == 0.0
    assert 4 <= i1["placements"] <= 12 and 0.0 < i1["placement_best_ms"] <= i1["placement_worst_ms"]
    assert c0 == c1
    for name in s0:
        np.testing.assert_array_equal(s0[name].view(np.uint32), s1[name].view(np.uint32), err_msg=name)


# ---- BASELINE C4 / C5 at their own vocabulary on ONE GPU -----------------------------------------------------
# V = 5 M: a record table is 5 M x 416 floats = 2.08e9 elements (fp32, dim 200) -- every row offset past row 5.16 M / 2 needs the
# 64-bit row arithmetic of k_init_java, k_adagrad_runs, k_extract and the state copies.  The oracle cannot hold 16 GB of tables in
# test time, so the checks are the size-independent ones: the Java draw order on sampled rows (an independent Python model of
# java.util.Random with the jump the kernel uses), one visit per nonzero, a finite and falling cost, extract = (focus + context) / 2.
_MASK48 = (1 << 48) - 1


def _java_rows(seed, D, rows, bf16=False):
    """Optimizer.java:50-57 for the given rows: per row fB, cB, then foc[d], ctx[d] interleaved, each f(f(d(nextFloat()) - 0.5) / f(D))."""
    a, c = 0x5DEECE66D, 0xB
    out = {}
    for r in rows:
        n = (2 + 2 * D) * int(r)                    # draws before this row
        A, Cc, s = 1, 0, (seed ^ a) & _MASK48        # x -> A x + Cc  composed n times by squaring
        ba, bc = a, c
        while n:
            if n & 1:
                A, Cc = (ba * A) & _MASK48, (ba * Cc + bc) & _MASK48
            ba, bc = (ba * ba) & _MASK48, (ba * bc + bc) & _MASK48
            n >>= 1
        s = (A * s + Cc) & _MASK48
        vals = np.empty(2 + 2 * D, np.float32)
        for k in range(2 + 2 * D):
            s = (s * a + c) & _MASK48
            nf = np.float32((s >> 24) / float(1 << 24))
            vals[k] = np.float32(np.float32(np.float64(nf) - 0.5) / np.float32(D))
        row = {"fbias": vals[0], "cbias": vals[1], "focus": vals[2::2].copy(), "context": vals[3::2].copy()}
        if bf16:
            for k in ("focus", "context"):
                u = row[k].view(np.uint32).astype(np.uint64)
                row[k] = ((((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16) & 0xFFFFFFFF).astype(np.uint32).view(np.float32)
        out[int(r)] = row
    return out


@pytest.mark.parametrize("D,dtype,V", [(200, "f32", 5_200_000), (300, "bf16", 5_000_000)])
def test_c4_c5_vocabulary_on_one_gpu(gpu, D, dtype, V):
    """fp32 rows: C4's own V = 5 M gives 5 M x 416 = 2.08e9 floats per record table, just UNDER 2^31 elements (its byte offsets
    pass 2^32 at row 2.58 M); V = 5.2 M (2.16e9 floats, still V * D < 2^31 as the Java arrays need) puts element offsets past
    2^31 as well.  bf16 rows at C5's V = 5 M: 4.64e9 bf16 elements per record table."""
    rng = np.random.default_rng(3)
    # a hub-heavy matrix over the WHOLE id range (rows and columns up to V - 1) + a conflict-free batch to count visits with
    n_cf = 200_000
    ci, cj, cx = synth.conflict_free_batch(V, n_cf, seed=21)
    I, J, X, xmax = synth.synthetic_coo_shard(V, (0, V), 12_000_000, seed=0xC0FFEE)      # the 5 M diagonal entries + 7 M hub-heavy ones
    assert I.max() == V - 1 and J.max() == V - 1 and len(I) > 11_000_000
    cfg = make_config(D, "glove", mode="hogwild", shuffle="device", seed=42, dtype=dtype)
    opt = geglove.Adagrad(geglove.CooMatrix(V, I, J, X, xmax), cfg, cfg.costFunction())
    info = opt.info()
    assert V * info["row_stride"] > 2 ** 31 and V * D < 2 ** 31                                          # the point of the test
    # 1. the Java draw order on rows across the whole table, the last one included
    rows = sorted(set([0, 1, 2_581_110, 2_581_111, 5_162_220, 5_162_221, V // 2, V - 2, V - 1][:9 if V > 5_162_221 else 4] + [V // 2, V - 2, V - 1] + rng.integers(0, V, 12).tolist()))
    want = _java_rows(42, D, rows, bf16=dtype == "bf16")
    foc = opt.get_state("focus").reshape(V, D); ctx = opt.get_state("context").reshape(V, D)
    fb, cb = opt.get_state("fbias"), opt.get_state("cbias")
    for r in rows:
        assert np.array_equal(foc[r].view(np.uint32), want[r]["focus"].view(np.uint32)), ("focus row", r)
        assert np.array_equal(ctx[r].view(np.uint32), want[r]["context"].view(np.uint32)), ("context row", r)
        assert fb[r] == want[r]["fbias"] and cb[r] == want[r]["cbias"], ("biases of row", r)
    assert np.all(opt.get_state("gsq_focus").reshape(V, D)[rows] == 1.0)
    # 2. epochs over the hub-heavy matrix: finite, falling
    n = len(I)
    costs = [opt.epoch(it) / n for it in range(3)]
    print("V=5M D=%d %s: cost %s, kernel %.2f ms, %d workers" % (D, dtype, np.round(costs, 5).tolist(), opt.last_kernel_ms()[0], info["groups_in_flight"]))
    assert np.all(np.isfinite(costs)) and costs[2] < costs[1] < costs[0]
    # extract = (focus + context) / 2 on the sampled rows, row V - 1 included (fp32 arithmetic, widened: Optimizer.java:129-140)
    foc2 = opt.get_state("focus").reshape(V, D); ctx2 = opt.get_state("context").reshape(V, D)
    ext = opt.extractResult().reshape(V, D)
    for r in rows:
        np.testing.assert_array_equal(ext[r], ((foc2[r] + ctx2[r]) / np.float32(2)).astype(np.float64))
    touched = np.unique(I)
    g = opt.get_state("gsq_fbias")
    assert np.all(g[touched] > 1.0) and np.count_nonzero(g != 1.0) == len(touched)
    del foc, ctx, foc2, ctx2, ext
    opt.close()
    # 3. every nonzero visited exactly once, rows and columns spread over the whole range (conflict-free: each update adds wc^2 > 0
    #    to the bias accumulators of its row and its column, once)
    opt = geglove.Adagrad(geglove.CooMatrix(V, ci, cj, cx, 0.2), cfg, cfg.costFunction())
    opt.epoch(0)
    gf, gc = opt.get_state("gsq_fbias"), opt.get_state("gsq_cbias")
    assert np.array_equal(np.nonzero(gf != 1.0)[0], np.sort(ci)) and np.array_equal(np.nonzero(gc != 1.0)[0], np.sort(cj))
    opt.close()
