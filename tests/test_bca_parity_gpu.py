"""GPU parity of the BCA co-occurrence builder (ge_bca_build) against the CPU restatement.

Bar: BIT-EXACT.  Row order (java.util.HashMap iteration order), indices I/J, nnz, per-row counts,
fp32 values X and the fp64 matrix max must all be identical.
The oracle itself is "parity unpinned" w.r.t. Java except for the hand-derived KATs
(tests/test_oracle_kat.py), which are repeated here against the device.
"""
import numpy as np
import pytest

import geglove
from geglove import capi, synth
import oracle as O
from helpers import make_config

pytestmark = pytest.mark.gpu

NORMS = {"none": O.NORM_NONE, "unity": O.NORM_UNITY, "counts": O.NORM_COUNTS}


def _cfg(alpha=0.1, epsilon=1e-3, directed=True, normalize="none", device=None):
    c = make_config(8, **(device or {}))
    c.bca = {"alpha": alpha, "epsilon": epsilon, "directed": directed, "normalize": normalize}
    return c


def _check(graph, alpha=0.1, epsilon=1e-3, directed=True, normalize="none", device=None):
    dev = geglove.BookmarkColoring(graph, _cfg(alpha, epsilon, directed, normalize, device))
    ref = O.bca_build(graph["V"], graph["out"], graph["inn"], alpha, epsilon, directed, NORMS[normalize])
    assert dev.coOccurrenceCount() == ref["nnz"]
    np.testing.assert_array_equal(dev.row_ptr, ref["row_ptr"])
    np.testing.assert_array_equal(dev.I, ref["I"])
    np.testing.assert_array_equal(dev.J, ref["J"])          # HashMap iteration order
    assert np.array_equal(dev.X.view(np.uint32), ref["X"].view(np.uint32)), "paint values differ"
    assert dev.max() == ref["max"] or (np.isnan(dev.max()) and np.isnan(ref["max"]))
    return dev


def _graph(V, edges):
    src = [e[0] for e in edges]; dst = [e[1] for e in edges]; w = [e[2] if len(e) > 2 else 1.0 for e in edges]
    out, inn = synth.edges_to_csr(V, np.array(src, np.int64), np.array(dst, np.int64), np.array(w, np.float32))
    return dict(V=V, out=out, inn=inn)


def test_kat_isolated_vertex(gpu):
    """KAT-BCA-1: directed => X_bb = (float)alpha + (float)alpha; undirected => (float)alpha."""
    g = _graph(3, [(0, 1)])
    d = _check(g, directed=True)
    assert d.X[d.I == 2].tolist() == [np.float32(0.1) + np.float32(0.1)]
    u = _check(g, directed=False)
    assert u.X[u.I == 2].tolist() == [np.float32(0.1)]


def test_kat_two_vertices(gpu):
    """KAT-BCA-2: 0->1, alpha=0.1, eps=1e-3, directed, normalize none."""
    d = _check(_graph(2, [(0, 1)]))
    assert d.I.tolist() == [0, 0, 1, 1] and d.J.tolist() == [0, 1, 0, 1]
    assert d.X.tolist() == [np.float32(0.2), np.float32(0.09), np.float32(0.09), np.float32(0.2)]
    assert d.max() == float(np.float32(0.2))


def test_kat_epsilon_pruning_drops_paint(gpu):
    """KAT-BCA-3: a 1->many star with 0.9/deg < eps yields only the root entry in the forward pass."""
    deg = 1000
    g = _graph(deg + 1, [(0, k + 1) for k in range(deg)])
    d = _check(g)
    assert d.J[d.I == 0].tolist() == [0]


@pytest.mark.parametrize("normalize", ["none", "unity", "counts"])
@pytest.mark.parametrize("directed", [True, False])
def test_random_graph_bit_exact(gpu, directed, normalize):
    g = synth.synthetic_graph(600, avg_degree=3.0, seed=5, weights=(1.0, 0.1, 0.5))
    _check(g, directed=directed, normalize=normalize)


@pytest.mark.parametrize("alpha,epsilon", [(0.1, 1e-3), (0.5, 1e-2), (0.15, 1e-4), (0.9, 0.05)])
def test_parameters_bit_exact(gpu, alpha, epsilon):
    g = synth.synthetic_graph(400, avg_degree=4.0, seed=9, weights=(1.0, 2.0))
    _check(g, alpha=alpha, epsilon=epsilon)


def test_dblp_like_bit_exact_and_hub_rows(gpu):
    """C3 stand-in graph; hub authors/venues give rows with many HashMap resizes and merge() head insertions."""
    g = synth.dblp_like_graph(1500, 2500, 10)
    d = _check(g)
    assert np.max(np.diff(d.row_ptr)) > 48           # at least one row went through several resizes


def test_stride16_neighbourhood_resizes_early(gpu):
    """java.util.HashMap.treeifyBin below 64 buckets: vertex 0 with neighbours 16,32,..,128 puts nine keys into bin 0 of
    16; the 9th calls treeifyBin, which only resizes -> row 0 is [0,32,64,96,128,16,48,80,112] (hand-derived KAT,
    tests/test_oracle_kat.py), not the load-factor order 0,16,..,128."""
    g = _graph(129, [(0, 16 * k) for k in range(1, 9)])
    d = _check(g, directed=False)
    assert d.J[d.I == 0].tolist() == [0, 32, 64, 96, 128, 16, 48, 80, 112]
    for normalize in ("none", "unity", "counts"):
        _check(g, directed=True, normalize=normalize)            # the reverse pass merges the same keys head-first
        _check(g, directed=False, normalize=normalize)


def test_merge_path_treeifies_with_the_eighth_key(gpu):
    """HashMap.merge counts every node of the bin: in-neighbours 16,..,128 of vertex 0 reach the forward BCV {0} through
    BCV.merge and the 8th of them in bin 0 already triggers the resize (putVal needs nine)."""
    g = _graph(200, [(16 * k, 0) for k in range(1, 9)] + [(0, 1)])
    _check(g, directed=True)
    _check(g, directed=True, normalize="unity")


@pytest.mark.parametrize("normalize", ["none", "unity", "counts"])
@pytest.mark.parametrize("directed", [True, False])
def test_tree_bins_are_replayed_exactly(gpu, directed, normalize):
    """Rows whose keys share a bin at table length >= 64: the bin becomes a red-black tree (root first, later keys behind
    their tree parent, split / untreeify on resize, removeTreeNode for the normalised rows' remove(rootNode)).
    Strides of 64, 256 and 1024 around several hubs, plus ordinary neighbours to push the table through resizes."""
    rng = np.random.default_rng(3)
    V = 70000
    edges = []
    for hub, stride, n in [(0, 64, 40), (5, 256, 25), (9, 1024, 30), (64, 64, 14), (70, 4096, 12)]:
        edges += [(hub, (hub + stride * k) % V) for k in range(1, n)]
        edges += [((hub + stride * k + 7 * stride) % V, hub) for k in range(1, n // 2)]
        edges += [(hub, int(x)) for x in rng.integers(1, 3000, 30)]
    edges = list({(a, b) for a, b in edges if a != b})
    g = _graph(V, [(a, b, 1.0) for a, b in edges])
    _check(g, alpha=0.2, epsilon=1e-4, directed=directed, normalize=normalize)


def test_regular_id_graph_bit_exact(gpu):
    """A lattice-like graph whose neighbour ids are regular multiples (what sequentially numbered entity types give):
    many rows meet the early resize, some the tree bins; every row must still equal the oracle's."""
    V = 8192
    edges = []
    for v in range(0, V, 3):
        edges += [(v, (v + 16 * k) % V) for k in range(1, 11)] + [(v, (v + 1) % V)]
    g = _graph(V, [(a, b, 1.0) for a, b in set(edges) if a != b])
    _check(g, alpha=0.1, epsilon=1e-3, directed=True)
    _check(g, alpha=0.1, epsilon=1e-3, directed=False, normalize="unity")


def test_self_loops_zero_weights_and_sinks(gpu):
    g = _graph(6, [(0, 0, 1.0), (0, 1, 1.0), (1, 2, 0.0), (2, 3, 1.0), (3, 2, 1.0), (4, 4, 2.0)])
    _check(g, directed=True)
    g2 = _graph(6, [(0, 0, 1.0), (0, 1, 1.0), (1, 2, 0.5), (2, 3, 1.0), (3, 2, 1.0), (4, 4, 2.0)])
    _check(g2, directed=False)


def test_row_range_shards_concatenate(gpu):
    """Multi-GPU sharding of the builder: bookmark blocks are independent."""
    g = synth.synthetic_graph(300, avg_degree=3.0, seed=21)
    full = _check(g)
    a = geglove.BookmarkColoring(g, _cfg(), row_range=(0, 130))
    b = geglove.BookmarkColoring(g, _cfg(), row_range=(130, 300))
    np.testing.assert_array_equal(np.concatenate([a.I, b.I]), full.I)
    np.testing.assert_array_equal(np.concatenate([a.J, b.J]), full.J)
    np.testing.assert_array_equal(np.concatenate([a.X, b.X]).view(np.uint32), full.X.view(np.uint32))
    assert max(a.max(), b.max()) == full.max()


def test_table_growth_path(gpu):
    """Force the smallest work table: the builder must grow it and still be exact."""
    g = synth.dblp_like_graph(300, 500, 4)
    _check(g, device={"bca_table_slots": 64})


def test_row_pool_overflow_path(gpu):
    """A row pool that is far too small: the rows that did not fit are re-run alone into a second, exact pool."""
    g = synth.dblp_like_graph(300, 500, 4)
    _check(g, device={"bca_pool_entries": 1000})
    _check(g, normalize="unity", device={"bca_pool_entries": 1000})


def test_bad_arguments(gpu):
    g = _graph(3, [(0, 1)])
    with pytest.raises(geglove.GeError) as e:
        geglove.BookmarkColoring(g, _cfg(alpha=0.0))
    assert e.value.status == capi.GE_ERR_ARG
    bad = dict(V=3, out=(np.array([0, 1, 1, 1]), np.array([7]), np.array([1.0])), inn=g["inn"])
    with pytest.raises(geglove.GeError):
        geglove.BookmarkColoring(bad, _cfg())


def test_pipeline_builder_to_trainer(gpu):
    """BookmarkColoring (device) -> Adagrad (device, deterministic) equals the oracle pipeline bit for bit."""
    g = synth.dblp_like_graph(200, 300, 5)
    cfg = make_config(32, "pglove", mode="deterministic", shuffle="java", seed=3)
    bca = geglove.BookmarkColoring(g, cfg)
    opt = geglove.Adagrad(bca, cfg, cfg.costFunction())
    ref = O.bca_build(g["V"], g["out"], g["inn"], 0.1, 1e-3, True, O.NORM_NONE)
    ora = O.Glove(g["V"], 32, ref["I"], ref["J"], ref["X"], ref["max"], O.COST_PGLOVE, seed=3, threads=1)
    for it in range(2):
        assert opt.epoch(it) / bca.coOccurrenceCount() == ora.epoch()
    np.testing.assert_array_equal(opt.extractResult(), ora.extract().reshape(-1))
