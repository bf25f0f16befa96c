"""Committed fixtures (tests/golden/dblp_like_60_90_4.npz, made by tests/golden/make_golden.py from the
CPU oracle -- restatement-generated, not Java-generated).  The oracle must keep reproducing them (CPU),
and the device must match them bit for bit (GPU)."""
import os

import numpy as np
import pytest

import geglove
from geglove import synth
import oracle as O
from helpers import make_config

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "dblp_like_60_90_4.npz"))
G = synth.dblp_like_graph(60, 90, 4, seed=11)
BCA_CASES = [("dir_none", True, "none"), ("und_none", False, "none"), ("dir_unity", True, "unity"), ("dir_counts", True, "counts")]
NORM = {"none": O.NORM_NONE, "unity": O.NORM_UNITY, "counts": O.NORM_COUNTS}


@pytest.mark.parametrize("name,directed,norm", BCA_CASES)
def test_oracle_reproduces_golden_bca(name, directed, norm):
    c = O.bca_build(G["V"], G["out"], G["inn"], 0.1, 1e-3, directed, NORM[norm])
    np.testing.assert_array_equal(c["I"], GOLD["bca_%s_I" % name])
    np.testing.assert_array_equal(c["J"], GOLD["bca_%s_J" % name])
    assert np.array_equal(c["X"].view(np.uint32), GOLD["bca_%s_X" % name].view(np.uint32))
    assert c["max"] == float(GOLD["bca_%s_max" % name])


@pytest.mark.parametrize("method", ["glove", "pglove"])
@pytest.mark.parametrize("D", [8, 50])
def test_oracle_reproduces_golden_training(method, D):
    c = O.bca_build(G["V"], G["out"], G["inn"], 0.1, 1e-3, True, O.NORM_NONE)
    m = O.Glove(G["V"], D, c["I"], c["J"], c["X"], c["max"], O.COST_GLOVE if method == "glove" else O.COST_PGLOVE, seed=42, threads=1)
    hist = [m.epoch() for _ in range(3)]
    assert hist == GOLD["train_%s_%d_hist" % (method, D)].tolist()
    np.testing.assert_array_equal(m.extract(), GOLD["train_%s_%d_vec" % (method, D)])


@pytest.mark.gpu
@pytest.mark.parametrize("name,directed,norm", BCA_CASES)
def test_device_matches_golden_bca(gpu, name, directed, norm):
    cfg = make_config(8)
    cfg.bca = {"alpha": 0.1, "epsilon": 1e-3, "directed": directed, "normalize": norm}
    d = geglove.BookmarkColoring(G, cfg)
    np.testing.assert_array_equal(d.I, GOLD["bca_%s_I" % name])
    np.testing.assert_array_equal(d.J, GOLD["bca_%s_J" % name])
    assert np.array_equal(d.X.view(np.uint32), GOLD["bca_%s_X" % name].view(np.uint32))
    assert d.max() == float(GOLD["bca_%s_max" % name])


@pytest.mark.gpu
@pytest.mark.parametrize("method", ["glove", "pglove"])
@pytest.mark.parametrize("D", [8, 50])
def test_device_matches_golden_training(gpu, method, D):
    cfg = make_config(D, method, mode="deterministic", shuffle="java", seed=42)
    bca = geglove.BookmarkColoring(G, cfg)
    opt = geglove.Adagrad(bca, cfg, cfg.costFunction())
    hist = [opt.epoch(it) / bca.coOccurrenceCount() for it in range(3)]
    assert hist == GOLD["train_%s_%d_hist" % (method, D)].tolist()
    np.testing.assert_array_equal(opt.perm(), GOLD["train_%s_%d_perm" % (method, D)])
    got = opt.extractResult().reshape(G["V"], D)
    ref = GOLD["train_%s_%d_vec" % (method, D)]
    assert np.max(np.abs(got - ref)) <= 1e-4          # north-star tolerance
    np.testing.assert_array_equal(got, ref)           # in fact bit-exact


# ---- a regular-id lattice: rows that meet java.util.HashMap's treeifyBin (early resize, tree bins) --------------------------
LAT = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lattice_2048_16.npz"))
GL = synth.lattice_graph(2048, 16, 10, 3)
LAT_CASES = [("dir_none", True, "none"), ("und_unity", False, "unity"), ("dir_counts", True, "counts")]


@pytest.mark.parametrize("name,directed,norm", LAT_CASES)
def test_oracle_reproduces_golden_lattice(name, directed, norm):
    c = O.bca_build(GL["V"], GL["out"], GL["inn"], 0.1, 1e-4, directed, NORM[norm])
    np.testing.assert_array_equal(c["J"], LAT["bca_%s_J" % name])
    assert np.array_equal(c["X"].view(np.uint32), LAT["bca_%s_X" % name].view(np.uint32))
    assert c["max"] == float(LAT["bca_%s_max" % name])


@pytest.mark.gpu
@pytest.mark.parametrize("name,directed,norm", LAT_CASES)
def test_device_matches_golden_lattice(gpu, name, directed, norm):
    cfg = make_config(8)
    cfg.bca = {"alpha": 0.1, "epsilon": 1e-4, "directed": directed, "normalize": norm}
    d = geglove.BookmarkColoring(GL, cfg)
    np.testing.assert_array_equal(d.I, LAT["bca_%s_I" % name])
    np.testing.assert_array_equal(d.J, LAT["bca_%s_J" % name])
    assert np.array_equal(d.X.view(np.uint32), LAT["bca_%s_X" % name].view(np.uint32))
    assert d.max() == float(LAT["bca_%s_max" % name])
