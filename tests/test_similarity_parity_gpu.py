"""ge_similarity_pairs (HIP) against the oracle's CompareJob loop: same pairs, same order, same float similarity
(bit-exact: integer/byte work plus the reference's own float / double arithmetic).  SURVEY.md 8f rank 4."""
import ctypes as C

import numpy as np
import pytest

import oracle as O
import geglove
from geglove import capi

pytestmark = pytest.mark.gpu


def mutate(rng, w, alphabet):
    w = list(w)
    for _ in range(rng.integers(0, 3)):
        op = rng.integers(0, 3)
        if op == 0 and w:
            w[rng.integers(0, len(w))] = rng.choice(alphabet)
        elif op == 1 and w:
            del w[rng.integers(0, len(w))]
        else:
            w.insert(rng.integers(0, len(w) + 1), rng.choice(alphabet))
    return "".join(w)


def names(rng, n, max_len=18, alphabet=None):
    alphabet = alphabet or list("abcdefghij klmn") + ["é", "ß", "𝄞", "\t"]
    base = ["".join(rng.choice(alphabet, size=rng.integers(1, max_len))) for _ in range(max(n // 4, 1))]
    out = [mutate(rng, base[rng.integers(0, len(base))], alphabet) for _ in range(n)]
    out[0] = ""                                               # an empty label
    if n > 3:
        out[3] = out[2]                                       # equal labels on two vertices
    return out


def titles(rng, n):
    vocab = ["graph", "embedding", "the", "of", "learning", "deep", "a", "network", "x", "knowledge", "for", "bookmark", "coloring",
             "glove", "vectors", "on", "and", "large", "scale", "rdf", "data\tsets", "in"]
    return [" ".join(rng.choice(vocab, size=rng.integers(0, 9))) + ("  " if rng.random() < 0.2 else "") for _ in range(n)]


def run_both(method, labels, source, target, upper, source_vertex=None, target_vertex=None, **kw):
    sv = list(source) if source_vertex is None else source_vertex
    tv = list(target) if target_vertex is None else target_vertex
    oi, oj, osim = O.compare_group(O.sim_cfg(method, **kw), labels, source, target, sv, tv, upper_triangle=upper)
    d = dict(method=method, sourcePredicate="p", targetPredicate="p" if upper else "q", **kw)
    grp = geglove.CompareGroup(geglove.SimilarityGroup(d))
    # vertex ids: position in `labels` unless given
    verts = {}
    lab = {}
    for k, (p, v) in enumerate(zip(source, sv)):
        grp.addToSource(1000 + v); lab[1000 + v] = labels[p]
    for k, (p, v) in enumerate(zip(target, tv)):
        grp.addToTarget(1000 + v); lab[1000 + v] = labels[p]
    a, b, sim = grp.compare(lab)
    gi, gj, _ = grp.pairs
    return (oi, oj, osim), (gi, gj, sim), (a, b)


def assert_same(ref, got, what=""):
    np.testing.assert_array_equal(got[0], ref[0], err_msg=what + " source positions")
    np.testing.assert_array_equal(got[1], ref[1], err_msg=what + " target positions")
    np.testing.assert_array_equal(got[2].view(np.uint32), ref[2].view(np.uint32), err_msg=what + " similarities (bits)")


@pytest.mark.parametrize("method,kw", [
    ("jarowinkler", dict(threshold=0.8)), ("jarowinkler", dict(threshold=0.0)), ("jarowinkler", dict(threshold=0.95)),
    ("levenshtein", dict(threshold=0.7)), ("levenshtein", dict(threshold=0.0)), ("levenshtein", dict(threshold=1.5)),
    ("ngram_jaccard", dict(threshold=0.4, ngram=2)), ("ngram_jaccard", dict(threshold=0.3, ngram=0)),
    ("ngram_cosine", dict(threshold=0.5, ngram=3)), ("ngram_cosine", dict(threshold=0.0, ngram=4)),
])
@pytest.mark.parametrize("upper", [True, False])
def test_string_metrics_match_the_oracle(gpu, method, kw, upper):
    rng = np.random.default_rng(hash((method, upper)) % 1000)
    labels = names(rng, 700)
    if upper:
        src = tgt = list(rng.permutation(700)[:600])          # ragged: 600 = 2 * 256 + 88
    else:
        src, tgt = list(range(0, 300)), list(range(250, 700))
    ref, got, _ = run_both(method, labels, src, tgt, upper, **kw)
    assert len(ref[0]) > 0 or kw["threshold"] > 1
    assert_same(ref, got, method)


@pytest.mark.parametrize("method,kw", [("token_jaccard", dict(threshold=0.5)), ("token_cosine", dict(threshold=0.6)), ("token_jaccard", dict(threshold=0.0))])
@pytest.mark.parametrize("upper", [True, False])
def test_token_metrics_match_the_oracle(gpu, method, kw, upper):
    rng = np.random.default_rng(7 + upper)
    labels = titles(rng, 500)
    src, tgt = (list(range(500)),) * 2 if upper else (list(range(0, 200)), list(range(200, 500)))
    ref, got, _ = run_both(method, labels, src, tgt, upper, **kw)
    assert len(ref[0]) > 0
    assert_same(ref, got, method)


@pytest.mark.parametrize("max_len", [64, 65, 256, 300, 1024])
def test_long_labels_take_the_wider_kernels(gpu, max_len):
    rng = np.random.default_rng(max_len)
    alphabet = list("abcdef ")
    base = "".join(rng.choice(alphabet, size=max_len))
    labels = [base] + [mutate(rng, base, alphabet)[:max_len] for _ in range(60)] + names(rng, 40)
    for method, thr in (("jarowinkler", 0.7), ("levenshtein", 0.9), ("ngram_cosine", 0.8)):
        ref, got, _ = run_both(method, labels, list(range(len(labels))), list(range(len(labels))), True, threshold=thr)
        assert len(ref[0]) > 50
        assert_same(ref, got, "%s at %d units" % (method, max_len))


def test_numeric_with_dying_jobs(gpu):
    rng = np.random.default_rng(3)
    typ = "^^http://www.w3.org/2001/XMLSchema#integer"
    labels = [str(int(v)) + (typ if rng.random() < 0.7 else "") for v in rng.integers(1500, 2100, size=400)]
    labels += ["", "12", "x12" + typ, "-5" + typ, "+7", "99999999999" + typ, "2147483647", "-2147483648"]
    n = len(labels)
    for upper in (True, False):
        src, tgt = (list(range(n)),) * 2 if upper else (list(range(0, 150)), list(range(150, n)))
        for kw in (dict(threshold=0.3, smooth=0.5), dict(threshold=0.25, smooth=0.5, distance=3.0), dict(threshold=1.0), dict(threshold=0.2, smooth=0.3)):
            ref, got, _ = run_both("numeric", labels, src, tgt, upper, **kw)
            assert_same(ref, got, "numeric %s" % kw)
    # typed short labels ("12" is shorter than the '^' position of the long ones) kill most typed jobs in the square group
    ref, got, _ = run_both("numeric", labels, list(range(n)), list(range(n)), True, threshold=0.3, smooth=0.5)
    alive = set(ref[0])
    assert alive and len(alive) < n


@pytest.mark.parametrize("method", ["date_days", "date_months", "date_years"])
@pytest.mark.parametrize("time", ["bidirectional", "backwards", "forwards"])
def test_dates_match_the_oracle(gpu, method, time):
    rng = np.random.default_rng(11)
    iso = ["%04d%02d%02d" % (y, m, d) for y, m, d in zip(rng.integers(1890, 2030, 300), rng.integers(1, 13, 300), rng.integers(1, 32, 300))]   # some invalid days
    iso += ["", "2020", "20200101Z", "20200101+0200", "20200101^^http://www.w3.org/2001/XMLSchema#date", "2020-01-01"]
    ref, got, _ = run_both(method, iso, list(range(len(iso))), list(range(len(iso))), True, threshold=0.2, smooth=0.6, distance=1.0, time=time)
    assert len(ref[0]) > 0
    assert_same(ref, got, method + " iso")
    dashed = ["%04d-%02d-%02d" % (y, m, d) for y, m, d in zip(rng.integers(1, 2030, 200), rng.integers(1, 14, 200), rng.integers(0, 32, 200))]
    ref, got, _ = run_both(method, dashed, list(range(0, 80)), list(range(80, 200)), False, threshold=0.05, smooth=0.7, pattern="yyyy-MM-dd", time=time)
    assert_same(ref, got, method + " pattern")


def test_vertex_identity_and_result_views(gpu):
    labels = ["anna", "anne", "anna", "hanna"]
    # vertices 5 and 7 carry the same label; vertex 5 sits in both lists and is never compared with itself
    ref, got, (a, b) = run_both("jarowinkler", labels, [0, 1], [0, 2, 3], False, source_vertex=[5, 6], target_vertex=[5, 7, 8], threshold=0.8)
    assert_same(ref, got)
    assert list(zip(a - 1000, b - 1000)) == [(5, 7), (5, 8), (6, 5), (6, 7)]         # never (5, 5); anne~hanna stays below 0.8
    assert got[2][0] == 1.0


def test_arguments_are_checked(gpu):
    L = capi.lib()
    off = np.array([0, 3, 2000], np.int64); units = np.zeros(2000, np.uint16)
    table = capi.Strings(2, off.ctypes.data_as(C.POINTER(C.c_int64)), units.ctypes.data_as(C.POINTER(C.c_uint16)))
    idx = np.array([0, 1], np.int32); p = idx.ctypes.data_as(C.POINTER(C.c_int32))
    cfg = capi.SimCfg(); L.ge_sim_cfg_default(C.byref(cfg))
    assert (cfg.method, cfg.ngram, cfg.smooth, cfg.time) == (4, 3, 1.0, 2)
    h = C.c_void_p()
    assert L.ge_similarity_pairs(C.byref(table), p, p, 2, p, p, 2, C.byref(cfg), C.byref(h)) == capi.GE_ERR_ARG      # 1997 units > 1024
    assert b"1024" in L.ge_last_error()
    cfg.method = 7; cfg.pattern = b"yyyy-MMM-dd"
    off2 = np.array([0, 3, 6], np.int64)
    table2 = capi.Strings(2, off2.ctypes.data_as(C.POINTER(C.c_int64)), units.ctypes.data_as(C.POINTER(C.c_uint16)))
    assert L.ge_similarity_pairs(C.byref(table2), p, p, 2, p, p, 2, C.byref(cfg), C.byref(h)) == capi.GE_ERR_ARG
    assert b"pattern" in L.ge_last_error()
    cfg.method = 42; cfg.pattern = None
    assert L.ge_similarity_pairs(C.byref(table2), p, p, 2, p, p, 2, C.byref(cfg), C.byref(h)) == capi.GE_ERR_ARG
    cfg.method = 4
    bad = np.array([0, 9], np.int32).ctypes.data_as(C.POINTER(C.c_int32))
    assert L.ge_similarity_pairs(C.byref(table2), bad, p, 2, p, p, 2, C.byref(cfg), C.byref(h)) == capi.GE_ERR_ARG
    cfg.upper_triangle = 1
    assert L.ge_similarity_pairs(C.byref(table2), p, p, 2, p, p, 1, C.byref(cfg), C.byref(h)) == capi.GE_ERR_ARG
    cfg.upper_triangle = 0
    assert L.ge_similarity_pairs(C.byref(table2), p, p, 0, p, p, 2, C.byref(cfg), C.byref(h)) == capi.GE_OK             # empty group
    n = C.c_int64(-1)
    assert L.ge_sim_pairs_get(h, C.byref(n), None, None, None) == capi.GE_OK and n.value == 0
    L.ge_sim_pairs_destroy(h)


def test_many_pairs_grow_the_result_buffers(gpu):
    # threshold 0 keeps every pair: 3000 x 3000 / 2 = 4.5 M pairs > the first buffer (4 M)
    rng = np.random.default_rng(5)
    labels = names(rng, 3000, max_len=8)
    grp = geglove.CompareGroup(geglove.SimilarityGroup(dict(method="jarowinkler", threshold=0.0, predicate="p")))
    for v in range(3000):
        grp.addToSource(v); grp.addToTarget(v)
    a, b, sim = grp.compare({v: labels[v] for v in range(3000)})
    assert len(a) == 3000 * 2999 // 2
    i, j, _ = grp.pairs
    assert np.all(np.diff(i.astype(np.int64) * 3000 + j) > 0)           # job order, every pair once
    k = rng.integers(0, len(a), size=2000)
    for q in k:
        v, _ = O.sim_pair(O.sim_cfg("jarowinkler"), labels[a[q]], labels[b[q]])
        assert np.float32(v) == sim[q]


def test_job_ranges_shard_a_group(gpu):
    """The multi-GPU form: each rank runs a block of CompareJobs; the blocks' results concatenated are the group's."""
    from geglove import parallel
    rng = np.random.default_rng(21)
    labels = names(rng, 900)
    lab = dict(enumerate(labels))

    def group():
        g = geglove.CompareGroup(geglove.SimilarityGroup(dict(method="jarowinkler", threshold=0.85, predicate="p")))
        g.source = list(range(900)); g.target = list(range(900))
        return g

    whole = group(); whole.compare(lab)
    parts = []
    for rank in range(3):
        g = group(); g.compare(lab, job_range=parallel.shard_rows(900, 3, rank)); parts.append(g.pairs)
    assert all(len(p[0]) > 0 for p in parts)
    for k in range(3):
        np.testing.assert_array_equal(np.concatenate([p[k] for p in parts]), whole.pairs[k])
    L = capi.lib()
    cfg = capi.SimCfg(); L.ge_sim_cfg_default(C.byref(cfg)); cfg.job_begin, cfg.job_end = 5, 3
    off = np.array([0, 3, 6], np.int64); units = np.zeros(6, np.uint16); idx = np.array([0, 1], np.int32)
    table = capi.Strings(2, off.ctypes.data_as(C.POINTER(C.c_int64)), units.ctypes.data_as(C.POINTER(C.c_uint16)))
    p = idx.ctypes.data_as(C.POINTER(C.c_int32)); h = C.c_void_p()
    assert L.ge_similarity_pairs(C.byref(table), p, p, 2, p, p, 2, C.byref(cfg), C.byref(h)) == capi.GE_ERR_ARG
