"""Synthetic inputs for the hot path: co-occurrence triples and graphs.

Deterministic: everything derives from SplitMix64 streams, so the same arguments give the
same arrays on every machine (the GPU box regenerates instead of shipping data).
"""
import numpy as np

_GOLD = np.uint64(0x9E3779B97F4A7C15)


def splitmix64(seed, n, offset=0):
    """n outputs of SplitMix64 seeded with `seed`, starting at draw `offset` (vectorised)."""
    with np.errstate(over="ignore"):
        k = np.arange(offset + 1, offset + n + 1, dtype=np.uint64)
        z = np.uint64(seed & 0xFFFFFFFFFFFFFFFF) + k * _GOLD
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def _u01(u):
    return (u >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def synthetic_coo(V, N, seed=0xC0FFEE, zipf_s=1.0, dedupe=True):
    """Hub-heavy co-occurrence triples shaped like a BCA matrix with `normalize: none`.

    Every vertex has its (i,i) entry = 0.2f (the merged forward+reverse root paint at alpha=0.1);
    the other entries have i uniform, j ~ Zipf(s) over a fixed random relabelling, duplicates
    removed; X = 10^U(-3.5,-0.7) clipped to (1e-4, 0.2], so 0 < X < 1 (pGloVe-valid) and max = 0.2.
    Returns (I, J, X, max) with entries grouped by row like BookmarkColoring emits them.
    """
    V = int(V); N = int(N)
    n_extra = max(N - V, 0)
    ui = splitmix64(seed, n_extra, 0)
    uj = splitmix64(seed ^ 0x5A5A5A5A, n_extra, 0)
    ux = splitmix64(seed ^ 0x0F0F0F0F, n_extra, 0)
    i = (_u01(ui) * V).astype(np.int64)
    ranks = np.arange(1, V + 1, dtype=np.float64)
    cdf = np.cumsum(ranks ** (-float(zipf_s)))
    cdf /= cdf[-1]
    r = np.searchsorted(cdf, _u01(uj), side="right")
    r = np.minimum(r, V - 1)
    relabel = np.argsort(splitmix64(seed ^ 0x77777777, V, 0), kind="stable")
    j = relabel[r].astype(np.int64)
    x = np.power(10.0, -3.5 + _u01(ux) * 2.8)
    x = np.clip(x, 1.0001e-4, 0.2).astype(np.float32)
    keep = i != j
    i, j, x = i[keep], j[keep], x[keep]
    if dedupe:
        key = i * V + j
        _, first = np.unique(key, return_index=True)
        first.sort()
        i, j, x = i[first], j[first], x[first]
    di = np.arange(V, dtype=np.int64)
    I = np.concatenate([di, i]); J = np.concatenate([di, j])
    X = np.concatenate([np.full(V, np.float32(0.2), np.float32), x])
    order = np.argsort(I, kind="stable")          # group by row, diagonal entry first
    return (I[order].astype(np.int32), J[order].astype(np.int32), X[order].astype(np.float32),
            float(np.float32(0.2)))


def conflict_free_batch(V, n, seed=1):
    """n <= V nonzeros with all i distinct and all j distinct: any schedule gives one result."""
    assert n <= V
    pi = np.argsort(splitmix64(seed, V), kind="stable")[:n]
    pj = np.argsort(splitmix64(seed ^ 0xABCDEF, V), kind="stable")[:n]
    x = np.power(10.0, -3.5 + _u01(splitmix64(seed ^ 0x1234, n)) * 2.8)
    x = np.clip(x, 1.0001e-4, 0.2).astype(np.float32)
    return pi.astype(np.int32), pj.astype(np.int32), x


def edges_to_csr(V, src, dst, w):
    """Weighted CSR (out-neighbours) + CSC (in-neighbours) with the canonical choices of
    SURVEY.md 8(c): unique neighbours per row, ascending neighbour id, parallel edges collapse
    to the FIRST one in input order (EdgeNeighborhoodAlgorithm.getEdge returns the first match)."""
    src = np.asarray(src, np.int64); dst = np.asarray(dst, np.int64); w = np.asarray(w, np.float32)

    def build(a, b):
        order = np.lexsort((np.arange(a.shape[0]), b, a))       # by (a, b), input order last
        a2, b2, w2 = a[order], b[order], w[order]
        if a2.shape[0]:
            first = np.ones(a2.shape[0], bool)
            first[1:] = (a2[1:] != a2[:-1]) | (b2[1:] != b2[:-1])
            a2, b2, w2 = a2[first], b2[first], w2[first]
        ptr = np.zeros(V + 1, np.int64)
        np.add.at(ptr, a2 + 1, 1)
        ptr = np.cumsum(ptr)
        return ptr.astype(np.int64), b2.astype(np.int32), w2.astype(np.float32)

    return build(src, dst), build(dst, src)


def synthetic_graph(V, avg_degree=4.0, seed=7, weights=(1.0,)):
    """Random directed multigraph with a few hubs; returns dict(V, out, inn)."""
    E = int(V * avg_degree)
    us = _u01(splitmix64(seed, E)); ud = _u01(splitmix64(seed ^ 0x999, E)); uw = splitmix64(seed ^ 0x333, E)
    src = (us * V).astype(np.int64)
    dst = np.minimum((V * ud ** 2.0).astype(np.int64), V - 1)      # skew towards low ids = hubs
    wt = np.asarray(weights, np.float32)[(uw % np.uint64(len(weights))).astype(np.int64)]
    out, inn = edges_to_csr(V, src, dst, wt)
    return dict(V=V, out=out, inn=inn)


def dblp_like_graph(n_authors=2000, n_papers=3000, n_venues=20, seed=11):
    """Stand-in for the DBLP graph the shipped dblp.config.yml points at (data not available):
    paper -creator-> author (1-4 per paper), paper -references-> paper (0-5), paper -venue-> venue,
    author -name-> literal, paper -title-> literal; all weights 1 (dblp.config.yml:5-9)."""
    rng_k = 0
    def u(n):
        nonlocal rng_k
        rng_k += 1
        return _u01(splitmix64(seed + 1000 * rng_k, n))
    A, P, Vn = n_authors, n_papers, n_venues
    a0, p0, v0 = 0, A, A + P
    ln0 = v0 + Vn                  # author-name literals
    lt0 = ln0 + A                  # title literals
    V = lt0 + P
    src, dst = [], []
    n_auth = 1 + (u(P) * 4).astype(np.int64)
    for k in range(4):
        m = n_auth > k
        au = np.minimum((A * u(P) ** 1.5).astype(np.int64), A - 1)
        src.append(p0 + np.nonzero(m)[0]); dst.append(a0 + au[m])
    n_ref = (u(P) * 6).astype(np.int64)
    for k in range(5):
        m = n_ref > k
        ref = np.minimum((P * u(P) ** 2.0).astype(np.int64), P - 1)
        src.append(p0 + np.nonzero(m)[0]); dst.append(p0 + ref[m])
    src.append(p0 + np.arange(P)); dst.append(v0 + np.minimum((Vn * u(P) ** 2).astype(np.int64), Vn - 1))
    src.append(a0 + np.arange(A)); dst.append(ln0 + np.arange(A))
    src.append(p0 + np.arange(P)); dst.append(lt0 + np.arange(P))
    src = np.concatenate(src); dst = np.concatenate(dst)
    keep = src != dst
    src, dst = src[keep], dst[keep]
    out, inn = edges_to_csr(V, src, dst, np.ones(src.shape[0], np.float32))
    types = np.zeros(V, np.int8); types[ln0:] = 2          # NodeInfo: URI=0, LITERAL=2
    keys = (["http://dblp.l3s.de/d2r/resource/authors/a%d" % k for k in range(A)]
            + ["http://dblp.l3s.de/d2r/resource/publications/p%d" % k for k in range(P)]
            + ["http://dblp.l3s.de/d2r/resource/venues/v%d" % k for k in range(Vn)]
            + ["Author Name %d" % k for k in range(A)] + ["Title of paper %d" % k for k in range(P)])
    return dict(V=V, out=out, inn=inn, keys=keys, types=types)


def synthetic_coo_shard(V_total, row_range, n_local, seed=0xC0FFEE):
    """Bench-scale generator for one GPU's row block of a V_total-vertex matrix (BASELINE C4 recipe):
    i uniform over the block, j log-uniform over a fixed relabelling (density ~ 1/rank, i.e. the
    continuous form of Zipf(1.0)), one (i,i) entry = 0.2f per owned row, duplicates removed,
    X = 10^U(-3.5,-0.7) in (1e-4, 0.2].  Output is sorted by (i, j).  The relabelling depends on
    `seed` only, so every rank sees the same hub columns; the draws depend on (seed, row_begin)."""
    rb, re = row_range
    rows = re - rb
    n_extra = max(int(n_local) - rows, 0)
    s = (seed + 0x1000003 * (rb + 1)) & 0xFFFFFFFFFFFFFFFF
    i = rb + np.minimum((_u01(splitmix64(s, n_extra)) * rows).astype(np.int64), rows - 1)
    r = np.power(float(V_total + 1), _u01(splitmix64(s ^ 0x5A5A5A5A, n_extra))).astype(np.int64) - 1
    np.clip(r, 0, V_total - 1, out=r)
    relabel = np.argsort(splitmix64(seed ^ 0x77777777, V_total), kind="stable")
    j = relabel[r]
    del r, relabel
    key = i * np.int64(V_total) + j
    del i, j
    diag = np.arange(rb, re, dtype=np.int64) * np.int64(V_total + 1)
    key = np.unique(np.concatenate([diag, key]))
    I = (key // V_total).astype(np.int32)
    J = (key % V_total).astype(np.int32)
    del key
    X = np.power(10.0, -3.5 + _u01(splitmix64(s ^ 0x0F0F0F0F, I.shape[0])) * 2.8)
    X = np.clip(X, 1.0001e-4, 0.2).astype(np.float32)
    X[I == J] = np.float32(0.2)
    return I, J, X, float(np.float32(0.2))


def lattice_graph(V=2048, stride=16, reach=10, every=3):
    """Vertices numbered so that neighbours are regular multiples apart (what sequentially numbered entity types give): vertex v
    (every `every`-th one) points at v + stride, v + 2 stride, ..., v + reach * stride (mod V) and at v + 1.  The keys of a row then
    pile up in single java.util.HashMap bins: rows meet the early resize of treeifyBin and, from 64 buckets on, tree bins."""
    src, dst = [], []
    for v in range(0, V, every):
        for k in range(1, reach + 1):
            src.append(v); dst.append((v + stride * k) % V)
        src.append(v); dst.append((v + 1) % V)
    src = np.array(src, np.int64); dst = np.array(dst, np.int64)
    keep = src != dst
    pairs = np.unique(np.stack([src[keep], dst[keep]], 1), axis=0)
    out, inn = edges_to_csr(V, pairs[:, 0], pairs[:, 1], np.ones(len(pairs), np.float32))
    return dict(V=V, out=out, inn=inn)
