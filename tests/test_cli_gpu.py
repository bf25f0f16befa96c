"""End to end: `geglove -c tests/golden/tiny.config.yml` (the reference's CLI shape, J/Main.java:133-160)
writes out/<name>.vectors.tsv + .dict.tsv in the reference's format; the numbers equal the oracle pipeline."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

import oracle as O
from geglove import synth

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(REPO, "graph-embeddings_amd", "bin", "geglove")
GOLD = os.path.join(REPO, "tests", "golden")


def _graph_from_host(config="tiny.config.yml", nt="tiny.nt", similarity=False):
    L = C.CDLL(os.path.join(REPO, "graph-embeddings_amd", "lib", "libgehost.so"))
    fn = L.geh_graph_summary_similarity if similarity else L.geh_graph_summary
    fn.restype = C.c_char_p
    out = fn(os.path.join(REPO, "tests/golden", config).encode(), os.path.join(REPO, "tests/golden", nt).encode()).decode().splitlines()
    V = int(re.match(r"V=(\d+)", out[1]).group(1))
    src, dst, w, keys, types = [], [], [], [], []
    for line in out[2:]:
        v, t, key, o, _ = line.split("\t")
        keys.append(key); types.append(int(t))
        for m in re.finditer(r"(\d+)\(([^)]+)\)", o):
            src.append(int(v)); dst.append(int(m.group(1))); w.append(float(m.group(2)))
    o_csr, i_csr = synth.edges_to_csr(V, np.array(src), np.array(dst), np.array(w, np.float32))
    return dict(V=V, out=o_csr, inn=i_csr), keys, types


def test_cli_end_to_end(gpu, tmp_path):
    assert os.path.exists(EXE), "host CLI not built"
    cwd = tmp_path
    os.makedirs(cwd / "tests" / "golden")
    for f in ("tiny.config.yml", "tiny.nt"):
        (cwd / "tests" / "golden" / f).write_bytes(open(os.path.join(REPO, "tests", "golden", f), "rb").read())
    r = subprocess.run([EXE, "-c", "tests/golden/tiny.config.yml"], cwd=cwd, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr + r.stdout
    assert "Writing files with prefix: tiny_pglove_partial_directed_0.1_0.001_adagrad_pca_8" in r.stdout
    name = "tiny_pglove_partial_directed_0.1_0.001_adagrad_pca_8"
    vec = (cwd / "out" / (name + ".vectors.tsv")).read_text().splitlines()
    dic = (cwd / "out" / (name + ".dict.tsv")).read_text().splitlines()
    # header = writeConfig(); SimilarityGroup.toString() contains a newline, so its second line has no '#'
    # (EmbeddingTextWriter.java:66-68 writes "# " + s.toString() verbatim) -- reproduced as is
    k = dic.index("key\ttype")
    hd, rows = dic[:k], dic[k + 1:]
    hv, body_v = vec[:k], vec[k:]
    assert hv == hd and hv[0] == "# Starting the embedding creation process with following settings:"
    assert "# BCA Alpha: 0.1" in hv and "# pglove Maximum Iterations: 3" in hv
    assert " method:jarowinkler, threshold: 0.95" in hv
    assert rows == ["http://ex.org/authors/a1\tURI", "http://ex.org/authors/a2\tURI", "http://ex.org/authors/a3\tURI"]   # prefix filter
    assert len(body_v) == len(rows)
    assert all(re.fullmatch(r"-?\d\.\d{6}E[+-]\d{2}", x) for l in body_v for x in l.split("\t"))      # %11.6E
    # numbers: oracle pipeline on the same graph, same seed, threads 1, tolerance/maxiter of the config
    g, keys, types = _graph_from_host()
    coo = O.bca_build(g["V"], g["out"], g["inn"], 0.1, 1e-3, True, O.NORM_NONE)
    m = O.Glove(g["V"], 8, coo["I"], coo["J"], coo["X"], coo["max"], O.COST_PGLOVE, seed=42, threads=1)
    m.optimize(3, 1e-4)
    ref = m.extract()
    want = [i for i, k in enumerate(keys) if k.startswith("http://ex.org/authors/")]
    got = np.array([[float(x) for x in l.split("\t")] for l in body_v])
    np.testing.assert_allclose(got, ref[want], rtol=5e-7, atol=0)
    for l, i in zip(body_v, want):
        assert l.split("\t") == [O.format_11_6E(v) for v in ref[i]]          # byte-identical text


def test_cli_error_paths(gpu, tmp_path):
    r = subprocess.run([EXE, "-c", "missing.yml"], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 1 and "Cannot find configuration file + missing.yml" in r.stderr
    (tmp_path / "bad.yml").write_text("graph: g.nt\nmethod: glove\n")
    r = subprocess.run([EXE, "-c", "bad.yml"], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 1 and "Invalid configuration: No dimension specified" in r.stderr
    r = subprocess.run([EXE, "-c"], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 1 and "No configuration file specified, exiting..." in r.stderr


def test_cli_coo_checkpoint_and_other_optimisers(gpu, tmp_path):
    """device.save_coo / device.load_coo give the same vectors as the direct run; opt.method
    adam is accepted (Main.createOptimizer, J/Main.java:121-130)."""
    cwd = tmp_path
    os.makedirs(cwd / "tests" / "golden")
    (cwd / "tests" / "golden" / "tiny.nt").write_bytes(open(os.path.join(REPO, "tests", "golden", "tiny.nt"), "rb").read())
    base = open(os.path.join(REPO, "tests", "golden", "tiny.config.yml")).read()
    (cwd / "a.yml").write_text(base.replace("  seed: 42", "  seed: 42\n  save_coo: tiny.gecoo") + "\n")
    (cwd / "b.yml").write_text(base.replace("  seed: 42", "  seed: 42\n  load_coo: tiny.gecoo").replace("graph: tests/golden/tiny.nt", "graph: missing.nt") + "\n")
    name = "tiny_pglove_partial_directed_0.1_0.001_adagrad_pca_8"
    r = subprocess.run([EXE, "-c", "a.yml"], cwd=cwd, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr + r.stdout
    first = (cwd / "out" / (name + ".vectors.tsv")).read_text()
    os.remove(cwd / "out" / (name + ".vectors.tsv"))
    r = subprocess.run([EXE, "-c", "b.yml"], cwd=cwd, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "loaded COO checkpoint tiny.gecoo" in r.stdout, r.stderr + r.stdout
    second = (cwd / "out" / "missing_pglove_partial_directed_0.1_0.001_adagrad_pca_8.vectors.tsv").read_text()
    assert first.splitlines()[2:] == second.splitlines()[2:]            # same numbers (the header names the graph file)
    (cwd / "c.yml").write_text(base.replace("method: adagrad", "method: adam"))
    r = subprocess.run([EXE, "-c", "c.yml"], cwd=cwd, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and os.path.exists(cwd / "out" / "tiny_pglove_partial_directed_0.1_0.001_adam_pca_8.vectors.tsv"), r.stderr


def test_host_ingest_adds_the_similarity_edges(gpu):
    """C++ host ingest with `similarity:` groups (Rdf2GrphConverter.convert :41-56, :100-110, :127-186) against the
    oracle's compare loop run on the vertex labels of the same graph."""
    L = C.CDLL(os.path.join(REPO, "graph-embeddings_amd", "lib", "libgehost.so"))
    for f in ("geh_graph_summary", "geh_graph_summary_similarity", "geh_java_float"):
        getattr(L, f).restype = C.c_char_p
    L.geh_java_float.argtypes = [C.c_float]
    cfg, nt = os.path.join(GOLD, "similar.config.yml").encode(), os.path.join(GOLD, "similar.nt").encode()

    def parse(text):
        lines = text.decode().splitlines()
        assert lines[0] == "OK", text
        out = {}
        for l in lines[2:]:
            v, typ, key, outs, _ = l.split("\t")
            out[int(v)] = (key, dict((int(e.split("(")[0]), e.split("(")[1][:-1]) for e in outs.split()[1:]))
        return lines[1], out

    head0, plain = parse(L.geh_graph_summary(cfg, nt))
    head1, full = parse(L.geh_graph_summary_similarity(cfg, nt))
    labels = {v: k for v, (k, _) in plain.items()}
    ids = {k: v for v, k in labels.items()}
    assert sum(k == "Renée de Vries" for k in labels.values()) == 1                # "Ren\\u00E9e" is decoded and merges with the UTF-8 spelling
    objects = {}                                                                    # predicate -> object vertices, from the fixture itself
    literal_vertex = {}                                                             # (predicate, label) -> vertex: literals merge per predicate
    for line in open(nt.decode(), encoding="utf-8"):
        if line.startswith("#") or not line.strip():
            continue
        s, p, o = line.rstrip(" .\n").split(" ", 2)
        p = p.strip("<>")
        if o.startswith('"'):
            lex = re.sub(r"\\u([0-9A-Fa-f]{4})", lambda m: chr(int(m.group(1), 16)), o[1:o.rindex('"')])
            suffix = o[o.rindex('"') + 1:]
            key = lex + ("^^" + suffix[3:-1] if suffix.startswith("^^") else suffix)
            cands = [v for v, k in labels.items() if k == key]
        else:
            cands = [v for v, k in labels.items() if k == o.strip("<>")]
        # literals merge per predicate: several vertices may carry the label; the right one has an in-edge from s
        sv = [v for v, k in labels.items() if k == s.strip("<>")][0]
        if o.startswith('"'):                                                       # ids are handed out in file order
            if (p, key) not in literal_vertex:
                literal_vertex[(p, key)] = min(v for v in cands if v not in literal_vertex.values())
            ov = literal_vertex[(p, key)]
        else:
            ov = cands[0]
        assert ov in plain[sv][1]
        objects.setdefault(p, set()).add(ov)
    groups = [("http://xmlns.com/foaf/0.1/name", "http://xmlns.com/foaf/0.1/name", O.sim_cfg("jarowinkler", 0.9)),
              ("http://ex.org/altName", "http://purl.org/dc/elements/1.1/creator", O.sim_cfg("levenshtein", 0.1)),
              ("http://purl.org/dc/elements/1.1/title", "http://purl.org/dc/elements/1.1/title", O.sim_cfg("token_jaccard", 0.4)),
              ("http://ex.org/year", "http://ex.org/year", O.sim_cfg("numeric", 0.5, smooth=0.5))]
    expected = {v: dict(e) for v, (_, e) in plain.items()}
    pairs = 0
    for sp, tp, c in groups:
        src, tgt = sorted(objects[sp]), sorted(objects[tp])
        verts = sorted(set(src) | set(tgt))
        pos = {v: k for k, v in enumerate(verts)}
        i, j, sim = O.compare_group(c, [labels[v] for v in verts], [pos[v] for v in src], [pos[v] for v in tgt], src, tgt, upper_triangle=sp == tp)
        pairs += len(i)
        for a, b, w in zip(i, j, sim):
            va, vb = src[a], tgt[b]
            expected[va].setdefault(vb, L.geh_java_float(float(w)).decode())       # an existing edge keeps its weight (first edge wins)
            expected[vb].setdefault(va, L.geh_java_float(float(w)).decode())
    assert pairs >= 6
    assert head1 == head0 + " pairs=%d" % pairs
    assert {v: e for v, (_, e) in full.items()} == expected
    # spot checks a reader can follow: the two spellings of the same name are linked, equal names on two predicates are not merged
    a, b = ids["Jan Jansen"], ids["Jan Janssen"]
    assert b in full[a][1] and a in full[b][1] and full[a][1][b] == full[b][1][a]


def test_cli_end_to_end_with_similarity_edges(gpu, tmp_path):
    """similar.config.yml through the CLI: ingest, four similarity groups on the device, builder, trainer (deterministic mode),
    writer.  The vectors equal the oracle pipeline run on the graph the ingest test above checks edge by edge."""
    cwd = tmp_path
    os.makedirs(cwd / "tests" / "golden")
    for f in ("similar.config.yml", "similar.nt"):
        (cwd / "tests" / "golden" / f).write_bytes(open(os.path.join(GOLD, f), "rb").read())
    r = subprocess.run([EXE, "-c", "tests/golden/similar.config.yml"], cwd=cwd, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr + r.stdout
    assert "Processing similarities for predicate http://xmlns.com/foaf/0.1/name" in r.stdout
    made = [int(m) for m in re.findall(r"Created links for (\d+) literal pairs", r.stdout)]
    assert len(made) == 4 and sum(made) >= 6
    name = "similar_glove_partial_directed_0.1_0.001_adagrad_8"
    vec = (cwd / "out" / (name + ".vectors.tsv")).read_text().splitlines()
    dic = (cwd / "out" / (name + ".dict.tsv")).read_text().splitlines()
    k = dic.index("key\ttype")
    rows, body_v = dic[k + 1:], vec[k:]
    assert [r_.split("\t")[0] for r_ in rows] == ["http://ex.org/a/%d" % a for a in range(1, 10)]
    g, keys, types = _graph_from_host("similar.config.yml", "similar.nt", similarity=True)
    coo = O.bca_build(g["V"], g["out"], g["inn"], 0.1, 1e-3, True, O.NORM_NONE)
    m = O.Glove(g["V"], 8, coo["I"], coo["J"], coo["X"], coo["max"], O.COST_GLOVE, seed=42, threads=1)
    m.optimize(3, 1e-4)
    ref = m.extract()
    want = [i for i, key in enumerate(keys) if key.startswith("http://ex.org/a/")]
    for l, i in zip(body_v, want):
        assert l.split("\t") == [O.format_11_6E(v) for v in ref[i]]          # byte-identical text
    # the similarity edges matter: without them the numbers differ
    g0, _, _ = _graph_from_host("similar.config.yml", "similar.nt", similarity=False)
    coo0 = O.bca_build(g0["V"], g0["out"], g0["inn"], 0.1, 1e-3, True, O.NORM_NONE)
    assert len(coo0["I"]) < len(coo["I"])


def _write_synthetic_nt(path, n_authors=300, n_papers=450, n_venues=8):
    """The DBLP-like generator as N-Triples with one weighted predicate (vertex ids by first appearance, like the converter)."""
    g = synth.dblp_like_graph(n_authors, n_papers, n_venues)
    ptr, idx, _ = g["out"]
    with open(path, "w") as f:
        for v in range(g["V"]):
            for k in range(ptr[v], ptr[v + 1]):
                f.write("<http://ex.org/n/%d> <http://ex.org/p> <http://ex.org/n/%d> .\n" % (v, idx[k]))


_MULTI_YML = """graph: g.nt
method: pglove
dim: 24
threads: 1
weights:
  http://ex.org/p: 1
bca:
  alpha: 1e-1
  epsilon: 1e-3
  directed: true
opt:
  method: adagrad
  tolerance: 0
  maxiter: 6
output:
  uri: [ http://ex.org/n/ ]
device:
  mode: hogwild
  shuffle: device
  seed: 42
  save_coo: %s
%s"""


def test_cli_two_ranks_in_one_process(gpu, tmp_path):
    """`device: {gpus: 2}`: the C++ host runs two ranks (threads) through ge_sync -- on this one-GPU box both ranks use device
    0 and meet in host memory (ge_local_group); with one device per rank the same code runs over RCCL.  The sharded builder
    gives the same COO byte for byte; the sharded trainer follows the one-GPU run (cost per epoch, vectors)."""
    _write_synthetic_nt(tmp_path / "g.nt")
    runs = {}
    for name, extra in (("one", ""), ("sync", "  gpus: 2\n  exchange: sync\n  wire: f32\n"), ("overlap", "  gpus: 2\n  exchange: overlap\n")):
        (tmp_path / (name + ".yml")).write_text(_MULTI_YML % (name + ".gecoo", extra))
        r = subprocess.run([EXE, "-c", name + ".yml"], cwd=tmp_path, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr + r.stdout
        costs = [float(m.group(1)) for m in re.finditer(r"epoch \d+  cost ([0-9.eE+-]+)", r.stdout)]
        vec = [l for l in (tmp_path / "out" / "g_pglove_exact_directed_0.1_0.001_adagrad_24.vectors.tsv").read_text().splitlines() if not l.startswith("#")]
        runs[name] = (costs, np.array([[float(x) for x in l.split("\t")] for l in vec]))
    assert (tmp_path / "one.gecoo").read_bytes() == (tmp_path / "sync.gecoo").read_bytes()          # the builder's shards concatenate to the same matrix
    one, E1 = runs["one"]
    assert len(one) == 6
    for name, tol in (("sync", 0.20), ("overlap", 0.30)):             # a small matrix: the first epochs react most to who sees what when
        costs, E = runs[name]
        print("cli gpus 2 %s: cost / one GPU %s" % (name, np.round(np.array(costs) / np.array(one), 3).tolist()))
        assert len(costs) == 6 and costs[-1] < costs[0]
        np.testing.assert_allclose(costs[:2], one[:2], rtol=tol)
        np.testing.assert_allclose(costs[2:], one[2:], rtol=0.06 if name == "sync" else 0.12)    # plumbing test on 750 vertices; the
        assert E.shape == E1.shape and np.all(np.isfinite(E))                                     # statistics are test_parallel_gpu.py's
        n1 = E1 / np.linalg.norm(E1, axis=1, keepdims=True); n2 = E / np.linalg.norm(E, axis=1, keepdims=True)
        iu = np.triu_indices(len(E), 1)
        rho = np.corrcoef((n1 @ n1.T)[iu], (n2 @ n2.T)[iu])[0, 1]
        print("cli gpus 2 %s: cost / one GPU %s, pairwise-cosine correlation %.4f" % (name, np.round(np.array(costs) / np.array(one), 3).tolist(), rho))
        assert rho > 0.9


def test_cli_exits_nonzero_when_one_rank_fails(gpu, tmp_path):
    """with `gpus: 2` a rank that fails (here: told to, before its ge_glove_create) must end the run with a non-zero
    status like Main.java:150-153 does -- not leave the other rank waiting at the host barrier or inside the group's exchange."""
    _write_synthetic_nt(tmp_path / "g.nt")
    (tmp_path / "two.yml").write_text(_MULTI_YML % ("two.gecoo", "  gpus: 2\n  exchange: sync\n  wire: f32\n"))
    env = dict(os.environ, GE_HOST_FAIL_RANK="1")
    r = subprocess.run([EXE, "-c", "two.yml"], cwd=tmp_path, capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode != 0
    assert "GE_HOST_FAIL_RANK" in (r.stderr + r.stdout)
