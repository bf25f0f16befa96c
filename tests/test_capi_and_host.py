"""CPU-only checks of the boundary: the C-ABI library loads, exports every symbol of include/geglove.h,
validates arguments before touching a device, and fails loudly without a GPU; plus the C++ host logic
(YAML subset, bean check, file name, banner, Java number formats, N-Triples ingest)."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

import geglove
from geglove import capi
import oracle as O

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(REPO, "tests", "golden")


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(REPO, "include", "geglove.h")).read()
    declared = set(re.findall(r"\b(ge_[a-z0-9_]+)\s*\(", header))
    declared -= {"ge_status"}
    lib = capi.lib()
    for name in sorted(declared):
        assert hasattr(lib, name), name
    assert declared == set(capi.SYMBOLS)
    out = subprocess.check_output(["nm", "-D", "--defined-only", capi.LIB_PATH]).decode()
    exported = set(re.findall(r" T (ge_[a-z0-9_]+)", out))
    assert declared <= exported


def test_struct_layout_matches_header():
    """ctypes mirrors must have the C sizes (the library is the authority: compile a probe)."""
    src = '#include <stdio.h>\n#include "geglove.h"\nint main(){printf("%zu %zu %zu %zu %zu %zu", sizeof(ge_glove_cfg), sizeof(ge_glove_info), sizeof(ge_csr), sizeof(ge_bca_cfg), sizeof(ge_sync_cfg), sizeof(ge_transport));return 0;}'
    exe = os.path.join(REPO, "tests", ".probe_sizes")
    subprocess.run(["gcc", "-x", "c", "-", "-I", os.path.join(REPO, "include"), "-o", exe], input=src.encode(), check=True)
    try:
        sizes = [int(x) for x in subprocess.check_output([exe]).split()]
    finally:
        os.remove(exe)
    assert sizes == [C.sizeof(capi.GloveCfg), C.sizeof(capi.GloveInfo), C.sizeof(capi.Csr), C.sizeof(capi.BcaCfg), C.sizeof(capi.SyncCfg), C.sizeof(capi.Transport)]


def test_jni_glue_compiles_against_a_declared_types_stub():
    """graph-embeddings_amd/jni/geglove_jni.c is the binding a maintainer adds to the Java host.  No JDK here,
    so it is compiled -- syntax and types only -- against tests/jni_stub/jni.h, which declares the JNI calls it uses with the
    specification's signatures: a call with the wrong argument list, a ge_* prototype that moved, a cfg field that no longer
    exists all fail here.  Every Native method INTEGRATION.md declares has its Java_..._Native_<name> definition."""
    src = os.path.join(REPO, "graph-embeddings_amd", "jni", "geglove_jni.c")
    subprocess.run(["gcc", "-fsyntax-only", "-std=c11", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(REPO, "tests", "jni_stub"),
                    "-I", os.path.join(REPO, "include"), src], check=True)
    glue = set(re.findall(r"Java_org_uu_nl_embedding_hip_Native_(\w+)\(", open(src).read()))
    doc = open(os.path.join(REPO, "INTEGRATION.md")).read()
    native = doc[doc.index("final class Native {"):]
    native = native[:native.index("\n}\n")]
    declared = set(re.findall(r"static native [\w\[\]]+\s+(\w+)\(", native))
    assert declared and declared == glue, (sorted(declared - glue), sorted(glue - declared))


def test_argument_errors_come_before_any_device_work():
    cfg = capi.GloveCfg(); capi.lib().ge_glove_cfg_default(C.byref(cfg))
    assert (cfg.learning_rate, cfg.threads, cfg.mode, cfg.shuffle) == (np.float32(0.05), 1, capi.GE_MODE_HOGWILD, capi.GE_SHUFFLE_DEVICE)
    h = C.c_void_p()
    assert capi.lib().ge_glove_create(None, None, None, None, C.byref(h)) == capi.GE_ERR_ARG
    cfg.vocab_size, cfg.dim, cfg.nnz = 10, 0, 0
    assert capi.lib().ge_glove_create(C.byref(cfg), None, None, None, C.byref(h)) == capi.GE_ERR_ARG
    assert b"No dimension specified" in capi.lib().ge_last_error()
    cfg.dim = 2 ** 30
    assert capi.lib().ge_glove_create(C.byref(cfg), None, None, None, C.byref(h)) == capi.GE_ERR_ARG
    assert capi.lib().ge_bca_build(None, None, None, C.byref(h)) == capi.GE_ERR_ARG
    assert capi.lib().ge_glove_epoch(None, 0, None) == capi.GE_ERR_ARG


@pytest.mark.skipif(capi.lib().ge_device_count() > 0, reason="only meaningful on a box without a GPU")
def test_no_cpu_fallback_without_a_device():
    m = geglove.CooMatrix(3, [0], [1], [0.1], 0.2)
    cfg = geglove.Configuration({"graph": "g", "method": "glove", "dim": 4, "bca": {"alpha": .1, "epsilon": 1e-3},
                                 "opt": {"maxiter": 1}, "output": {"uri": []}})
    with pytest.raises(geglove.GeError) as e:
        geglove.Adagrad(m, cfg, cfg.costFunction())
    assert e.value.status == capi.GE_ERR_HIP and "no CPU fallback" in str(e.value)


# ---------------------------------------------------------------- C++ host (libgehost.so)
@pytest.fixture(scope="module")
def host():
    capi._share_hip_runtime_with_torch()            # libgehost.so pulls in libgeglove.so: same rule as capi.lib()
    L = C.CDLL(os.environ.get("GE_HOST_LIB") or os.path.join(REPO, "graph-embeddings_amd", "lib", "libgehost.so"))
    for f in ("geh_format_11_6E", "geh_java_double", "geh_java_float", "geh_config_summary", "geh_graph_summary"):
        getattr(L, f).restype = C.c_char_p
    L.geh_format_11_6E.argtypes = [C.c_double]; L.geh_java_double.argtypes = [C.c_double]; L.geh_java_float.argtypes = [C.c_float]
    return L


def test_java_number_formats(host):
    assert host.geh_java_double(0.1) == b"0.1" and host.geh_java_double(0.001) == b"0.001"
    assert host.geh_java_double(1e-4) == b"1.0E-4" and host.geh_java_double(12345678.0) == b"1.2345678E7"
    assert host.geh_java_double(100.0) == b"100.0" and host.geh_java_float(0.1) == b"0.1" and host.geh_java_float(1.0) == b"1.0"
    rng = np.random.default_rng(0)
    wide = rng.standard_normal(3000) * 10.0 ** rng.integers(-8, 8, 3000)
    floats = (rng.standard_normal(3000) * 10.0 ** rng.integers(-6, 3, 3000)).astype(np.float32).astype(np.float64)   # what the writer gets: widened floats
    edge = [0.12345675, 9.9999999e-5, 0.0, -0.0, -2.5, 9.9999995, 9.9999994e7, 1e100, -3.3e-120, 5e-324, 1.7976931348623157e308, 1e-5, 123456.75]
    for v in np.concatenate([wide, floats, edge]):
        assert host.geh_format_11_6E(float(v)).decode() == O.format_11_6E(float(v)), v   # two implementations (std::to_chars / print-and-parse), one spec
    assert host.geh_format_11_6E(0.12345675) == b"1.234568E-01"       # HALF_UP on the shortest digits 12345675 (a binary double slightly below ...75)
    assert host.geh_format_11_6E(1e100) == b"1.000000E+100" and host.geh_format_11_6E(-2.5) == b"-2.500000E+00" and host.geh_format_11_6E(0.0) == b"0.000000E+00"


def test_yaml_subset_bean_and_file_name(host):
    out = host.geh_config_summary(os.path.join(GOLD, "tiny.config.yml").encode(), 1).decode().splitlines()
    assert out[0] == "OK"
    assert "# BCA Alpha: 0.1" in out and "# BCA Epsilon: 0.001" in out and "# pglove Tolerance: 1.0E-4" in out
    assert "# http://purl.org/dc/terms/references: 0.5" in out and "# PCA Minimum Variance: 0.95" in out
    assert "name=tiny_pglove_partial_directed_0.1_0.001_adagrad_pca_8" in out
    assert "ignored=bca.reverse" in out and "ignored=bca.predicates" in out         # legacy keys are tolerated
    assert "uri=1:1" in out
    assert "# http://purl.org/dc/elements/1.1/title -> http://purl.org/dc/elements/1.1/title" in out
    assert " method:ngram_jaccard, threshold: 0.5, ngram: 4" in out


def test_configuration_check_messages(host, tmp_path):
    base = {"graph": "graph: g.nt\n", "method": "method: glove\n", "dim": "dim: 4\n",
            "bca": "bca:\n  alpha: 0.1\n  epsilon: 0.001\n", "output": "output:\n  uri: []\n"}
    msgs = {"dim": "No dimension specified", "graph": "No input graph specified",
            "method": "Invalid method, choose one of: glove, pglove",
            "bca": "Invalid BCA parameters, alpha and epsilon are mandatory",
            "output": "Invalid output parameters, specify at least one group"}
    for missing, msg in msgs.items():
        p = tmp_path / (missing + ".yml")
        p.write_text("".join(v for k, v in base.items() if k != missing))
        out = host.geh_config_summary(str(p).encode(), 1).decode()
        assert out == "ERR\nInvalid configuration: " + msg
    good = tmp_path / "ok.yml"; good.write_text("".join(base.values()))
    assert host.geh_config_summary(str(good).encode(), 1).decode().startswith("OK")
    # same checks in the Python mirror
    with pytest.raises(geglove.InvalidConfigurationException, match="No dimension specified"):
        geglove.Configuration.check(geglove.Configuration({"graph": "g"}))


@pytest.mark.skipif(not os.path.exists("/root/reference/dblp.config.yml"), reason="reference tree not present on this box")
def test_shipped_reference_yamls_load(host):
    names = {"dblp": "dblp-2015-2017_pglove_partial_directed_0.1_0.001_adagrad_pca_300",
             "onstage": "onstage_pglove_partial_directed_0.1_0.001_adagrad_pca_300",
             "saa": "saa_pglove_partial_directed_0.1_0.001_adagrad_pca_300"}
    for n, fname in names.items():
        out = host.geh_config_summary(("/root/reference/%s.config.yml" % n).encode(), 1).decode().splitlines()
        assert out[0] == "OK" and ("name=" + fname) in out
        py = geglove.Configuration.load("/root/reference/%s.config.yml" % n)
        geglove.Configuration.check(py)
        assert py.getDim() == 300 and py.getMethod() == "pglove" and py.getAlpha() == 0.1 and py.isDirected()


def test_ntriples_ingest_follows_the_converter_rules(host):
    out = host.geh_graph_summary(os.path.join(GOLD, "tiny.config.yml").encode(), os.path.join(GOLD, "tiny.nt").encode()).decode().splitlines()
    assert out[0] == "OK" and out[1] == "V=12 triples=14 skipped=1"        # the unweighted predicate is dropped
    rows = [l.split("\t") for l in out[2:]]
    keys = [r[2] for r in rows]
    assert keys.count("Ada Lovelace") == 1                                  # literals merge per predicate
    assert rows[1][1] == "2" and rows[9][1] == "1" and rows[9][2] == "b1"   # LITERAL / BLANK types
    assert rows[6][3] == "out: 2(1.0) 4(1.0) 5(0.5) 8(1.0)"                 # parallel p2->p1 edges collapse, ascending ids
    assert rows[1][4] == "in: 0 4"
    assert keys[8] == "On machines@en" and keys[11] == "1843^^http://www.w3.org/2001/XMLSchema#gYear"


def test_legacy_similarity_dialect_is_skipped_and_unknown_methods_fail(host, tmp_path):
    """The shipped YAMLs use `predicate:` + methods `token` / `jaccard`, which this revision's SimilarityMethod enum
    (Configuration.java:27-29) does not have: such entries are skipped; in the current dialect valueOf's failure is kept."""
    host.geh_graph_summary_similarity.restype = C.c_char_p
    base = ("graph: g.nt\nmethod: glove\ndim: 4\nbca:\n  alpha: 0.1\n  epsilon: 0.001\noutput:\n  uri: []\n"
            "weights:\n  http://xmlns.com/foaf/0.1/name: 1\n  http://purl.org/dc/elements/1.1/title: 1\n  http://purl.org/dc/elements/1.1/creator: 1\n  http://purl.org/dc/terms/references: 1\n")
    legacy = tmp_path / "legacy.yml"
    legacy.write_text(base + "similarity:\n  - predicate: http://purl.org/dc/elements/1.1/title\n    method: token\n    threshold: 0.5\n"
                             "  - predicate: http://xmlns.com/foaf/0.1/name\n    method: jaccard\n    ngram: 4\n    threshold: 0.75\n")
    nt = os.path.join(GOLD, "tiny.nt").encode()
    out = host.geh_graph_summary_similarity(str(legacy).encode(), nt).decode().splitlines()
    assert out[0] == "OK" and out[1].endswith("pairs=0")                      # no group left, so no device call either
    assert out[2:] == host.geh_graph_summary(str(legacy).encode(), nt).decode().splitlines()[2:]
    modern = tmp_path / "modern.yml"
    modern.write_text(base + "similarity:\n  - sourcePredicate: http://xmlns.com/foaf/0.1/name\n    targetPredicate: http://xmlns.com/foaf/0.1/name\n"
                             "    method: token\n    threshold: 0.5\n")
    out = host.geh_graph_summary_similarity(str(modern).encode(), nt).decode()
    assert out.startswith("ERR") and "No enum constant" in out and "SimilarityMethod.TOKEN" in out
    dates = tmp_path / "dates.yml"
    dates.write_text(base + "similarity:\n  - sourcePredicate: http://xmlns.com/foaf/0.1/name\n    targetPredicate: http://xmlns.com/foaf/0.1/name\n"
                            "    method: date_days\n    pattern: yyyy-MMM-dd\n    threshold: 0.5\n")
    out = host.geh_graph_summary_similarity(str(dates).encode(), nt).decode()
    assert out.startswith("ERR") and "pattern" in out


def test_edge_list_reader(host, tmp_path):
    """`source<TAB>target[<TAB>weight]`: ids by first appearance, parallel edges collapse (first wins)."""
    g = tmp_path / "g.tsv"
    g.write_text("# a comment\nn1\tn2\nn2\tn3\t0.5\nn1\tn2\t9\nn3\tn1\t2\r\n")
    cfg = tmp_path / "c.yml"
    cfg.write_text("graph: %s\nmethod: glove\ndim: 4\nbca:\n  alpha: 0.1\n  epsilon: 0.001\noutput:\n  uri: []\n" % g)
    out = host.geh_graph_summary(str(cfg).encode(), str(g).encode()).decode().splitlines()
    assert out[0] == "OK" and out[1] == "V=3 triples=4 skipped=0"
    rows = [l.split("\t") for l in out[2:]]
    assert [r[2] for r in rows] == ["n1", "n2", "n3"] and all(r[1] == "0" for r in rows)
    assert rows[0][3] == "out: 1(1.0)" and rows[1][3] == "out: 2(0.5)" and rows[2][3] == "out: 0(2.0)"
    assert rows[0][4] == "in: 2" and rows[1][4] == "in: 0"
    bad = tmp_path / "bad.tsv"; bad.write_text("n1\tn2\tx\n")
    assert host.geh_graph_summary(str(cfg).encode(), str(bad).encode()).decode().startswith("ERR")
    other = tmp_path / "g.ttl"; other.write_text("")
    err = host.geh_graph_summary(str(cfg).encode(), str(other).encode()).decode()
    assert err.startswith("ERR") and "N-Triples" in err and ".tsv" in err


def test_measurement_scripts_compile():
    """tools/ and tests/tools/ hold the scripts behind the numbers in DESIGN.md; they need a GPU to run but must at least parse."""
    import glob
    import py_compile
    scripts = (glob.glob(os.path.join(REPO, "tools", "*.py")) + glob.glob(os.path.join(REPO, "tools", "r02", "*.py")) +
               glob.glob(os.path.join(REPO, "tools", "r03", "*.py")) + glob.glob(os.path.join(REPO, "tests", "tools", "*.py")))
    assert len(scripts) >= 10
    for path in scripts:
        py_compile.compile(path, doraise=True)
    # the scripts that load the CPU oracle live under tests/ (oracle/ is test infrastructure)
    for path in (glob.glob(os.path.join(REPO, "tools", "*.py")) + glob.glob(os.path.join(REPO, "tools", "r02", "*.py")) +
                 glob.glob(os.path.join(REPO, "tools", "r03", "*.py"))):
        text = open(path).read()
        assert "import oracle" not in text and "from helpers import" not in text, path          # (tests/helpers.py loads the oracle)
