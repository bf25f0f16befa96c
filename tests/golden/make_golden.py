#!/usr/bin/env python3
"""Generates tests/golden/*.npz.

RESTATEMENT-GENERATED, NOT JAVA-GENERATED: the reference cannot run in the build image (Java, no
JDK) and ships no fixtures, so these vectors come from the CPU oracle (oracle/ge_oracle.c) and pin
it against regressions; they do not pin it against Java (the KATs in tests/test_oracle_kat.py are
the only hand-derived anchors).  Re-run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [os.path.join(REPO, "graph-embeddings_amd"), os.path.join(REPO, "oracle")]
import oracle as O            # noqa: E402
from geglove import synth     # noqa: E402


def main():
    g = synth.dblp_like_graph(60, 90, 4, seed=11)
    out = {}
    for name, directed, norm in (("dir_none", True, O.NORM_NONE), ("und_none", False, O.NORM_NONE),
                                 ("dir_unity", True, O.NORM_UNITY), ("dir_counts", True, O.NORM_COUNTS)):
        c = O.bca_build(g["V"], g["out"], g["inn"], 0.1, 1e-3, directed, norm)
        out.update({"bca_%s_I" % name: c["I"], "bca_%s_J" % name: c["J"], "bca_%s_X" % name: c["X"],
                    "bca_%s_max" % name: np.float64(c["max"])})
    c = O.bca_build(g["V"], g["out"], g["inn"], 0.1, 1e-3, True, O.NORM_NONE)
    for method, kind in (("glove", O.COST_GLOVE), ("pglove", O.COST_PGLOVE)):
        for D in (8, 50):
            m = O.Glove(g["V"], D, c["I"], c["J"], c["X"], c["max"], kind, seed=42, threads=1)
            hist = [m.epoch() for _ in range(3)]
            out["train_%s_%d_hist" % (method, D)] = np.array(hist)
            out["train_%s_%d_vec" % (method, D)] = m.extract().astype(np.float64)
            out["train_%s_%d_perm" % (method, D)] = m.perm.copy()
    np.savez_compressed(os.path.join(HERE, "dblp_like_60_90_4.npz"), **out)
    print("wrote", os.path.join(HERE, "dblp_like_60_90_4.npz"), "keys:", len(out))
    # a regular-id lattice: rows whose keys share java.util.HashMap bins (early treeifyBin resize, tree bins from 64 buckets on)
    g = synth.lattice_graph(2048, 16, 10, 3)
    out = {}
    for name, directed, norm in (("dir_none", True, O.NORM_NONE), ("und_unity", False, O.NORM_UNITY), ("dir_counts", True, O.NORM_COUNTS)):
        c = O.bca_build(g["V"], g["out"], g["inn"], 0.1, 1e-4, directed, norm)
        out.update({"bca_%s_I" % name: c["I"], "bca_%s_J" % name: c["J"], "bca_%s_X" % name: c["X"], "bca_%s_max" % name: np.float64(c["max"])})
    np.savez_compressed(os.path.join(HERE, "lattice_2048_16.npz"), **out)
    print("wrote", os.path.join(HERE, "lattice_2048_16.npz"), "nnz:", {k: len(v) for k, v in out.items() if k.endswith("_J")})


if __name__ == "__main__":
    main()


# c2_oracle_costs.npz: the first 16 per-epoch mean costs of the sequential oracle on BASELINE C2 (V = 100 k, 10.0 M nonzeros, dim 100,
# glove, Java order, seed 42), cut from profiles/r03_convergence_oracle.npz, which `python3 tests/tools/convergence_oracle.py --epochs 48
# --out profiles/r03_convergence_oracle.npz` writes (restatement-generated, not Java-generated; 20 s per epoch on one core).
