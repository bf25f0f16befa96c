#!/usr/bin/env python3
"""Times the device co-occurrence builder next to the CPU oracle on a synthetic DBLP-like graph and checks
that the two outputs are identical.  Diagnostic (numbers quoted in DESIGN.md); not the headline bench."""
import argparse, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(REPO, "graph-embeddings_amd"), os.path.join(REPO, "oracle")]
import numpy as np
import geglove
from geglove import synth
import oracle as O

ap = argparse.ArgumentParser()
ap.add_argument("--authors", type=int, default=60000)
ap.add_argument("--papers", type=int, default=90000)
ap.add_argument("--venues", type=int, default=50)
ap.add_argument("--no-oracle", action="store_true")
a = ap.parse_args()
g = synth.dblp_like_graph(a.authors, a.papers, a.venues)
print("graph: V=%d, out-pairs=%d" % (g["V"], len(g["out"][1])), flush=True)
cfg = geglove.Configuration({"graph": "s", "method": "pglove", "dim": 8, "bca": {"alpha": 0.1, "epsilon": 1e-3, "directed": True},
                             "opt": {"maxiter": 1}, "output": {"uri": []}})
geglove.BookmarkColoring(synth.dblp_like_graph(100, 100, 3), cfg)      # warm up (context, code objects)
t = time.perf_counter(); dev = geglove.BookmarkColoring(g, cfg); t_dev = time.perf_counter() - t
print("device: nnz=%d  max=%.9g  %.3f s  (%.0f bookmarks/s, %.2f M entries/s)" % (
    dev.coOccurrenceCount(), dev.max(), t_dev, g["V"] / t_dev, dev.coOccurrenceCount() / t_dev / 1e6), flush=True)
if not a.no_oracle:
    t = time.perf_counter(); ref = O.bca_build(g["V"], g["out"], g["inn"], 0.1, 1e-3, True, O.NORM_NONE); t_ref = time.perf_counter() - t
    same = (np.array_equal(dev.I, ref["I"]) and np.array_equal(dev.J, ref["J"])
            and np.array_equal(dev.X.view(np.uint32), ref["X"].view(np.uint32)) and dev.max() == ref["max"])
    print("oracle (1 thread): %.3f s  -> device/oracle speed %.1fx, identical=%s" % (t_ref, t_ref / t_dev, same), flush=True)
    sizes = np.diff(ref["row_ptr"])
    print("row sizes: mean %.1f  median %d  max %d" % (sizes.mean(), np.median(sizes), sizes.max()))
