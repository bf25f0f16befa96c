"""Literal-similarity throughput: ge_similarity_pairs on the GPU next to the oracle's CompareJob
loop on one host core, on synthetic person names / titles.   python tests/tools/similarity_bench.py [n] [--cpu-sample rows]

Prints one JSON line per metric: comparisons/s on the device (whole call: upload, profiles, kernel, sort, download),
comparisons/s of the CPU restatement on a sample of source rows, and that the two agree on the sample."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "graph-embeddings_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import geglove                      # noqa: E402

FIRST = ["jan", "piet", "klaas", "anna", "maria", "johannes", "cornelis", "willem", "hendrik", "pieter", "elisabeth", "catharina",
         "geertruida", "adriana", "gerrit", "dirk", "jacob", "margaretha", "wilhelmina", "petrus"]
LAST = ["jansen", "de vries", "van den berg", "bakker", "visser", "smit", "meijer", "de boer", "mulder", "de groot", "bos", "vos",
        "peters", "hendriks", "van leeuwen", "dekker", "brouwer", "de wit", "dijkstra", "smits"]
WORDS = ["graph", "embedding", "learning", "deep", "network", "knowledge", "bookmark", "coloring", "glove", "vectors", "large", "scale",
         "rdf", "data", "semantic", "web", "entity", "resolution", "linked", "archives", "the", "of", "for", "and", "a", "on", "in"]


def person_names(rng, n):
    out = []
    letters = list("abcdefghijklmnopqrstuvwxyz")
    for _ in range(n):
        s = list(rng.choice(FIRST) + " " + rng.choice(LAST) + (" " + rng.choice(LAST) if rng.random() < 0.2 else ""))
        for _ in range(rng.integers(0, 3)):                   # spelling variants
            k = rng.integers(0, len(s))
            if rng.random() < 0.5: s[k] = rng.choice(letters)
            else: s.insert(k, rng.choice(letters))
        out.append("".join(s))
    return out


def titles(rng, n):
    return [" ".join(rng.choice(WORDS, size=rng.integers(3, 12))) for _ in range(n)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("n", nargs="?", type=int, default=100_000)
    ap.add_argument("--cpu-sample", type=int, default=8, help="source rows the CPU restatement is timed on")
    ap.add_argument("--no-cpu", action="store_true")
    a = ap.parse_args()
    rng = np.random.default_rng(0)
    cases = [("jarowinkler", dict(threshold=0.95), person_names(rng, a.n)),
             ("levenshtein", dict(threshold=0.9), person_names(rng, a.n)),
             ("token_jaccard", dict(threshold=0.8), titles(rng, a.n)),
             ("ngram_jaccard", dict(threshold=0.8, ngram=4), titles(rng, a.n))]
    for method, kw, labels in cases:
        labels = sorted(set(labels))                           # literals merge per predicate: distinct labels only
        n = len(labels)
        grp = geglove.CompareGroup(geglove.SimilarityGroup(dict(method=method, predicate="p", **kw)))
        grp.source = list(range(n)); grp.target = list(range(n))
        lab = dict(enumerate(labels))
        t0 = time.perf_counter()
        va, vb, sim = grp.compare(lab)
        dt = time.perf_counter() - t0
        comparisons = n * (n - 1) // 2
        line = {"method": method, "labels": n, "mean_units": float(np.mean([len(x) for x in labels])), "threshold": kw["threshold"],
                "comparisons": comparisons, "pairs_kept": int(len(sim)), "gpu_seconds": dt, "gpu_comparisons_per_s": comparisons / dt}
        if not a.no_cpu:
            import oracle as O
            rows = list(range(min(a.cpu_sample, n)))
            c = O.sim_cfg(method, **kw)
            t0 = time.perf_counter()
            oi, oj, osim = O.compare_group(c, labels, rows, list(range(n)), upper_triangle=False)
            cdt = time.perf_counter() - t0
            keep = oj > oi                                     # the upper triangle of those rows
            i, j, _ = grp.pairs
            sel = i < len(rows)
            same = (np.array_equal(i[sel], oi[keep]) and np.array_equal(j[sel], oj[keep]) and
                    np.array_equal(sim[sel].view(np.uint32), osim[keep].view(np.uint32)))
            line.update({"cpu_rows": len(rows), "cpu_seconds": cdt, "cpu_comparisons_per_s": len(rows) * (n - 1) / cdt, "cpu_cores": 1,
                         "sample_identical": bool(same)})
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
