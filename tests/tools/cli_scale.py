#!/usr/bin/env python3
"""The C++ CLI end to end at the shipped-YAML scale: DBLP-like graph (300 k vertices) as N-Triples -> `geglove -c` (ingest,
builder, trainer, writer), wall time per stage from the CLI's own log lines.    python3 tests/tools/cli_scale.py [authors papers]"""
import os, re, subprocess, sys, tempfile, time
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(REPO, "graph-embeddings_amd"), os.path.join(REPO, "tests"), os.path.join(REPO, "oracle")]
from test_cli_gpu import _write_synthetic_nt, EXE

A, P = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (60000, 90000)
d = tempfile.mkdtemp(prefix="ge_cli_")
t0 = time.time(); _write_synthetic_nt(os.path.join(d, "g.nt"), A, P, 50); t_gen = time.time() - t0
open(os.path.join(d, "c.yml"), "w").write("""graph: g.nt
method: pglove
dim: 200
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
  maxiter: 5
output:
  uri: [ http://ex.org/n/ ]
device:
  seed: 42
""")
t0 = time.time()
r = subprocess.run([EXE, "-c", "c.yml"], cwd=d, capture_output=True, text=True)
wall = time.time() - t0
print("nt file written in %.1f s; CLI wall %.1f s, rc %d" % (t_gen, wall, r.returncode))
stamps = [(m.group(1), m.group(2).strip(), m.group(3)) for m in re.finditer(r"^(\d\d:\d\d:\d\d) INFO  (\S+)\s+:: (.*)$", r.stdout, re.M)]
for ts, who, msg in stamps:
    if not msg.startswith(("Starting", "Graph File", "Embedding", "Using", "Writing", "Number", "BCA Alpha", "BCA Eps", "Gradient", "pglove", "Output", "http", "No ", "Unweighted", "Similarity")):
        print(ts, who, msg[:110])
print(r.stderr[-500:])
out = os.path.join(d, "out")
print({f: os.path.getsize(os.path.join(out, f)) for f in os.listdir(out)} if os.path.isdir(out) else "no output")
