#!/bin/bash
# one box: the bench kernel time (default layout and separate tables), the box's copy rate, per-XCD streaming rate
set -o pipefail
O=gpurun_out/r02/box_$1; mkdir -p $O
python3 bench.py --no-cpu-baseline --steps 5 > $O/default.json 2> $O/default.err || exit 1
python3 bench.py --no-cpu-baseline --steps 5 --layout separate_tables > $O/separate.json 2> $O/separate.err || exit 1
GE_PROBE_XCD=1 python3 tools/r02/mode_probe.py xcd 3 > $O/xcd.json 2> $O/xcd.err || exit 1
python3 - $O <<'P'
import json, sys
O = sys.argv[1]
for n in ("default", "separate"):
    d = json.loads(open(O + "/%s.json" % n).read().strip().splitlines()[-1]); r = d["roofline"]
    print(n, "kernel_ms %.2f" % r["kernel_ms"], "box_copy_GBps %.0f" % (r["box_copy_GBps"] or 0), "frac %.3f" % r["frac"])
x = json.loads(open(O + "/xcd.json").read().strip().splitlines()[-1])["runs"][0]
print("probe epoch_ms", x["kernel_ms"], "xcd_GBps", x["xcd_GBps"], "total_TBps", x["xcd_total_TBps"])
P
