#!/bin/bash
# default layout vs round-1 fixed cuts vs separate_tablesd records, alternating fresh processes (compare only in alternation)
set -o pipefail
O=gpurun_out/r02/ab; mkdir -p $O
for i in 1 2 3; do
  for L in default fixed_cuts separate_tables; do
    A=""; [ $L != default ] && A="--layout $L"
    python bench.py --no-cpu-baseline --steps 5 $A > $O/${L}_$i.json 2> $O/${L}_$i.err || { tail -5 $O/${L}_$i.err; exit 1; }
  done
done
python - <<'P'
import json, glob
for f in sorted(glob.glob('gpurun_out/r02/ab/*.json')):
    d = json.loads(open(f).read().strip().splitlines()[-1]); r = d['roofline']; t = d['trainer']
    print(f.split('/')[-1], 'kernel_ms %.2f' % r['kernel_ms'], 'create_s %.2f' % d['create_seconds'], 'frac %.3f' % r['frac'], 'chunks', t['chunks'], 'runs', t['runs'], 'long', t['long_rows'], 'cost', [round(c, 5) for c in d['mean_cost_first_last']])
P
