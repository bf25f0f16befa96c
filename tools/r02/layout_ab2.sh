#!/bin/bash
set -o pipefail
O=gpurun_out/r02/ab2; mkdir -p $O
for i in 1 2 3 4 5 6; do
  for L in default separate_tables; do
    A=""; [ $L != default ] && A="--layout $L"
    python bench.py --no-cpu-baseline --steps 5 $A > $O/${L}_$i.json 2> $O/${L}_$i.err || { tail -5 $O/${L}_$i.err; exit 1; }
  done
done
python - <<'P'
import json, glob
for f in sorted(glob.glob('gpurun_out/r02/ab2/*.json')):
    d = json.loads(open(f).read().strip().splitlines()[-1]); r = d['roofline']; t = d['trainer']
    print(f.split('/')[-1], 'kernel_ms %.2f' % r['kernel_ms'], 'frac %.3f' % r['frac'])
P
