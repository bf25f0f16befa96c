#!/bin/bash
set -o pipefail
O=gpurun_out/r02/p3; mkdir -p $O
for i in 1 2 3 4 5 6 7 8 9 10; do
  GE_PROBE_CHASE=1 python3 tools/r02/mode_probe.py c_$i 3 > $O/c_$i.json 2> $O/c_$i.err || { tail -5 $O/c_$i.err; exit 1; }
done
python3 - <<'P'
import glob, json
for f in sorted(glob.glob('gpurun_out/r02/p3/c_*.json'), key=lambda x: int(x.split('_')[-1].split('.')[0])):
    d = json.loads(open(f).read().strip().splitlines()[-1]); r = d['runs'][0]
    print(f.split('/')[-1], r['kernel_ms'][-1], ' '.join('%s=%s' % (k.replace('chase_',''), v) for k, v in r.items() if k.startswith('chase') or k.startswith('ptr')))
P
