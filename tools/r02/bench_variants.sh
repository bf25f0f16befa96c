#!/bin/bash
# the bench at the other named shapes (one call, one box): dim 100, dim 300 fp32, dim 300 bf16 rows (C5), adam, pglove
set -o pipefail
O=gpurun_out/r02/variants; mkdir -p $O
python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 || { tail -5 $O/smoke.log; exit 1; }
tail -1 $O/smoke.log
for V in "d200:--dim 200" "d100:--dim 100" "d300:--dim 300" "d300bf16:--dim 300 --dtype bf16" "d200bf16:--dim 200 --dtype bf16" "adam:--opt adam" "pglove:--method pglove"; do
  name=${V%%:*}; args=${V#*:}
  python3 bench.py --no-cpu-baseline --steps 5 $args > $O/$name.json 2> $O/$name.err || { tail -5 $O/$name.err; exit 1; }
done
python3 - <<'P'
import json, glob
for f in sorted(glob.glob('gpurun_out/r02/variants/*.json')):
    d = json.loads(open(f).read().strip().splitlines()[-1]); r = d['roofline']
    print(f.split('/')[-1], 'updates/s %.3e' % d['value'], 'kernel_ms %.2f' % r['kernel_ms'], 'frac %.3f' % r['frac'], 'sched B/upd %.0f' % r['schedule_bytes_per_update'], 'box_copy %.0f' % (r['box_copy_GBps'] or 0), 'cost', [round(c, 5) for c in d['mean_cost_first_last']])
P
