#!/bin/bash
# Round 2, first look at the bimodal epoch time: fresh processes of the same binary, alternating variants.
set -o pipefail
mkdir -p gpurun_out/r02
rocprofv3 -L > gpurun_out/r02/counters.txt 2>&1 || true
for i in 1 2 3 4; do
  python bench.py --no-cpu-baseline --steps 5 > gpurun_out/r02/m_base_$i.json 2> gpurun_out/r02/m_base_$i.err || exit 1
  python bench.py --no-cpu-baseline --steps 5 --hot none > gpurun_out/r02/m_hotnone_$i.json 2> gpurun_out/r02/m_hotnone_$i.err || exit 1
  GE_GLOVE_STALE_BUDGET=16000 python bench.py --no-cpu-baseline --steps 5 > gpurun_out/r02/m_stale16k_$i.json 2> gpurun_out/r02/m_stale16k_$i.err || exit 1
  echo "round $i done"
done
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r02/m_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split('/')[-1], round(d['roofline']['kernel_ms'],2), round(d['ms_per_step'],2), d['mean_cost_first_last'])
P
