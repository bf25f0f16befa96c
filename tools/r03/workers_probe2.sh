#!/bin/bash
O=gpurun_out/r03/workers2; mkdir -p $O
echo "== C2 (V=100k, 10 M nonzeros, dim 100): lag at epoch 32 and epoch time against the worker count"
for W in 0 3840 3072 2560 2048 1536; do
  python3 tools/r03/convergence.py device --epochs 32 --ref profiles/r03_convergence_oracle.npz --device-cfg "workers=$W" --out $O/c2_$W.json > $O/c2_$W.txt 2>&1
  python3 -c "
import json;d=json.load(open('$O/c2_$W.json'));r=d['device_over_oracle'];print('workers %5d: kernel_ms %.3f  ratio @10 %.3f @20 %.3f @32 %.3f' % (d['workers'], d['kernel_ms_median'], r[9], r[19], r[31]))"
done
echo "== bench size, dim 100"
for rep in 1 2; do for W in 5120 3072 2560; do
  python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --dim 100 --workers $W > $O/d100_$W.$rep.json 2>/dev/null
  python3 -c "
import json;d=json.loads(open('$O/d100_$W.$rep.json').read().strip().splitlines()[-1]);print('dim 100 workers $W: kernel_ms %.2f (probe %.2f / %.2f)' % (d['roofline']['kernel_ms'], d['trainer']['placement_best_ms'], d['trainer']['placement_worst_ms']))"
done; done
