#!/bin/bash
# Epoch time against the number of resident workers (wavefronts) at the bench size, alternating, one process each.
O=gpurun_out/r03/workers; mkdir -p $O
for rep in 1 2; do for W in 5120 4096 3072 2560 2048; do
  python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --workers $W > $O/bench_$W.$rep.json 2>/dev/null
  python3 -c "
import json;d=json.loads(open('$O/bench_$W.$rep.json').read().strip().splitlines()[-1]);print('workers $W: kernel_ms %.2f  cost after 10 epochs %.5f  placements %d (%.2f / %.2f)' % (d['roofline']['kernel_ms'], d['mean_cost_first_last'][1], d['trainer']['placements'], d['trainer']['placement_best_ms'], d['trainer']['placement_worst_ms']))"
done; done
