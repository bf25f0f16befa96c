#!/bin/bash
O=gpurun_out/r03/theta; mkdir -p $O
for rep in 1 2; do for T in 0.25 0.1 0.05 0.02; do
  python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --hot-theta $T > $O/bench_$T.$rep.json 2> $O/err.txt || tail -2 $O/err.txt
  python3 -c "
import json;d=json.loads(open('$O/bench_$T.$rep.json').read().strip().splitlines()[-1]);t=d['trainer'];print('theta $T: kernel_ms %.2f  hub columns %d holding %.1f M nonzeros, chunks %d (hub %d), runs %d, schedule GB %.1f, cost %.5f' % (d['roofline']['kernel_ms'], t['hot_columns'], t['hot_nonzeros']/1e6, t['chunks'], t['hub_chunks'], t['runs'], t['schedule_bytes']/1e9, d['mean_cost_first_last'][1]))"
done; done
