#!/bin/bash
# Four ranks on one GPU (gloo), the bench's matrix (625 k vertices, 103 M nonzeros): the cost per epoch against how many context rows are
# reconciled inside the epoch (count >= max(MIN, N / DIV)) and how often (--hub-segments).
O=gpurun_out/r03/heavy; mkdir -p $O
run() { # name, extra env..., -- bench args
  name=$1; shift
  env GE_BENCH_BACKEND=gloo GE_BENCH_ONE_DEVICE=1 "$@" 2>/dev/null | tail -1 > $O/$name.json
  python3 -c "
import json;d=json.load(open('$O/$name.json'));print('$name', [round(x,4) for x in d['mean_cost_per_step']], 'ms/step %.0f' % d['ms_per_step'])"
}
B="python3 bench.py --gpus 4 --rows-per-gpu 156250 --nnz-per-gpu 31250000 --steps 4 --warmup 3 --no-cpu-baseline --no-other-form --exchange sync"
run default $B
run seg32 $B --hub-segments 32
run div81920 GE_SYNC_HEAVY_DIV=81920 GE_SYNC_HEAVY_MIN=64 $B
run div327680 GE_SYNC_HEAVY_DIV=327680 GE_SYNC_HEAVY_MIN=64 $B
run div327680_seg32 GE_SYNC_HEAVY_DIV=327680 GE_SYNC_HEAVY_MIN=64 $B --hub-segments 32
