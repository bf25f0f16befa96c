#!/bin/bash
# The AdaGrad-aware merge of the hub rows (csrc/sync.hip merge_scale; the default -- GE_SYNC_MERGE=sum is the plain sum, tools/r03/world_probe.sh): the bench's matrix over 4 and 6 ranks on one GPU
# (gloo), 8 and 16 exchanges per epoch, synchronous large exchange.
O=gpurun_out/r03/merge; mkdir -p $O
for W in 4 6; do for S in 8 16; do for M in default; do
  R=$((625000 / W)); NZ=$((125000000 / W))
  GE_SYNC_MERGE=$M GE_BENCH_BACKEND=gloo GE_BENCH_ONE_DEVICE=1 python3 bench.py --gpus $W --rows-per-gpu $R --nnz-per-gpu $NZ --steps 6 --warmup 3 --no-cpu-baseline --no-other-form --exchange sync --hub-segments $S 2>/dev/null | tail -1 > $O/w${W}_s${S}_$M.json
  python3 -c "
import json;d=json.load(open('$O/w${W}_s${S}_$M.json'));print('ranks $W exchanges $S merge $M', [round(x,4) for x in d['mean_cost_per_step']])"
done; done; done
