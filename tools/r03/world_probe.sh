#!/bin/bash
# Does the number of hub-row exchanges a matrix needs go with the updates ALL ranks put on its busiest column (what the library's rule
# counts) or with one rank's?  The bench's matrix (625 k vertices, 103 M nonzeros) split over 2, 4 and 6 ranks on one GPU (gloo), 8 and 16
# exchanges per epoch, synchronous large exchange, 9 epochs.
O=gpurun_out/r03/world; mkdir -p $O
for W in 2 6; do for S in 8 16; do   # (run with GE_SYNC_MERGE=sum for the plain sum this probe was written for)
  R=$((625000 / W)); NZ=$((125000000 / W))
  GE_SYNC_MERGE=${MERGE:-sum} GE_BENCH_BACKEND=gloo GE_BENCH_ONE_DEVICE=1 python3 bench.py --gpus $W --rows-per-gpu $R --nnz-per-gpu $NZ --steps 6 --warmup 3 --no-cpu-baseline --no-other-form --exchange sync --hub-segments $S 2>/dev/null | tail -1 > $O/w${W}_s$S.json
  python3 -c "
import json;d=json.load(open('$O/w${W}_s$S.json'));print('ranks $W exchanges $S', [round(x,4) for x in d['mean_cost_per_step']], d['config']['vocab'], d['config']['nnz_per_gpu'])"
done; done
