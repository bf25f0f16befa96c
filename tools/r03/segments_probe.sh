#!/bin/bash
# Four ranks on one GPU (gloo), the bench's matrix (625 k vertices, 103 M nonzeros, the C4 recipe: the busiest column holds 7 % of the
# nonzeros): cost per epoch over 12 epochs against the number of hub-row exchanges per epoch, both exchange forms.
O=gpurun_out/r03/segments; mkdir -p $O
for form in sync overlap; do for S in 8 16 24 32 48 64; do
  GE_BENCH_BACKEND=gloo GE_BENCH_ONE_DEVICE=1 python3 bench.py --gpus 4 --rows-per-gpu 156250 --nnz-per-gpu 31250000 --steps 9 --warmup 3 --no-cpu-baseline --no-other-form --exchange $form --hub-segments $S 2>/dev/null | tail -1 > $O/${form}_$S.json
  python3 -c "
import json;d=json.load(open('$O/${form}_$S.json'));print('$form S=$S', [round(x,4) for x in d['mean_cost_per_step']], 'ms/step %.0f kernel_ms %.1f' % (d['ms_per_step'], d['roofline']['kernel_ms']))"
done; done
