#!/bin/bash
# The side numbers DESIGN.md quotes, re-measured with the final library in one call: one rank's share of the 8-GPU problem with ge_sync's
# passes (shard_rehearsal), Adam / AMSGrad and bf16 epochs at the bench size, the BCA-built pipeline (pglove, D = 200).
mkdir -p gpurun_out/r02
O=gpurun_out/r02/refresh.log
: > $O
echo "== shard_rehearsal 8" >> $O; timeout -k 10 400 python3 tools/r02/shard_rehearsal.py 8 >> $O 2>&1 || { tail -5 $O; exit 1; }
for A in "--opt adam" "--opt amsgrad" "--dtype bf16" "--dim 300" "--dim 300 --dtype bf16"; do
  echo "== bench $A" >> $O; timeout -k 10 300 python3 bench.py $A --steps 10 --warmup 2 --no-cpu-baseline >> $O 2>&1 || { tail -5 $O; exit 1; }
done
echo "== pipeline_bench" >> $O; timeout -k 10 400 python3 tools/pipeline_bench.py >> $O 2>&1 || { tail -5 $O; exit 1; }
grep -v "^\s*$" $O | cut -c1-1500 | tail -40
