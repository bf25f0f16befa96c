#!/bin/bash
# Parity of the trainer after a kernel change, then the epoch time at D = 200 / 100 and with bf16 rows.
mkdir -p gpurun_out/r02
OUT=gpurun_out/r02/pipeline_probe.log
: > $OUT
timeout -k 10 900 python -m pytest tests/test_glove_parity_gpu.py tests/test_configs_gpu.py -x -q > gpurun_out/r02/pipeline_tests.log 2>&1
echo "tests rc=$?" >> $OUT; tail -5 gpurun_out/r02/pipeline_tests.log >> $OUT
grep -q "tests rc=0" $OUT || { cat $OUT; exit 1; }
for ARGS in "--dim 200" "--dim 100" "--dim 200 --dtype bf16" "--dim 300 --dtype bf16" "--dim 300"; do
  echo "== $ARGS" >> $OUT
  timeout -k 10 300 python3 bench.py $ARGS --steps 10 --warmup 2 --no-cpu-baseline >> $OUT 2>&1 || { cat $OUT; exit 1; }
done
grep -o '"ms_per_step": [0-9.]*\|== .*\|tests rc.*\|[0-9]* passed.*' $OUT
