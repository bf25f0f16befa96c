#!/bin/bash
# Epoch time against the number of resident workers (wavefronts), D = 200 and D = 100: a kernel at the bandwidth ceiling
# does not care, one that waits for its own loads scales with the workers.
mkdir -p gpurun_out/r02
OUT=gpurun_out/r02/occupancy_probe.log
: > $OUT
for D in 200 100; do
  for W in 2560 3840 5120; do
    echo "== dim $D workers $W" >> $OUT
    python3 bench.py --dim $D --workers $W --steps 10 --warmup 2 --no-cpu-baseline >> $OUT 2>&1 || exit 1
  done
done
grep -o '"ms_per_step": [0-9.]*\|== .*' $OUT
