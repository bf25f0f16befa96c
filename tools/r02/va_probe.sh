#!/bin/bash
# Which of its two modes a process lands in against where the driver put its record tables: 14 fresh processes, D = 200.
mkdir -p gpurun_out/r02
OUT=gpurun_out/r02/va_probe.log
: > $OUT
for R in $(seq 1 14); do
  timeout -k 10 300 python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
b = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print(round(b['roofline']['kernel_ms'], 2), b['table_ptrs'])" >> $OUT || exit 1
done
cat $OUT
