#!/bin/bash
# Placement selection (default: each side's record table allocated up to six times, the candidate with the fastest probe kept) against
# `--layout first_placement`, fresh process per run, alternating, D = 200 bench size.
mkdir -p gpurun_out/r02
OUT=gpurun_out/r02/placement_ab.log
: > $OUT
for ROUND in 1 2 3 4 5; do
  for L in first_placement default; do
    A=""; [ $L != default ] && A="--layout $L"
    timeout -k 10 300 python3 bench.py $A --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
b = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); t = b['trainer']
print('round $ROUND $L', round(b['roofline']['kernel_ms'], 2), 'ms; create', round(b['create_seconds'], 3), 's; placements', t['placements'], 'probe kept / worst', round(t['placement_best_ms'], 3), round(t['placement_worst_ms'], 3))" >> $OUT || exit 1
  done
done
cat $OUT
