#!/bin/bash
# Line-aligned records and rows (default) against `--layout packed_records` (rows dim + 4 wide, records back to back), same library,
# fresh process per run, alternating.  ROUNDS / CASES as in kernel_ab.sh.
CASES=${CASES:-"--dim 200|--dim 100|--dim 300"}; ROUNDS=${ROUNDS:-3}
mkdir -p gpurun_out/r02
OUT=gpurun_out/r02/records_ab.log
: > $OUT
IFS='|' read -ra CASE_LIST <<< "$CASES"
for ROUND in $(seq 1 $ROUNDS); do
  for L in packed_records default; do
    A=""; [ $L != default ] && A="--layout $L"
    for C in "${CASE_LIST[@]}"; do
      echo "== round $ROUND $L $C" >> $OUT
      timeout -k 10 300 python3 bench.py $C $A --steps 10 --warmup 2 --no-cpu-baseline >> $OUT 2>&1 || { tail -5 $OUT; exit 1; }
    done
  done
done
grep -o '"kernel_ms": [0-9.]*\|== .*' $OUT | paste - -
