#!/bin/bash
# Row width rounded up to whole 64-byte lines (GE_PROBE_RW16=1: D = 200 -> 208 floats, every store covers whole lines and the row and
# its accumulator row share no line) against D + 4 (the row ends mid-line).  Fresh process per run, alternating.
# (GE_PROBE_RW16 existed in a probe build only; the result became the default row width, `--layout packed_records` is the old one:
# tools/r02/records_ab.sh repeats the comparison with the shipped library.)
mkdir -p gpurun_out/r02
OUT=gpurun_out/r02/rw_probe.log
: > $OUT
for ROUND in 1 2 3; do
  for R in 0 1; do
    for C in "--dim 200" "--dim 100" "--dim 200 --dtype bf16"; do
      echo "== round $ROUND rw16 $R $C" >> $OUT
      GE_PROBE_RW16=$R timeout -k 10 300 python3 bench.py $C --steps 10 --warmup 2 --no-cpu-baseline >> $OUT 2>&1 || { tail -5 $OUT; exit 1; }
    done
  done
done
grep -o '"kernel_ms": [0-9.]*\|== .*' $OUT | paste - -
