#!/bin/bash
# Do the two sides' record tables behave differently when they are ONE allocation (GE_PROBE_ARENA=1 in a probe build) instead of two?
# The question is the process-to-process spread on boxes whose processes differ: 8 fresh processes each, alternating, D = 200.
mkdir -p gpurun_out/r02
OUT=gpurun_out/r02/arena_probe.log
: > $OUT
LIB=graph-embeddings_amd/lib/libgeglove.so
cp $LIB /tmp/libgeglove_keep.so
cp tools/r02/_ab/libgeglove_arena.so $LIB
for ROUND in 1 2 3 4 5 6 7 8; do
  for A in 0 1; do
    echo "== round $ROUND arena $A" >> $OUT
    if [ $A = 1 ]; then export GE_PROBE_ARENA=1; else unset GE_PROBE_ARENA; fi
    timeout -k 10 300 python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline >> $OUT 2>&1 || { cp /tmp/libgeglove_keep.so $LIB; tail -5 $OUT; exit 1; }
  done
done
cp /tmp/libgeglove_keep.so $LIB
grep -o '"kernel_ms": [0-9.]*\|== .*' $OUT | paste - -
