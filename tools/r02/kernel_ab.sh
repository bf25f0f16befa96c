#!/bin/bash
# A/B of two builds of libgeglove.so (tools/r02/_ab/libgeglove_{head,new}.so, built by hand from two revisions of glove.hip)
# in alternation on one box: fresh process per run, three rounds, D = 200 fp32 (the bench line) and D = 100.
mkdir -p gpurun_out/r02
OUT=gpurun_out/r02/kernel_ab.log
: > $OUT
LIB=graph-embeddings_amd/lib/libgeglove.so
cp $LIB /tmp/libgeglove_keep.so
for ROUND in 1 2 3; do
  for WHICH in head new; do
    cp tools/r02/_ab/libgeglove_$WHICH.so $LIB
    for D in 200 100; do
      echo "== round $ROUND $WHICH dim $D" >> $OUT
      timeout -k 10 300 python3 bench.py --dim $D --steps 10 --warmup 2 --no-cpu-baseline >> $OUT 2>&1 || { cp /tmp/libgeglove_keep.so $LIB; tail -5 $OUT; exit 1; }
    done
  done
done
cp /tmp/libgeglove_keep.so $LIB
grep -o '"kernel_ms": [0-9.]*\|== .*' $OUT | paste - -
