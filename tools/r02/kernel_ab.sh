#!/bin/bash
# A/B of builds of libgeglove.so (tools/r02/_ab/libgeglove_<name>.so, built by hand from revisions / probe variants of glove.hip)
# in alternation on one box: fresh process per run.
#   VARIANTS="head new"  CASES="--dim 200|--dim 100"  ROUNDS=3  bash tools/r02/kernel_ab.sh
VARIANTS=${VARIANTS:-"head new"}; CASES=${CASES:-"--dim 200|--dim 100"}; ROUNDS=${ROUNDS:-3}
mkdir -p gpurun_out/r02
OUT=gpurun_out/r02/kernel_ab.log
: > $OUT
LIB=graph-embeddings_amd/lib/libgeglove.so
cp $LIB /tmp/libgeglove_keep.so
IFS='|' read -ra CASE_LIST <<< "$CASES"
for ROUND in $(seq 1 $ROUNDS); do
  for WHICH in $VARIANTS; do
    cp tools/r02/_ab/libgeglove_$WHICH.so $LIB
    for C in "${CASE_LIST[@]}"; do
      echo "== round $ROUND $WHICH $C" >> $OUT
      timeout -k 10 300 python3 bench.py $C --steps 10 --warmup 2 --no-cpu-baseline >> $OUT 2>&1 || { cp /tmp/libgeglove_keep.so $LIB; tail -5 $OUT; exit 1; }
    done
  done
done
cp /tmp/libgeglove_keep.so $LIB
grep -o '"kernel_ms": [0-9.]*\|== .*' $OUT | paste - -
