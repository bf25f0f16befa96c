#!/bin/bash
# Does the start alignment of a record (row | accumulator row) and of the accumulator row inside it matter?  Probe build:
# GE_PROBE_ROW_ALIGN = offset of the accumulator row rounded up to that many floats, GE_PROBE_RECORD_ALIGN = record stride likewise.
# (The probe build was glove.hip with two getenv lines in ge_glove_create_impl -- `ds` and the accumulator row's offset rounded up to the
# given number of floats -- linked as tools/r02/_ab/libgeglove_align.so; not kept: the result became the default layout.)
# Fresh process per run, alternating.   CASES: "rowalign:recalign" pairs.
mkdir -p gpurun_out/r02
OUT=gpurun_out/r02/align_probe.log
: > $OUT
LIB=graph-embeddings_amd/lib/libgeglove.so
cp $LIB /tmp/libgeglove_keep.so
cp tools/r02/_ab/libgeglove_align.so $LIB
for ROUND in 1 2; do
  for A in 0:0 0:16 0:32 16:0 16:32 32:0 32:64; do
    for D in 200 100 300; do
      echo "== round $ROUND row:record align $A dim $D" >> $OUT
      GE_PROBE_ROW_ALIGN=${A%%:*} GE_PROBE_RECORD_ALIGN=${A##*:} timeout -k 10 300 python3 bench.py --dim $D --steps 10 --warmup 2 --no-cpu-baseline >> $OUT 2>&1 || { cp /tmp/libgeglove_keep.so $LIB; tail -5 $OUT; exit 1; }
    done
  done
done
cp /tmp/libgeglove_keep.so $LIB
grep -o '"kernel_ms": [0-9.]*\|== .*' $OUT | paste - -
