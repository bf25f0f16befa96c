#!/bin/bash
# Round-2 kernel-trace summaries of the two other device paths (the builder as changed this round, the similarity edges), final library.
O=gpurun_out/r02/side; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bca -- python3 tests/tools/bca_bench.py > $O/bca_bench.log 2> $O/bca.err || { tail -5 $O/bca.err; exit 1; }
rocprofv3 --kernel-trace --stats --output-format csv -d $O/sim -- python3 tests/tools/similarity_bench.py > $O/sim_bench.log 2> $O/sim.err || { tail -5 $O/sim.err; exit 1; }
mkdir -p $O/summary
cp $(find $O/bca -name '*kernel_stats.csv' | head -1) $O/summary/r02_bca_bench_kernel_stats.csv
cp $(find $O/sim -name '*kernel_stats.csv' | head -1) $O/summary/r02_similarity_bench_kernel_stats.csv
cp $O/bca_bench.log $O/summary/r02_bca_bench.log; cp $O/sim_bench.log $O/summary/r02_similarity_bench.log
find $O -name '*.csv' -size +3M -delete
cat $O/bca_bench.log; tail -5 $O/sim_bench.log; head -4 $O/summary/r02_bca_bench_kernel_stats.csv | cut -c1-200
