#!/bin/bash
# Round-3 profile set, ONE gpurun call (one box): `python3 bench.py` un-profiled, under --kernel-trace --stats, and under the two
# counter passes (FETCH_SIZE | WRITE_SIZE cannot share one); the same for --dim 100 and --dim 300 --dtype bf16; the exchange pass at the
# 8-GPU table size (kernel stats + FETCH / WRITE); the builder (kernel stats + SQ / TCC counters).  tools/r03/profile_summary.py turns
# the CSVs into profiles/r03_* and profiles/traffic.json.
set -o pipefail
O=gpurun_out/r03/prof; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
PART=${1:-all}
if [ $PART = all ] || [ $PART = bench ]; then
for L in default dim100 bf16d300; do
  A=""; [ $L = dim100 ] && A="--dim 100"; [ $L = bf16d300 ] && A="--dim 300 --dtype bf16"
  python3 bench.py --steps 20 --warmup 5 $A > $O/bench_$L.json 2> $O/bench_$L.err || { tail -5 $O/bench_$L.err; exit 1; }
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$L -- python3 bench.py --no-cpu-baseline $A > $O/bench_kt_$L.json 2> $O/kt_$L.err || { tail -5 $O/kt_$L.err; exit 1; }
  rocprofv3 --pmc FETCH_SIZE TCC_EA0_ATOMIC_sum --kernel-trace --output-format csv -d $O/pmcA_$L -- python3 bench.py --no-cpu-baseline $A > $O/bench_pmcA_$L.json 2> $O/pmcA_$L.err || { tail -5 $O/pmcA_$L.err; exit 1; }
  rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/pmcB_$L -- python3 bench.py --no-cpu-baseline $A > $O/bench_pmcB_$L.json 2> $O/pmcB_$L.err || { tail -5 $O/pmcB_$L.err; exit 1; }
  echo "$L done: $(python3 -c "import json;d=json.loads(open('$O/bench_$L.json').read().strip().splitlines()[-1]);print(d['roofline']['kernel_ms'], d['value'])")"
done
python3 tools/r03/profile_summary.py $O
fi
if [ $PART = all ] || [ $PART = side ]; then
mkdir -p $O/summary
echo "== exchange pass"
for B in 4 8 16 32; do GE_SYNC_BLOCKS_PER_CU=$B python3 tools/r03/turn_bench.py 200 f32 > $O/turn_b$B.json 2>> $O/turn.err; echo "blocks/CU $B: $(python3 -c "import json;print(json.loads(open('$O/turn_b$B.json').read().strip().splitlines()[-1])['turn_ms'][1:4])")"; done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/turn_kt -- python3 tools/r03/turn_bench.py 200 f32 > $O/turn_kt.json 2> $O/turn_kt.err || tail -3 $O/turn_kt.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/turn_pmcA -- python3 tools/r03/turn_bench.py 200 f32 > $O/turn_pmcA.json 2> $O/turn_pmcA.err || tail -3 $O/turn_pmcA.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/turn_pmcB -- python3 tools/r03/turn_bench.py 200 f32 > $O/turn_pmcB.json 2> $O/turn_pmcB.err || tail -3 $O/turn_pmcB.err
echo "== builder"
GE_BCA_TIMING=1 python3 tests/tools/bca_bench.py > $O/bca_bench.log 2>&1; grep -v amdgpu.ids $O/bca_bench.log | tail -9
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bca_kt -- python3 tests/tools/bca_bench.py --no-oracle > $O/bca_kt.log 2>&1 || tail -3 $O/bca_kt.log
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/bca_pmcA -- python3 tests/tools/bca_bench.py --no-oracle > $O/bca_pmcA.log 2>&1 || tail -3 $O/bca_pmcA.log
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum --kernel-trace --output-format csv -d $O/bca_pmcB -- python3 tests/tools/bca_bench.py --no-oracle > $O/bca_pmcB.log 2>&1 || tail -3 $O/bca_pmcB.log
python3 tools/r03/side_summary.py $O
find $O -name '*.csv' -size +3M -delete
echo "== long run, 128 epochs"
python3 tools/r03/convergence.py device --epochs 128 --ref profiles/r03_convergence_oracle128.npz --out $O/summary/r03_convergence128.json > $O/convergence128.txt 2>&1; tail -2 $O/convergence128.txt | cut -c1-400
fi
