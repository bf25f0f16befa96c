#!/bin/bash
# Round-2 profile set of `python3 bench.py` (the driver's command shape), taken in ONE gpurun call:
#   1. un-profiled bench line                        -> profiles/r02_bench.json
#   2. rocprofv3 --kernel-trace --stats              -> profiles/r02_bench_kernel_stats.csv (+ the bench line it printed)
#   3. rocprofv3 --pmc, two passes (FETCH_SIZE | WRITE_SIZE cannot share one; MI355X_MICROARCH.md)  -> r02_bench_pmc_rows.csv
#   4. the same three for `--layout separate_tables` (the round-1 storage) as the counter evidence of the record layout
#      and for `--layout packed_records` (rows dim + 4 wide, records back to back: the storage before the alignment work)
#   5. and for `--dim 100` (BASELINE C2's width) and `--dim 300 --dtype bf16` (C5's rows): kernel time and traffic of those instances
# tools/r02/profile_summary.py turns the CSVs into profiles/r02_bench_pmc_summary.json and profiles/traffic.json.
set -o pipefail
O=gpurun_out/r02/prof; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
for L in default separate_tables packed_records dim100 bf16d300; do
  A=""; [ $L = separate_tables ] && A="--layout $L"; [ $L = packed_records ] && A="--layout $L"; [ $L = dim100 ] && A="--dim 100"
  [ $L = bf16d300 ] && A="--dim 300 --dtype bf16"
  python3 bench.py --steps 20 --warmup 5 $A > $O/bench_$L.json 2> $O/bench_$L.err || { tail -5 $O/bench_$L.err; exit 1; }
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$L -- python3 bench.py --no-cpu-baseline $A > $O/bench_kt_$L.json 2> $O/kt_$L.err || { tail -5 $O/kt_$L.err; exit 1; }
  rocprofv3 --pmc FETCH_SIZE TCC_EA0_ATOMIC_sum --kernel-trace --output-format csv -d $O/pmcA_$L -- python3 bench.py --no-cpu-baseline $A > $O/bench_pmcA_$L.json 2> $O/pmcA_$L.err || { tail -5 $O/pmcA_$L.err; exit 1; }
  rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/pmcB_$L -- python3 bench.py --no-cpu-baseline $A > $O/bench_pmcB_$L.json 2> $O/pmcB_$L.err || { tail -5 $O/pmcB_$L.err; exit 1; }
  case $L in dim100|bf16d300) echo "$L done"; continue;; esac          # the other widths: kernel time and traffic only
  rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_LEVEL_sum --kernel-trace --output-format csv -d $O/pmcC_$L -- python3 bench.py --no-cpu-baseline $A > $O/bench_pmcC_$L.json 2> $O/pmcC_$L.err || { tail -5 $O/pmcC_$L.err; exit 1; }
  rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VMEM SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/pmcD_$L -- python3 bench.py --no-cpu-baseline $A > $O/bench_pmcD_$L.json 2> $O/pmcD_$L.err || { tail -5 $O/pmcD_$L.err; exit 1; }
  echo "$L done"
done
python3 tools/r02/profile_summary.py $O
find $O -name '*.csv' -size +3M -delete
