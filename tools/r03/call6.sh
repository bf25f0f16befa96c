#!/bin/bash
# k_bca with its hot tables in LDS: parity tests, timing against the global-memory kernel, counters of both.
set -o pipefail
O=gpurun_out/r03/call6; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
echo "== bca parity tests"
timeout -k 10 600 python3 -m pytest tests/test_bca_parity_gpu.py tests/test_golden.py tests/test_configs_gpu.py tests/test_cli_gpu.py -m gpu -q -x -k "not c4_c5 and not c2_full" > $O/bca_tests.log 2>&1; echo "pytest rc=$?"; tail -3 $O/bca_tests.log
echo "== bca bench (V = 300 050), LDS tables / global tables"
GE_BCA_TIMING=1 python3 tests/tools/bca_bench.py > $O/bca_lds.log 2>&1; grep -v amdgpu.ids $O/bca_lds.log | tail -12
GE_BCA_TIMING=1 GE_BCA_TABLES=global python3 tests/tools/bca_bench.py --no-oracle > $O/bca_global.log 2>&1; grep -v amdgpu.ids $O/bca_global.log | tail -8
for V in lds global; do
  E=""; [ $V = global ] && export GE_BCA_TABLES=global || unset GE_BCA_TABLES
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$V -- python3 tests/tools/bca_bench.py --no-oracle > $O/kt_$V.log 2>&1 || tail -3 $O/kt_$V.log
  rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/pmcA_$V -- python3 tests/tools/bca_bench.py --no-oracle > $O/pmcA_$V.log 2>&1 || tail -3 $O/pmcA_$V.log
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum --kernel-trace --output-format csv -d $O/pmcB_$V -- python3 tests/tools/bca_bench.py --no-oracle > $O/pmcB_$V.log 2>&1 || tail -3 $O/pmcB_$V.log
done
unset GE_BCA_TABLES
python3 - $O <<'PY'
import csv, glob, sys, collections
O = sys.argv[1]
for V in ("lds", "global"):
    for f in glob.glob(O + "/kt_%s/**/*kernel_stats.csv" % V, recursive=True):
        for row in csv.DictReader(open(f)):
            if "k_bca" in row["Name"] or "k_gather_rows" in row["Name"] or "k_totals" in row["Name"]:
                print(V, row["Name"][:60], "calls", row["Calls"], "total ms %.2f" % (float(row["TotalDurationNs"]) / 1e6), "avg ms %.2f" % (float(row["AverageNs"]) / 1e6))
    acc = collections.defaultdict(float)
    for d in ("pmcA", "pmcB"):
        for c in glob.glob(O + "/%s_%s/**/*counter_collection.csv" % (d, V), recursive=True):
            for row in csv.DictReader(open(c)):
                if "k_bca" in row["Kernel_Name"]:
                    acc[row["Counter_Name"]] += float(row["Counter_Value"])
    print(V, {k: "%.4g" % v for k, v in acc.items()})
PY
find $O -name '*.csv' -size +8M -delete
echo "== parallel + adam tests again"
timeout -k 10 900 python3 -m pytest tests/test_glove_parity_gpu.py tests/test_parallel_gpu.py -m gpu -q -s -k "adam_amsgrad_hogwild_single or eight_ranks or failing_rank or bench_starts" > $O/partests.log 2>&1; echo "pytest rc=$?"; grep -n "eight ranks\|passed\|failed\|Error" $O/partests.log | head -40
