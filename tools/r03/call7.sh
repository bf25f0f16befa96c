#!/bin/bash
set -o pipefail
O=gpurun_out/r03/call7; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
echo "== bca parity tests"
timeout -k 10 600 python3 -m pytest tests/test_bca_parity_gpu.py tests/test_golden.py tests/test_configs_gpu.py tests/test_cli_gpu.py -m gpu -q -x -k "not c4_c5 and not c2_full" > $O/bca_tests.log 2>&1; echo "pytest rc=$?"; tail -3 $O/bca_tests.log
echo "== bca bench (V = 300 050)"
GE_BCA_TIMING=1 python3 tests/tools/bca_bench.py --no-oracle > $O/bca_lds.log 2>&1; grep -v amdgpu.ids $O/bca_lds.log | tail -7
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_lds -- python3 tests/tools/bca_bench.py --no-oracle > $O/kt_lds.log 2>&1 || tail -3 $O/kt_lds.log
python3 - $O <<'PY'
import csv, glob, sys
O = sys.argv[1]
for f in glob.glob(O + "/kt_lds/**/*kernel_stats.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "k_bca" in row["Name"]:
            print(row["Name"][:60], "calls", row["Calls"], "total ms %.2f" % (float(row["TotalDurationNs"]) / 1e6), "max ms %.2f" % (float(row["MaxNs"]) / 1e6))
PY
echo "== hub segments sensitivity (8 ranks, sync)"
for seg in 4 8; do
  timeout -k 10 300 python3 tools/r03/ranks_debug.py 8 sync bf16 2 100000 3000000 200 8 $seg > $O/ranks_seg$seg.txt 2>&1; grep "^epoch\|^segments" $O/ranks_seg$seg.txt | cut -c1-60
done
echo "== parallel tests"
timeout -k 10 900 python3 -m pytest tests/test_glove_parity_gpu.py tests/test_parallel_gpu.py -m gpu -q -s -k "adam_amsgrad_hogwild_single or eight_ranks or failing_rank or bench_starts" > $O/partests.log 2>&1; echo "pytest rc=$?"; grep -n "eight ranks\|passed\|failed\|Error" $O/partests.log | head -40
