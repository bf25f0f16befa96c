#!/bin/bash
set -o pipefail
O=gpurun_out/r03/call8; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
echo "== bca parity tests (vertex lines)"
timeout -k 10 600 python3 -m pytest tests/test_bca_parity_gpu.py tests/test_golden.py tests/test_configs_gpu.py tests/test_cli_gpu.py -m gpu -q -x -k "not c4_c5 and not c2_full" > $O/bca_tests.log 2>&1; echo "pytest rc=$?"; tail -3 $O/bca_tests.log
echo "== bca bench (V = 300 050): lines / no lines"
GE_BCA_TIMING=1 python3 tests/tools/bca_bench.py --no-oracle > $O/bca_lines.log 2>&1; grep -v amdgpu.ids $O/bca_lines.log | tail -6
GE_BCA_LINES=0 GE_BCA_TIMING=1 python3 tests/tools/bca_bench.py --no-oracle > $O/bca_nolines.log 2>&1; grep "k_bca passes\|device:" $O/bca_nolines.log | tail -2
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_lds -- python3 tests/tools/bca_bench.py --no-oracle > $O/kt_lds.log 2>&1 || tail -3 $O/kt_lds.log
python3 - $O <<'PY'
import csv, glob, sys
O = sys.argv[1]
for f in glob.glob(O + "/kt_lds/**/*kernel_stats.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "k_bca" in row["Name"] or "k_pack" in row["Name"]:
            print(row["Name"][:60], "calls", row["Calls"], "total ms %.2f" % (float(row["TotalDurationNs"]) / 1e6), "max ms %.2f" % (float(row["MaxNs"]) / 1e6))
PY
echo "== eight ranks bf16"
timeout -k 10 600 python3 -m pytest tests/test_parallel_gpu.py -m gpu -q -s -k "eight_ranks_share and bf16" > $O/par_bf16.log 2>&1; echo "pytest rc=$?"; grep -n "eight ranks\|passed\|failed" $O/par_bf16.log | head
echo "== bench.py --gpus 2 rehearsal (self-launch, gloo, one device)"
GE_BENCH_BACKEND=gloo GE_BENCH_ONE_DEVICE=1 timeout -k 10 600 python3 bench.py --gpus 2 --steps 4 --warmup 2 --rows-per-gpu 100000 --nnz-per-gpu 12000000 > $O/bench_2ranks.json 2> $O/bench_2ranks.err; echo "rc=$?"; python3 -c "
import json;d=json.loads(open('$O/bench_2ranks.json').read().strip().splitlines()[-1]);print(d['n_gpus'], d['ms_per_step'], d['exchange'], [round(x,5) for x in d['mean_cost_per_step']])"
echo "== C4 at its own size on one GPU"
timeout -k 10 1000 python3 bench.py --shards-on-one-gpu 8 --steps 3 --warmup 1 --no-cpu-baseline > $O/c4_full_1gpu.json 2> $O/c4_full.err; echo "rc=$?"; tail -2 $O/c4_full.err; python3 -c "
import json;d=json.loads(open('$O/c4_full_1gpu.json').read().strip().splitlines()[-1]);print(d['config']['vocab'], d['config']['nnz_per_gpu'], d['ms_per_step'], d['value'], d['roofline']['frac'], d['gen_seconds'], d['create_seconds'], d['mean_cost_first_last'])"
