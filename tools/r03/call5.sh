#!/bin/bash
set -o pipefail
O=gpurun_out/r03/call5; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
echo "== ranks debug with hub segments"
for cfg in "8 sync bf16 2 100000 3000000 200 8 0" "8 overlap bf16 2 100000 3000000 200 8 0" "8 sync bf16 2 100000 3000000 200 8 -1" "8 sync bf16 2 100000 3000000 200 8 32" "1 sync bf16 2 100000 3000000 200 8 0"; do
  timeout -k 10 300 python3 tools/r03/ranks_debug.py $cfg > $O/ranks_$(echo $cfg | tr ' ' '_').txt 2>&1; grep -v amdgpu.ids $O/ranks_$(echo $cfg | tr ' ' '_').txt
done
echo "== tests"
timeout -k 10 900 python3 -m pytest tests/test_glove_parity_gpu.py tests/test_parallel_gpu.py tests/test_cli_gpu.py -m gpu -q -s -k "adam_amsgrad_hogwild_single or eight_ranks or failing_rank or bench_starts or two_ranks or bit_for_bit or cli_two or exits_nonzero" > $O/newtests.log 2>&1; echo "pytest rc=$?"; grep -n "eight ranks\|passed\|failed\|Error" $O/newtests.log | head -40
echo "== convergence, device leg"
python3 tools/r03/convergence.py device --ref profiles/r03_convergence_oracle.npz --out $O/r03_convergence.json > $O/convergence.txt 2>&1; tail -3 $O/convergence.txt
