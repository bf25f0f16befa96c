#!/bin/bash
set -o pipefail
O=gpurun_out/r03/call4; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
echo "== ranks debug"
for cfg in "8 sync bf16 2" "8 sync f32 1" "4 sync bf16 2" "2 sync bf16 2" "1 sync bf16 2" "8 overlap bf16 2"; do
  timeout -k 10 200 python3 tools/r03/ranks_debug.py $cfg > $O/ranks_$(echo $cfg | tr ' ' '_').txt 2>&1; cat $O/ranks_$(echo $cfg | tr ' ' '_').txt | grep -v amdgpu.ids
done
echo "== turn bench (flat kernel), nt on / off"
python3 tools/r03/turn_bench.py 200 f32 > $O/turn_nt1.json 2> $O/turn.err || tail -3 $O/turn.err; cat $O/turn_nt1.json
GE_SYNC_NT=0 python3 tools/r03/turn_bench.py 200 f32 > $O/turn_nt0.json 2>> $O/turn.err || tail -3 $O/turn.err; cat $O/turn_nt0.json
python3 tools/r03/turn_bench.py 300 bf16 > $O/turn_bf16.json 2>> $O/turn.err || tail -3 $O/turn.err; cat $O/turn_bf16.json
echo "== tests"
timeout -k 10 900 python3 -m pytest tests/test_glove_parity_gpu.py tests/test_configs_gpu.py tests/test_parallel_gpu.py -m gpu -q -s -k "adam_amsgrad or c4_c5 or exchange_turn or two_ranks or bit_for_bit or failing_rank or bench_starts" > $O/newtests.log 2>&1; echo "pytest rc=$?"; grep -n "vs kernel model\|V=5M\|passed\|failed\|Error" $O/newtests.log | head -60
echo "== adam bench (sqrt now correctly rounded)"
python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --opt adam > $O/bench_adam.json 2> $O/bench_adam.err || tail -3 $O/bench_adam.err
python3 -c "import json;d=json.loads(open('$O/bench_adam.json').read().strip().splitlines()[-1]);print('adam kernel_ms', d['roofline']['kernel_ms'])"
python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench.json 2> $O/bench.err || tail -3 $O/bench.err
python3 -c "import json;d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]);print('adagrad kernel_ms', d['roofline']['kernel_ms'], d['trainer']['placements'])"
