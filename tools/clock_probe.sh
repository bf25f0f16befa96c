#!/bin/bash
# Runs bench.py a few times; while each runs, samples the GPU clocks and power (rocm-smi) twice a second.
# Prints the epoch time of each run next to the clock levels seen while it ran.   bash tools/clock_probe.sh [runs]
runs=${1:-6}
for i in $(seq 1 $runs); do
  ( while true; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|fclk|socclk|Power" | tr -s " " | tr "\n" ";"; echo; sleep 0.5; done ) > gpurun_out/clk_$i.log &
  sampler=$!
  python bench.py --steps 20 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('run $i: kernel %.2f ms' % d['roofline']['kernel_ms'])"
  kill $sampler 2>/dev/null; wait $sampler 2>/dev/null
  tail -3 gpurun_out/clk_$i.log | head -1 | cut -c1-300
done
