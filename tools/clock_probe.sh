#!/bin/bash
# Runs bench.py a few times; while each runs, samples the GPU clocks and power (rocm-smi) twice a second.
# Prints the epoch time of each run next to the clock levels seen while it ran.   bash tools/clock_probe.sh [runs]
runs=${1:-6}
for i in $(seq 1 $runs); do
  ( while true; do rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|fclk|Power|emperature" | tr -s " " | tr "\n" ";"; echo; sleep 0.5; done ) > gpurun_out/clk_$i.log &
  sampler=$!
  python bench.py --steps 20 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('run $i: kernel %.2f ms' % d['roofline']['kernel_ms'])"
  kill $sampler 2>/dev/null; wait $sampler 2>/dev/null
  # a sample from the middle of the run
  n=$(wc -l < gpurun_out/clk_$i.log); sed -n "$((n/2))p" gpurun_out/clk_$i.log | sed 's/GPU\[0\]\t\t: //g' | cut -c1-420
done
