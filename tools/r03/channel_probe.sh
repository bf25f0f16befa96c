#!/bin/bash
# Per-instance (per L2 channel) memory-side counters of the update kernel on fast and slow placements of one handle.
# One gpurun call; every rocprofv3 pass is its own process (counters only with --kernel-trace, as the pool requires).
set -o pipefail
O=gpurun_out/r03/chan; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rocprofv3 -L > $O/counters_list.txt 2>&1 || true
python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_plain.json 2> $O/bench_plain.err || { tail -5 $O/bench_plain.err; exit 1; }
echo "bench: $(python3 -c "import json;d=json.loads(open('$O/bench_plain.json').read().strip().splitlines()[-1]);print(d['roofline']['kernel_ms'], d['trainer']['placements'], d['trainer']['placement_best_ms'], d['trainer']['placement_worst_ms'])")"
python3 tools/r03/place_modes.py 6 first_placement > $O/modes_plain.jsonl 2> $O/modes_plain.err || { tail -5 $O/modes_plain.err; exit 1; }
cat $O/modes_plain.jsonl
for P in "TCC_EA0_RDREQ TCC_EA0_WRREQ" "TCC_EA0_RDREQ_LEVEL TCC_EA0_WRREQ_LEVEL" "TCC_REQ TCC_EA0_ATOMIC" "TCC_EA0_RDREQ_DRAM TCC_EA0_WRREQ_DRAM" "TCC_EA0_RD_UNCACHED_32B TCC_EA0_RDREQ_32B" "TCC_EA0_WRREQ_STALL TCC_EA0_WRREQ_64B"; do
  T=$(echo $P | tr ' ' '+')
  rocprofv3 --pmc $P --kernel-trace --output-format json csv -d $O/pmc_$T -- python3 tools/r03/place_modes.py 5 first_placement > $O/modes_$T.jsonl 2> $O/pmc_$T.err || { echo "pass $T failed"; tail -3 $O/pmc_$T.err; continue; }
  echo "$T: $(cat $O/modes_$T.jsonl | python3 -c "import sys,json;print([json.loads(l)['epoch_ms'][1] for l in sys.stdin if l.startswith('{')])")"
done
python3 tools/r03/channel_summary.py $O > $O/summary.txt 2>&1 || tail -5 $O/summary.txt
find $O -name '*.json' -size +20M -delete
find $O -name '*.csv' -size +8M -delete
du -sh $O
