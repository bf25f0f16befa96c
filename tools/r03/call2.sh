#!/bin/bash
# (1) translation-side counters of the update kernel on fast / slow placements, (2) physically contiguous record tables
# (hipExtMallocWithFlags + hipDeviceMallocContiguous, switched by GE_TABLE_ALLOC=contiguous), (3) the gpu test suite on the
# kernel whose counted stores are unconditional.
set -o pipefail
O=gpurun_out/r03/call2; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
python3 tools/r03/place_modes.py 6 first_placement > $O/modes_default.jsonl 2> $O/modes_default.err || { tail -5 $O/modes_default.err; exit 1; }
echo "default:    $(python3 -c "import sys,json;print([json.loads(l)['epoch_ms'][1] for l in open('$O/modes_default.jsonl') if l.startswith('{')])")"
GE_TABLE_ALLOC=contiguous python3 tools/r03/place_modes.py 6 first_placement > $O/modes_contig.jsonl 2> $O/modes_contig.err || { tail -5 $O/modes_contig.err; exit 1; }
echo "contiguous: $(python3 -c "import sys,json;print([json.loads(l)['epoch_ms'][1] for l in open('$O/modes_contig.jsonl') if l.startswith('{')])")"
GE_TABLE_ALLOC=contiguous python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --layout first_placement > $O/bench_contig.json 2> $O/bench_contig.err || tail -3 $O/bench_contig.err
python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --layout first_placement > $O/bench_first.json 2> $O/bench_first.err || tail -3 $O/bench_first.err
python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_search.json 2> $O/bench_search.err || tail -3 $O/bench_search.err
for f in contig first search; do echo "bench $f: $(python3 -c "import json;d=json.loads(open('$O/bench_$f.json').read().strip().splitlines()[-1]);print(d['roofline']['kernel_ms'], d['trainer']['placements'])")"; done
for P in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_CLIENT_UTCL1_INFLIGHT_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum" "TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum TCP_PENDING_STALL_CYCLES_sum" "GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE TCP_TCP_TA_ADDR_STALL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum"; do
  T=$(echo $P | tr ' ' '+' | cut -c1-60)
  rocprofv3 --pmc $P --kernel-trace --output-format csv -d $O/pmc_$T -- python3 tools/r03/place_modes.py 5 first_placement > $O/modes_$T.jsonl 2> $O/pmc_$T.err || { echo "pass $T failed"; tail -3 $O/pmc_$T.err; continue; }
  echo "$T: $(python3 -c "import sys,json;print([json.loads(l)['epoch_ms'][1] for l in open('$O/modes_$T.jsonl') if l.startswith('{')])")"
  python3 - "$O/pmc_$T" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for c in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(c)):
        if "k_adagrad_runs" in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, v in acc.items():
    print("   ", k, ["%.4g" % x for x in v])
PY
done
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > $O/gputest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/gputest.log
find $O -name '*.csv' -size +8M -delete
