#!/bin/bash
set -o pipefail
O=gpurun_out/r03/call3; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
echo "== xcd locality"; ./tools/r03/micro/xcd_locality > $O/xcd_hipmalloc.csv 2> $O/xcd.err || tail -3 $O/xcd.err
./tools/r03/micro/xcd_locality contiguous > $O/xcd_contig.csv 2>> $O/xcd.err || tail -3 $O/xcd.err
python3 - $O <<'PY'
import sys, numpy as np
for name in ("xcd_hipmalloc", "xcd_contig"):
    try:
        a = np.loadtxt("%s/%s.csv" % (sys.argv[1], name), delimiter=",", skiprows=2)
    except Exception as e:
        print(name, "unreadable", e); continue
    lat = a[:, 2:]
    print(name, "mean ticks per XCD", np.round(lat.mean(axis=0)).astype(int).tolist(), "overall min/median/max", int(lat.min()), int(np.median(lat)), int(lat.max()))
    near = lat.argmin(axis=1)
    print("   nearest-XCD histogram", np.bincount(near, minlength=8).tolist(), " mean (max - min) over XCDs per page %.0f" % (lat.max(axis=1) - lat.min(axis=1)).mean())
    print("   first 24 pages nearest XCD", near[:24].tolist())
PY
echo "== stride sweep, contiguous tables"
for PAD in 0 1 2 3 4 6 8 13 26; do
  GE_TABLE_ALLOC=contiguous GE_RECORD_PAD_LINES=$PAD python3 tools/r03/place_modes.py 2 first_placement > $O/stride_$PAD.jsonl 2> $O/stride_$PAD.err || { tail -3 $O/stride_$PAD.err; continue; }
  echo "pad $PAD: $(python3 -c "import json;print([(json.loads(l)['row_stride'], json.loads(l)['epoch_ms'][1]) for l in open('$O/stride_$PAD.jsonl') if l.startswith('{')])")"
done
echo "== default allocation, for this box"; python3 tools/r03/place_modes.py 5 first_placement > $O/modes_default.jsonl 2>/dev/null; python3 -c "import json;print([json.loads(l)['epoch_ms'][1] for l in open('$O/modes_default.jsonl') if l.startswith('{')])"
echo "== turn bench"
python3 tools/r03/turn_bench.py 200 f32 > $O/turn_f32.json 2> $O/turn_f32.err || tail -3 $O/turn_f32.err; cat $O/turn_f32.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/turn_kt -- python3 tools/r03/turn_bench.py 200 f32 > $O/turn_kt.json 2> $O/turn_kt.err || tail -3 $O/turn_kt.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/turn_pmcA -- python3 tools/r03/turn_bench.py 200 f32 > $O/turn_pmcA.json 2> $O/turn_pmcA.err || tail -3 $O/turn_pmcA.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/turn_pmcB -- python3 tools/r03/turn_bench.py 200 f32 > $O/turn_pmcB.json 2> $O/turn_pmcB.err || tail -3 $O/turn_pmcB.err
python3 - $O <<'PY'
import csv, glob, sys, collections
O = sys.argv[1]
for f in glob.glob(O + "/turn_kt/**/*kernel_stats.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "sync" in row["Name"] or "exchange" in row["Name"]:
            print("   ", row["Name"][:90], row["Calls"], "avg ns", row["AverageNs"])
acc = collections.defaultdict(list)
for d in ("turn_pmcA", "turn_pmcB"):
    for c in glob.glob(O + "/" + d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(c)):
            if "k_sync_turn_rows4" in row["Kernel_Name"]:
                acc[(row["Kernel_Name"][:80], row["Counter_Name"])].append(float(row["Counter_Value"]))
for k, v in acc.items():
    print("   ", k, ["%.4g" % x for x in v])
PY
echo "== new tests"
timeout -k 10 900 python3 -m pytest tests/test_glove_parity_gpu.py tests/test_configs_gpu.py tests/test_parallel_gpu.py tests/test_cli_gpu.py -m gpu -q -s -k "adam_amsgrad_hogwild_single or c4_c5 or eight_ranks or failing_rank or bench_starts or exits_nonzero or exchange_turn or two_ranks or bit_for_bit" > $O/newtests.log 2>&1; echo "pytest rc=$?"; grep -n "vs kernel model\|eight ranks\|V=5M\|passed\|failed\|Error" $O/newtests.log | head -60
find $O -name '*.csv' -size +8M -delete
