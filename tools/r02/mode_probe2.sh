#!/bin/bash
# fresh processes under rocprofv3 --pmc, one counter set per process, rotating: which counter moves with the epoch time?
set -o pipefail
O=gpurun_out/r02/p2; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
python3 tools/r02/mode_probe.py warm 2 > $O/warm.json 2> $O/warm.err || exit 1
S1="TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE"
S2="TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_LEVEL_sum"
S3="TCC_EA0_WRREQ_STALL_sum TCC_TAG_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum"
S4="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VMEM TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum"
for i in 1 2 3 4; do
  n=0
  for S in "$S1" "$S2" "$S3" "$S4"; do
    n=$((n+1))
    rocprofv3 --pmc $S --kernel-trace --output-format csv -d $O/s${n}_$i -- python3 tools/r02/mode_probe.py s${n}_$i 3 > $O/s${n}_$i.json 2> $O/s${n}_$i.err || { tail -5 $O/s${n}_$i.err; exit 1; }
  done
  python3 tools/r02/mode_probe.py plain_$i 3 > $O/plain_$i.json 2> $O/plain_$i.err || exit 1
  echo "round $i done"
done
python3 tools/r02/mode_probe.py recreate 3 4 > $O/recreate.json 2> $O/recreate.err
python3 - <<'P'
import csv, glob, json, os, collections
O='gpurun_out/r02/p2'
for f in sorted(glob.glob(O+'/*.json')):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f, 'unreadable'); continue
    line=[os.path.basename(f)]+[str(r['kernel_ms']) for r in d['runs']]
    tagdir=O+'/'+d['tag']
    acc=collections.defaultdict(list)
    for c in glob.glob(tagdir+'/**/*counter_collection.csv', recursive=True):
        for row in csv.DictReader(open(c)):
            if 'k_adagrad_runs' in row.get('Kernel_Name',''):
                acc[row['Counter_Name']].append(float(row['Counter_Value']))
    line += ['%s=%.4g' % (k, sum(v)/len(v)) for k,v in sorted(acc.items())]
    print(' '.join(line))
P
# keep only the small csv files
find $O -name '*.csv' -size +2M -delete
