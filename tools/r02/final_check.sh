mkdir -p gpurun_out/r02
python -m pytest tests -q -m gpu > gpurun_out/r02/gputest_full.log 2>&1; rc=$?
tail -6 gpurun_out/r02/gputest_full.log
[ $rc -eq 0 ] || exit $rc
python3 bench.py --steps 20 --warmup 5 > gpurun_out/r02/bench_final.json 2> gpurun_out/r02/bench_final.err || { tail -5 gpurun_out/r02/bench_final.err; exit 1; }
tail -1 gpurun_out/r02/bench_final.json | cut -c1-600
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
