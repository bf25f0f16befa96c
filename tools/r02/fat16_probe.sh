mkdir -p gpurun_out/r02
timeout -k 10 900 python -m pytest tests/test_glove_parity_gpu.py tests/test_parallel_gpu.py tests/test_configs_gpu.py tests/test_cli_gpu.py -x -q > gpurun_out/r02/fat16_tests.log 2>&1; rc=$?
tail -15 gpurun_out/r02/fat16_tests.log
[ $rc -eq 0 ] || exit $rc
VARIANTS="new fat16" CASES="--dim 200 --dtype bf16|--dim 300 --dtype bf16" ROUNDS=2 bash tools/r02/kernel_ab.sh
