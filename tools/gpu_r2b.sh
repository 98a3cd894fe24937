# round-2 GPU call B: bf16 tests + fixed scale tests, bf16 bench, fp32 bench
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_gpu_bf16.py -m gpu -q -s > gpurun_out/r2b_bf16_tests.log 2>&1
echo "bf16 tests rc=$?" >> gpurun_out/r2b_bf16_tests.log
python -m pytest tests/test_gpu_scale.py tests/test_gpu_train.py -m gpu -q > gpurun_out/r2b_scale_tests.log 2>&1
echo "scale/train tests rc=$?" >> gpurun_out/r2b_scale_tests.log
python bench.py --dtype bf16 > gpurun_out/r2b_bench_bf16.json 2> gpurun_out/r2b_bench_bf16.err
echo "bench bf16 rc=$?" >> gpurun_out/r2b_bench_bf16.err
python bench.py --no-cpu-baseline > gpurun_out/r2b_bench.json 2> gpurun_out/r2b_bench.err
grep -E "passed|failed|rc=|Error|assert" gpurun_out/r2b_bf16_tests.log | tail -30; tail -3 gpurun_out/r2b_scale_tests.log; tail -c 1800 gpurun_out/r2b_bench_bf16.json; tail -3 gpurun_out/r2b_bench_bf16.err; head -c 300 gpurun_out/r2b_bench.json
