# round-2 GPU call A: new parity-at-size tests, the whole GPU suite, default bench
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_gpu_scale.py tests/test_gpu_train.py -m gpu -x -q -s > gpurun_out/r2a_new_tests.log 2>&1
echo "new tests rc=$?" >> gpurun_out/r2a_new_tests.log
python -m pytest tests -m gpu -q > gpurun_out/r2a_all_tests.log 2>&1
echo "all tests rc=$?" >> gpurun_out/r2a_all_tests.log
python bench.py > gpurun_out/r2a_bench.json 2> gpurun_out/r2a_bench.err
echo "bench rc=$?" >> gpurun_out/r2a_bench.err
tail -5 gpurun_out/r2a_new_tests.log; tail -5 gpurun_out/r2a_all_tests.log; tail -c 1500 gpurun_out/r2a_bench.json; tail -3 gpurun_out/r2a_bench.err
