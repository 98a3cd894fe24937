cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_bf16.py -m gpu -q -s 2>&1 | grep -E "bf16:|passed|failed|Error|assert" > gpurun_out/r2f_bf16_tests.log
python tools/variants.py --dtype bf16 --rounds 3 dgnn_amd/libdgnn_hip.so@DGNN_BF16_MODE=single dgnn_amd/libdgnn_hip.so@DGNN_BF16_MODE=compensated > gpurun_out/r2f_variants.log 2>&1
python bench.py --dtype bf16 > gpurun_out/r2f_bench_bf16.json 2> gpurun_out/r2f_bench_bf16.err
cat gpurun_out/r2f_bf16_tests.log | cut -c1-220; tail -6 gpurun_out/r2f_variants.log; tail -c 900 gpurun_out/r2f_bench_bf16.json; tail -2 gpurun_out/r2f_bench_bf16.err
