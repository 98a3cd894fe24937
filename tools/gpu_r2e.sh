cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_bf16.py -m gpu -q -s 2>&1 | grep -E "bf16:|passed|failed|Error|assert" > gpurun_out/r2e_bf16_tests.log
DGNN_BF16_MODE=single python -m pytest tests/test_gpu_bf16.py -m gpu -q -s -k "metric_graph or golden_f2" 2>&1 | grep -E "bf16:|passed|failed" >> gpurun_out/r2e_bf16_tests.log
python tools/variants.py --dtype bf16 --rounds 3 dgnn_amd/libdgnn_hip.so@DGNN_BF16_MODE=single dgnn_amd/libdgnn_hip.so@DGNN_BF16_MODE=compensated > gpurun_out/r2e_variants.log 2>&1
cat gpurun_out/r2e_bf16_tests.log; tail -8 gpurun_out/r2e_variants.log
