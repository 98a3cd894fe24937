cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_bf16.py -m gpu -q -s 2>&1 | grep -E "bf16:|passed|failed|Error|assert" > gpurun_out/r2c_bf16_tests.log
bash tools/prof_round2.sh r2c_bf16 --dtype bf16
python tools/variants.py --rounds 3 dgnn_amd/libdgnn_hip.so > gpurun_out/r2c_variants.log 2>&1
cat gpurun_out/r2c_bf16_tests.log; tail -20 gpurun_out/r2c_variants.log
