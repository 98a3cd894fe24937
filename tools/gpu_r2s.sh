cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_train.py tests/test_gpu_parity.py tests/test_gpu_bf16.py -m gpu -q -x 2>&1 | tail -15 > gpurun_out/r2s_tests.log
cat gpurun_out/r2s_tests.log
for a in "--updated" "--updated --dtype bf16" "" "--updated" "--updated --dtype bf16" ""; do
  python tools/bench_train.py --steps 60 $a 2>/dev/null | cut -c1-420
done
