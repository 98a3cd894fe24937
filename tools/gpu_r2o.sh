cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_train.py tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -12 > gpurun_out/r2o_tests.log
cat gpurun_out/r2o_tests.log
for c in 1 0; do
  DGNN_TRAIN_COMPOSITE=$c python tools/bench_train.py --steps 60 > gpurun_out/r2o_train_c$c.json 2> gpurun_out/r2o_train_c$c.err
  cut -c1-400 gpurun_out/r2o_train_c$c.json; tail -2 gpurun_out/r2o_train_c$c.err
done
DGNN_TRAIN_COMPOSITE=1 DGNN_GEMM_MODE=f32 python tools/bench_train.py --steps 60 > gpurun_out/r2o_train_c1_f32.json 2>&1
cut -c1-400 gpurun_out/r2o_train_c1_f32.json
