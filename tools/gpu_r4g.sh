cd $GRAFT_REPO_ROOT
for e in "DGNN_TRAIN_COMPOSITE=0" "DGNN_TRAIN_WHOLE_MODEL=0" "DGNN_KHOP_ONE_CALL=0" "DGNN_CHAIN_DENSE=1" "DGNN_FUSED_LOSS=0" "DGNN_GEMM_MODE=f32" "DGNN_TRAIN_AUX_STREAM=1" "DGNN_X3_BIG=0 DGNN_X3_N64=0" "DGNN_KHOP_MAILBOX=0"; do
  echo "== $e"
  env $e python -m pytest tests/test_gpu_train.py tests/test_gpu_parity.py tests/test_gpu_bf16.py -m gpu -q -x 2>&1 | tail -2
done
