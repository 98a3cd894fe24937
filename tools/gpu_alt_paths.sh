#!/bin/bash
# Every alternate code path behind an environment switch through the GPU suites that cover it (run on the GPU box through gpurun):
# the separate autograd Functions instead of the composite training calls, the dense edge chaining, the multi-launch block builder, the
# unfused loss, the bit-faithful fp32 matrix-core mode, the second training stream, the small GEMM tiles, the layer and the decoder as two launches,
# unprepared parameters, the training step's unfused launch chains, the row-at-a-time aggregate backward, the decoder's output Linear as its own Function;
# round 6: the VALU forms of the filter products, the three-launch loss, fresh gradient tensors per step, the ReLU mask from y, the unmasked dx store,
# the first layer's backward with the filter recomputed, the GEMM selections without the 64 x 64 tiles / split K / tile counts.
cd $GRAFT_REPO_ROOT
for e in "DGNN_TRAIN_COMPOSITE=0" "DGNN_TRAIN_WHOLE_MODEL=0" "DGNN_KHOP_ONE_CALL=0" "DGNN_CHAIN_DENSE=1" "DGNN_FUSED_LOSS=0" "DGNN_GEMM_MODE=f32" \
         "DGNN_TRAIN_AUX_STREAM=1" "DGNN_X3_BIG=0 DGNN_X3_N64=0" "DGNN_KHOP_MAILBOX=0" "DGNN_FUSE_DECODER=0" "DGNN_PREPARED=0" \
         "DGNN_TRAIN_FUSED=0" "DGNN_TRAIN_FUSED=1" "DGNN_AGG_CHUNKED=0" "DGNN_TRAIN_DECODER_IN_CALL=0" "DGNN_UPDATED_STACK=0" "DGNN_BF16_SMALL=0" "DGNN_AGG_GROUPED=0" "DGNN_UPDATED_TAIL_IN_CALL=0" "DGNN_TORCH_ADAM=1" \
         "DGNN_AGG_MFMA=0" "DGNN_AGG_MFMA=2" "DGNN_FILTER_MFMA=0 DGNN_GEMM_MODE=f32" "DGNN_KL_LOSS_ONE_LAUNCH=0" "DGNN_TRAIN_KEEP_GRADS=0" "DGNN_BN_ZMASK=0" "DGNN_UPDATED_MASK_DX=0" \
         "DGNN_AGG_BWD_NODX=0 DGNN_AGG_MFMA=0" "DGNN_GEMM_MID=0" "DGNN_SMALL_SPLITK=0" "DGNN_SMALL_BY_TILES=0"; do
  echo "== $e"
  env $e timeout 600 python -m pytest tests/test_gpu_train.py tests/test_gpu_parity.py tests/test_gpu_bf16.py -m gpu -q -x 2>&1 | tail -2
done
