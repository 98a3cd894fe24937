#!/bin/bash
# The switches of the wave-specialised kernels (csrc/fused_ws.hip) and of the split-row wide layers through the GPU suites that cover them:
# the two-phase kernels instead (DGNN_WS=0), the barrier hand-off, 3 / 4 ring slots, the 64 -> 128 layer / the 16-bit rows kept on the older kernels,
# stage B of the decoder on the producers only / the consumers only, layer and decoder as two launches, per-layer calls, wide layers on fp32 rows.
cd $GRAFT_REPO_ROOT
for e in "DGNN_WS=0" "DGNN_WS_RING=2" "DGNN_WS_RING=3" "DGNN_WS_RING=4" "DGNN_WS_64=0" "DGNN_WS_16=0" "DGNN_WS_NT=1" "DGNN_WS_NT=49" "DGNN_WS_NT=32"; do
  echo "== $e"
  env $e timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_infer.py tests/test_gpu_bf16.py tests/test_gpu_wide.py -m gpu -q -x 2>&1 | tail -2
done
# (tests/test_gpu_infer.py and tests/test_gpu_wide.py assert that the DEFAULT paths are taken: these three switches go through the other suites)
for e in "DGNN_FUSE_DECODER=0" "DGNN_INFER_ONE_CALL=0" "DGNN_WIDE_SR=0"; do
  echo "== $e"
  env $e timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bf16.py tests/test_gpu_scale.py -m gpu -q -x -k "not training" 2>&1 | tail -2
done
