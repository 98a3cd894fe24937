#!/bin/bash
# The inference suites under the switches that select another path (companion of tools/gpu_alt_paths.sh, which covers the training / parity / bf16 suites):
# the two-phase kernels instead of the wave-specialised ones, unprepared parameters, layer and decoder as two launches, the exact-fp32 mode, the wide
# aggregate's static walk.  Every configuration under its own timeout.
cd $GRAFT_REPO_ROOT
for e in "DGNN_WS=0" "DGNN_PREPARED=0" "DGNN_FUSE_DECODER=0" "DGNN_GEMM_MODE=f32" "DGNN_AGG_SR_TICKETS=0"; do
  echo "== $e"
  env $e timeout 900 python -m pytest tests/test_gpu_infer.py tests/test_gpu_ws.py tests/test_gpu_wide.py tests/test_gpu_scale.py tests/test_gpu_reorder.py -m gpu -q -x 2>&1 | tail -2
done
