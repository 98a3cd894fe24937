cd $GRAFT_REPO_ROOT
for R in 22; do DGNN_WS_RING=$R timeout 900 python tools/det_ws_real.py 200 2>&1 | grep -v amdgpu.ids | tail -n 8; done
for D in 1 0; do DGNN_FUSE_DECODER=$D timeout 600 python tools/det_ws.py 80 2>&1 | grep -v amdgpu.ids | tail -n 2; done
timeout 600 python -m pytest tests/test_gpu_infer.py -x -q -m gpu -k "ignatius" 2>&1 | tail -n 5
