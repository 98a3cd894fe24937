cd $GRAFT_REPO_ROOT
timeout 600 python tools/det_ws_layer.py 3000 300 1 2>&1 | grep -v amdgpu.ids | tail -n 30
timeout 600 python tools/det_ws_layer.py 20000 100 1 2>&1 | grep -v amdgpu.ids | tail -n 12
