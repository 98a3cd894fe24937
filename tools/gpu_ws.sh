# wave-specialised 128 -> 128 layers: parity suites, determinism, same-box A/B of the knobs:  bash tools/gpu_ws.sh <tag> "<knob values>"
cd $GRAFT_REPO_ROOT
T=${1:-r5ws}; mkdir -p gpurun_out/$T
timeout 600 python tools/det_ws_layer.py 3000 300 1 2>&1 | grep -v amdgpu.ids | tail -n 4
timeout 600 python tools/det_ws_layer.py 20000 100 1 2>&1 | grep -v amdgpu.ids | tail -n 4
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_infer.py -x -q -m gpu > gpurun_out/$T/tests.log 2>&1; echo "parity+infer rc $?"; tail -n 3 gpurun_out/$T/tests.log
timeout 900 python -m pytest tests/test_gpu_scale.py -x -q -m gpu -k "metric or ignatius or out_of_range" > gpurun_out/$T/tests_scale.log 2>&1; echo "scale rc $?"; tail -n 2 gpurun_out/$T/tests_scale.log
timeout 600 python tools/det_ws.py 80 2>&1 | grep -v amdgpu.ids | tail -n 2
# (tools/ws_timing.py of round 5 was a one-off and is gone: bench.py --no-extras prints the same per-layer replay times)
bash tools/gpu_ab.sh "${2:-1}"
DGNN_WS=0 bash tools/gpu_ab.sh "1"
