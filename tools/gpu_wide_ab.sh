# A/B of the wide kernels' run-time knobs on one box: bash tools/gpu_wide_ab.sh <tag>
cd $GRAFT_REPO_ROOT
T=${1:-r5ab}; mkdir -p gpurun_out/$T
timeout 600 python -m pytest tests/test_gpu_wide.py -x -q -m gpu > gpurun_out/$T/tests.log 2>&1; echo "tests rc $?"; tail -n 4 gpurun_out/$T/tests.log
DGNN_GEMM_SR_CPS=1 timeout 600 python -m pytest tests/test_gpu_wide.py -x -q -m gpu -k "linear or whole" > gpurun_out/$T/tests_cps1.log 2>&1; echo "tests cps1 rc $?"; tail -n 2 gpurun_out/$T/tests_cps1.log
for CFG in "CPS=2 NT=1" "CPS=1 NT=1" "CPS=2 NT=0" "CPS=1 NT=0" "CPS=2 NT=1"; do
  eval $CFG
  for W in 64,128,256,512 128,256,512,1024; do
    DGNN_GEMM_SR_CPS=$CPS DGNN_SR_NT=$NT python bench.py --widths $W --no-train --no-extras --no-cpu-baseline --steps 10 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$CFG', '$W', round(d['value']/1e6,2), d['ms_per_step'], {k:round(v,3) for k,v in d['config']['replay_breakdown_ms'].items()})"
  done
done
