# wave-specialised 128 -> 128 layers: parity suites + same-box A/B against the two-phase kernels:  bash tools/gpu_ws.sh <tag>
cd $GRAFT_REPO_ROOT
T=${1:-r5ws}; mkdir -p gpurun_out/$T
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_infer.py -x -q -m gpu > gpurun_out/$T/tests.log 2>&1; echo "parity+infer rc $?"; tail -n 3 gpurun_out/$T/tests.log
timeout 900 python -m pytest tests/test_gpu_scale.py -x -q -m gpu -k "metric or ignatius or out_of_range" > gpurun_out/$T/tests_scale.log 2>&1; echo "scale rc $?"; tail -n 2 gpurun_out/$T/tests_scale.log
for i in 1 2 3; do timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "row_level or last_layer_and_decoder" > gpurun_out/$T/rows_$i.log 2>&1; echo "row-level run $i rc $?"; done
for CFG in "WS=1" "WS=0" "WS=1" "WS=0"; do
  eval $CFG
  DGNN_WS=$WS timeout 300 python bench.py --no-train --no-extras --no-cpu-baseline --steps 20 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$CFG', round(d['value']/1e6,2), d['ms_per_step'], {k:round(v,4) for k,v in d['config']['replay_breakdown_ms'].items()})"
done
