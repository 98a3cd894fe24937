# same-box A/B of DGNN_WS_NT knob values on the inference line:  bash tools/gpu_ab.sh "1 17 33 49"
cd $GRAFT_REPO_ROOT
for NT in $1 $1; do
  DGNN_WS_NT=$NT timeout 300 python bench.py --no-train --no-extras --no-cpu-baseline --steps 20 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('NT=$NT', round(d['value']/1e6,2), d['ms_per_step'], {k:round(v,4) for k,v in d['config']['replay_breakdown_ms'].items()})"
done
