# run-time knobs of the wave-specialised layer, same box: bash tools/gpu_ws_ab.sh <tag>
cd $GRAFT_REPO_ROOT
for CFG in "RING=4 NT=1" "RING=3 NT=1" "RING=22 NT=1" "RING=4 NT=0" "RING=3 NT=0" "RING=22 NT=0" "RING=2 NT=0" "RING=4 NT=1"; do
  eval $CFG
  DGNN_WS_RING=$RING DGNN_WS_NT=$NT timeout 300 python bench.py --no-train --no-extras --no-cpu-baseline --steps 20 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$CFG', round(d['value']/1e6,2), d['ms_per_step'], {k:round(v,4) for k,v in d['config']['replay_breakdown_ms'].items()})"
done
