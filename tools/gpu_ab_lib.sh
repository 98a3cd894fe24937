# same-box A/B of several builds of the library on the inference line:  bash tools/gpu_ab_lib.sh "<.so> <.so> ..."   (repo-relative; "-" = the default build)
cd $GRAFT_REPO_ROOT
for i in 1 2; do for L in $1; do
  P=""; [ "$L" != "-" ] && P=$GRAFT_REPO_ROOT/$L
  DGNN_LIB_PATH=$P timeout 300 python bench.py --no-train --no-extras --no-cpu-baseline --steps 20 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$L', round(d['value']/1e6,2), d['ms_per_step'], {k:round(v,4) for k,v in d['config']['replay_breakdown_ms'].items()})"
done; done
