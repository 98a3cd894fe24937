# same-box A/B of environment settings on an inference line:  bash tools/gpu_ab_env.sh "VAR=a VAR=b ..." [bench.py args]   (each item one setting, "-" = none)
cd $GRAFT_REPO_ROOT
S_=$1; shift
for i in 1 2; do for S in $S_; do
  E=""; [ "$S" != "-" ] && E="$S"
  env $E timeout 300 python bench.py --no-train --no-extras --no-cpu-baseline --steps 20 "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$S', round(d['value']/1e6,2), d['ms_per_step'], {k:round(v,4) for k,v in d['config']['replay_breakdown_ms'].items()})"
done; done
