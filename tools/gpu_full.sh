# the whole GPU suite (no -x) + the default bench line:  bash tools/gpu_full.sh <tag>
cd $GRAFT_REPO_ROOT
T=${1:-full}; mkdir -p gpurun_out/$T
timeout 3000 python -m pytest tests -q -m gpu -p no:cacheprovider > gpurun_out/$T/tests.log 2>&1; echo "suite rc $?"; tail -n 15 gpurun_out/$T/tests.log
timeout 900 python bench.py > gpurun_out/$T/bench.json 2> gpurun_out/$T/bench.err; echo "bench rc $?"; python3 -c "
import json; d=json.load(open('gpurun_out/$T/bench.json')); print(d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['avg_launch_ms'], d['roofline'].get('also',{}).get('avg_launch_ms'))"
