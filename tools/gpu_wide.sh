# wide-layer measurement set (run on the GPU box through gpurun): tests, both --widths bench lines, kernel traces
#   bash tools/gpu_wide.sh <tag>
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
T=${1:-r5w}
mkdir -p gpurun_out/$T
timeout 900 python -m pytest tests/test_gpu_wide.py -x -q -m gpu > gpurun_out/$T/tests.log 2>&1; echo "tests rc $?"; tail -n 6 gpurun_out/$T/tests.log
for W in 64,128,256,512 128,256,512,1024; do
  N=$(echo $W | tr ',' '_')
  timeout 300 python bench.py --widths $W --no-train --no-extras --steps 10 > gpurun_out/$T/bench_$N.json 2> gpurun_out/$T/bench_$N.err; echo "bench $W rc $?"
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$T/trace_$N -- python3 bench.py --widths $W --no-train --no-extras --no-cpu-baseline --steps 5 --warmup 2 > gpurun_out/$T/trace_$N.log 2>&1
done
python3 - <<'P'
import json,glob,csv,os,sys
T=sys.argv[1] if len(sys.argv)>1 else os.environ.get("T","r5w")
P
for f in gpurun_out/$T/bench_*.json; do python3 -c "
import json,sys
d=json.load(open('$f')); print('$f', d['value'], d['ms_per_step'], d['check']['max_abs_err'], d['check']['ok'], d['config']['replay_breakdown_ms'])"; done
for f in $(find gpurun_out/$T -name "*kernel_stats.csv"); do echo $f; head -8 $f | cut -d, -f1-4 | cut -c1-150; done
