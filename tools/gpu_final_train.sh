# training part of tools/gpu_final.sh (traces, bench lines, A/B) -- enough when only the Python host changed since the last full set
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
T=${1:-r5z}
for a in "" "--updated" "--updated --dtype bf16"; do
tag=${T}_train$(echo $a | tr -d ' -')
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag -- python3 tools/bench_train.py --steps 60 --warmup 100 --no-roofline $a > gpurun_out/$tag.log 2>&1
python tools/trace_gaps.py gpurun_out/$tag/*/*kernel_trace.csv 105 40 > gpurun_out/$tag.gaps.txt
done
for a in "" "--updated" "--updated --dtype bf16" "--dtype bf16"; do
  python tools/bench_train.py $a 2>/dev/null > gpurun_out/${T}_bench_train$(echo $a | tr -d ' -').json
done
python tools/ab_train.py whole=1 whole=0 composite=0 > gpurun_out/${T}_ab_train.txt 2>&1
python bench.py > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
for f in gpurun_out/${T}_bench*.json; do python -c "
import json,sys
j=json.loads(open('$f').read().strip().splitlines()[-1])
r=j.get('roofline') or {}
print('$f', j.get('value', j.get('targets_per_s')), j['ms_per_step'], str(r.get('kernel'))[:40], r.get('frac'), (j.get('check') or {}).get('ok'), (j.get('training_step') or {}).get('ms_per_step'))"; done
tail -3 gpurun_out/${T}_ab_train.txt
