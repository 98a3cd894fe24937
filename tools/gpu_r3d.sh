cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -x 2>&1 | tail -3
python tools/bench_agg.py 2>/dev/null | tail -5
python bench.py > gpurun_out/r3d_bench.json 2> gpurun_out/r3d_bench.err; tail -3 gpurun_out/r3d_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3d_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['check']['ok'], d['cpu_baseline']['value'])
print(json.dumps(d.get('training_step'))[:1500])
PY
