cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2i_train_trace -- python3 tools/bench_train.py --steps 10 --warmup 3 > gpurun_out/r2i_train_trace.log 2>&1
python3 - <<'PY'
import glob, pandas as pd
f = glob.glob('gpurun_out/r2i_train_trace/*/*kernel_stats.csv')[0]
d = pd.read_csv(f)
d['Name'] = d['Name'].str.replace('(anonymous namespace)::','').str.replace('void ','').str.slice(0,60)
print(d[['Name','Calls','TotalDurationNs','AverageNs','Percentage']].head(40).to_string(index=False))
print('total calls', d['Calls'].sum(), 'total ms', d['TotalDurationNs'].sum()/1e6)
PY
