cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2q_train_trace -- python3 tools/bench_train.py --steps 40 --warmup 5 > gpurun_out/r2q_train_trace.log 2>&1
tail -3 gpurun_out/r2q_train_trace.log | cut -c1-300
f=$(find gpurun_out/r2q_train_trace -name '*kernel_stats.csv' | head -1)
head -40 $f
