cd $GRAFT_REPO_ROOT
for a in "" "--dtype bf16" "--updated" "--updated --dtype bf16"; do
  python tools/bench_train.py --steps 60 $a 2>/dev/null | cut -c1-420
done
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2r_upd_bf16_trace -- python3 tools/bench_train.py --steps 40 --warmup 5 --updated --dtype bf16 > gpurun_out/r2r_upd_trace.log 2>&1
f=$(find gpurun_out/r2r_upd_bf16_trace -name '*kernel_stats.csv' | head -1)
head -30 $f | cut -c1-200
