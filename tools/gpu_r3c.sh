cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -x 2>&1 | tail -4
for rep in 1 2; do
  python tools/bench_train.py --steps 80 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['model'],d['dtype'],d['block_builder'],d['ms_per_step'],d['final_loss'])"
done
for a in "--updated" "--updated --dtype bf16" "--dtype bf16"; do
  python tools/bench_train.py --steps 80 $a 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['model'],d['dtype'],d['block_builder'],d['ms_per_step'],d['final_loss'])"
done
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3c_train_trace -- python3 tools/bench_train.py --steps 40 --warmup 5 > gpurun_out/r3c_train_trace.log 2>&1
python tools/trace_gaps.py gpurun_out/r3c_train_trace/*/*kernel_trace.csv | head -64
