cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_train.py tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -6
for rep in 1 2; do
for pf in stream thread; do
  python tools/bench_train.py --steps 80 --prefetch $pf 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['model'],d['dtype'],d['block_builder'],d['ms_per_step'],d['final_loss'])"
done
done
DGNN_FUSED_LOSS=0 python tools/bench_train.py --steps 80 --prefetch stream 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('unfused loss', d['block_builder'],d['ms_per_step'],d['final_loss'])"
python tools/bench_train.py --steps 80 --prefetch stream --updated --dtype bf16 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['model'],d['dtype'],d['block_builder'],d['ms_per_step'],d['final_loss'])"
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2v_train_trace -- python3 tools/bench_train.py --steps 40 --warmup 5 --prefetch stream > gpurun_out/r2v_train_trace.log 2>&1
python tools/trace_gaps.py gpurun_out/r2v_train_trace/*/*kernel_trace.csv | head -60
