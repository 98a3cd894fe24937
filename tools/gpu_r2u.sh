cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_train.py tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -4
for rep in 1 2; do
for oc in 0 1; do
for pf in stream thread; do
  DGNN_KHOP_ONE_CALL=$oc python tools/bench_train.py --steps 80 --prefetch $pf 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('one_call=$oc', d['model'],d['dtype'],d['block_builder'],d['ms_per_step'],d['final_loss'])"
done
done
done
