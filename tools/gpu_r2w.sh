cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "sampler or khop or block" 2>&1 | tail -3
DGNN_KHOP_ONE_CALL=0 python tools/diag_sampler.py 2>/dev/null
DGNN_KHOP_MAILBOX=0 python tools/diag_sampler.py 2>/dev/null
python tools/diag_sampler.py 2>/dev/null
for rep in 1 2; do
for mb in 0 1; do
for pf in stream thread; do
  DGNN_KHOP_MAILBOX=$mb python tools/bench_train.py --steps 80 --prefetch $pf 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('mailbox=$mb', d['model'],d['dtype'],d['block_builder'],d['ms_per_step'],d['final_loss'])"
done
done
done
